// What bandwidth do the stream MIXES of this package's dense kernels reach on their own (no arithmetic)?
//   copy4      : one float4 stream in, one out (the classic copy)
//   planar K/M : K planar float streams read, M written (hier_iteration: warp x3, canonical, previous gradient x3 read;
//                gradient x3 written), unit-stride dwords
//   planar+f4  : the same plus one float4 stream read (the packed live field, without the gather)
//   stencil    : planar K/M with the three "previous gradient" streams read through a 7-point stencil (z -/+ 1 slices)
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/probe/streams.hip -o /tmp/streams && /tmp/streams [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));

__global__ void copy4(const vf4* __restrict__ a, vf4* __restrict__ b, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}

template <int K, int M, bool F4, bool STENCIL>
__global__ void planar(const float* __restrict__ in, float* __restrict__ out, const vf4* __restrict__ p4, long long n,
                       int nx, int ny) {
    const long long slice = (long long)nx * ny;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float* a = in + k * n;
            if (STENCIL && k < 3) {
                const long long zm = i >= slice ? i - slice : i, zp = i + slice < n ? i + slice : i;
                const long long ym = i >= nx ? i - nx : i, yp = i + nx < n ? i + nx : i;
                const long long xm = i > 0 ? i - 1 : i, xp = i + 1 < n ? i + 1 : i;
                acc += a[i] + a[zm] + a[zp] + a[ym] + a[yp] + a[xm] + a[xp];
            } else {
                acc += __builtin_nontemporal_load(a + i);
            }
        }
        if (F4) {
            const vf4 v = p4[i];
            acc += v.x + v.y + v.z + v.w;
        }
#pragma unroll
        for (int m = 0; m < M; ++m) __builtin_nontemporal_store(acc + (float)m, out + m * n + i);
    }
}

// the same streams as planar<7, 3, true, STENCIL> with the three stencil planes read by a MARCH along z: a block owns a
// (64 x 4) tile and `chunk` consecutive slices, every thread keeps (z - 1, z, z + 1) of its three planes in registers and
// loads only the new z + 1 values per step; x -/+ 1 and y -/+ 1 of slice z come from the cache (they were some thread's
// z + 1 values one step earlier)
__global__ void march(const float* __restrict__ in, float* __restrict__ out, const vf4* __restrict__ p4, int nx, int ny,
                      int nz, int chunk) {
    const long long n = (long long)nx * ny * nz, slice = (long long)nx * ny;
    const int tiles_x = nx / 64, tiles_y = ny / 4;
    const int tile = blockIdx.x % (tiles_x * tiles_y), zc = blockIdx.x / (tiles_x * tiles_y);
    const int x = (tile % tiles_x) * 64 + (threadIdx.x & 63), y = (tile / tiles_x) * 4 + threadIdx.x / 64;
    const int z0 = zc * chunk, z1 = min(z0 + chunk, nz);
    const long long col = (long long)y * nx + x;
    float zm[3], c[3], zp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* a = in + k * n;
        c[k] = a[z0 * slice + col];
        zm[k] = a[(z0 > 0 ? z0 - 1 : z0) * slice + col];
    }
    for (int z = z0; z < z1; ++z) {
        const long long i = z * slice + col;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* a = in + k * n;
            zp[k] = a[(z + 1 < nz ? z + 1 : z) * slice + col];
            const long long ym = y > 0 ? i - nx : i, yp = y + 1 < ny ? i + nx : i;
            const long long xm = x > 0 ? i - 1 : i, xp = x + 1 < nx ? i + 1 : i;
            acc += c[k] + zm[k] + zp[k] + a[ym] + a[yp] + a[xm] + a[xp];
        }
#pragma unroll
        for (int k = 3; k < 7; ++k) acc += __builtin_nontemporal_load(in + k * n + i);
        const vf4 v = p4[i];
        acc += v.x + v.y + v.z + v.w;
#pragma unroll
        for (int m = 0; m < 3; ++m) __builtin_nontemporal_store(acc + (float)m, out + m * n + i);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            zm[k] = c[k];
            c[k] = zp[k];
        }
    }
}

// the same streams with the three stencil planes staged through LDS: a block owns a (64 x 4 x TZ) tile, copies tile + one
// voxel of shell per plane into LDS (clamped at the array's faces), and every thread handles the TZ voxels of its (x, y)
template <int TZ>
__global__ void lds3d(const float* __restrict__ in, float* __restrict__ out, const vf4* __restrict__ p4, int nx, int ny,
                      int nz) {
    constexpr int HX = 66, HY = 6, HZ = TZ + 2, SH = HX * HY * HZ;
    __shared__ float tile[3][SH];
    const long long n = (long long)nx * ny * nz, slice = (long long)nx * ny;
    const int tiles_x = nx / 64, tiles_y = ny / 4;
    const int t2 = blockIdx.x % (tiles_x * tiles_y), zc = blockIdx.x / (tiles_x * tiles_y);
    const int x0 = (t2 % tiles_x) * 64, y0 = (t2 / tiles_x) * 4, z0 = zc * TZ;
    for (int k = 0; k < 3; ++k) {
        const float* a = in + k * n;
        for (int e = threadIdx.x; e < SH; e += 256) {
            const int hx = e % HX, r = e / HX, hy = r % HY, hz = r / HY;
            const int gx = min(max(x0 - 1 + hx, 0), nx - 1), gy = min(max(y0 - 1 + hy, 0), ny - 1);
            const int gz = min(max(z0 - 1 + hz, 0), nz - 1);
            tile[k][e] = a[gz * slice + (long long)gy * nx + gx];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x / 64;
    for (int lz = 0; lz < TZ; ++lz) {
        const long long i = (z0 + lz) * slice + (long long)(y0 + ly) * nx + x0 + lx;
        const int c = ((lz + 1) * HY + ly + 1) * HX + lx + 1;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            acc += tile[k][c] + tile[k][c - 1] + tile[k][c + 1] + tile[k][c - HX] + tile[k][c + HX] + tile[k][c - HX * HY] +
                   tile[k][c + HX * HY];
#pragma unroll
        for (int k = 3; k < 7; ++k) acc += __builtin_nontemporal_load(in + k * n + i);
        const vf4 v = p4[i];
        acc += v.x + v.y + v.z + v.w;
#pragma unroll
        for (int m = 0; m < 3; ++m) __builtin_nontemporal_store(acc + (float)m, out + m * n + i);
    }
}

template <class F>
static double time_ms(F&& launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main(int argc, char** argv) {
    const int n1 = argc > 1 ? atoi(argv[1]) : 256;
    const long long n = (long long)n1 * n1 * n1;
    float *in, *out;
    vf4 *a4, *b4;
    hipMalloc(&in, 8 * n * 4);
    hipMalloc(&out, 6 * n * 4);  // up to six planes written (planar 7 / 6)
    hipMalloc(&a4, n * 16);
    hipMalloc(&b4, n * 16);
    hipMemset(in, 0, 8 * n * 4);
    hipMemset(a4, 0, n * 16);
    const int blocks = 127 * 8, threads = 256;  // the persistent grid of hier_iteration_kernel
    auto report = [&](const char* name, double bytes_per_voxel, double ms) {
        printf("%-34s %6.1f B/voxel  %.4f ms  %.0f GB/s\n", name, bytes_per_voxel, ms, bytes_per_voxel * n / ms / 1e6);
    };
    report("copy4 (1016 blocks)", 32, time_ms([&] { hipLaunchKernelGGL(copy4, dim3(blocks), dim3(threads), 0, 0, a4, b4, n); }));
    report("copy4 (8192 blocks)", 32, time_ms([&] { hipLaunchKernelGGL(copy4, dim3(8192), dim3(threads), 0, 0, a4, b4, n); }));
    report("planar 7 read / 3 written", 40, time_ms([&] { hipLaunchKernelGGL((planar<7, 3, false, false>), dim3(blocks), dim3(threads), 0, 0, in, out, a4, n, n1, n1); }));
    report("planar 7 / 3 + float4 read", 56, time_ms([&] { hipLaunchKernelGGL((planar<7, 3, true, false>), dim3(blocks), dim3(threads), 0, 0, in, out, a4, n, n1, n1); }));
    report("planar 7 / 3 + float4, 7-pt stencil", 56, time_ms([&] { hipLaunchKernelGGL((planar<7, 3, true, true>), dim3(blocks), dim3(threads), 0, 0, in, out, a4, n, n1, n1); }));
    report("planar 4 / 3 + float4 (no Tikhonov)", 44, time_ms([&] { hipLaunchKernelGGL((planar<4, 3, true, false>), dim3(blocks), dim3(threads), 0, 0, in, out, a4, n, n1, n1); }));
    report("planar 7 / 6 + float4 (update)", 68, time_ms([&] { hipLaunchKernelGGL((planar<7, 6, true, false>), dim3(blocks), dim3(threads), 0, 0, in, out, a4, n, n1, n1); }));
    report("planar 7 / 3 + float4, 8192 blocks", 56, time_ms([&] { hipLaunchKernelGGL((planar<7, 3, true, false>), dim3(8192), dim3(threads), 0, 0, in, out, a4, n, n1, n1); }));
    report("LDS-staged stencil, 64x4x4 tiles", 56, time_ms([&] { hipLaunchKernelGGL(lds3d<4>, dim3((n1 / 64) * (n1 / 4) * (n1 / 4)), dim3(256), 0, 0, in, out, a4, n1, n1, n1); }));
    report("LDS-staged stencil, 64x4x8 tiles", 56, time_ms([&] { hipLaunchKernelGGL(lds3d<8>, dim3((n1 / 64) * (n1 / 4) * (n1 / 8)), dim3(256), 0, 0, in, out, a4, n1, n1, n1); }));
    for (int chunk : {8, 16, 32, 64}) {
        char name[64];
        snprintf(name, sizeof name, "z-march, chunks of %d slices", chunk);
        const int nblocks = (n1 / 64) * (n1 / 4) * ((n1 + chunk - 1) / chunk);
        report(name, 56, time_ms([&] { hipLaunchKernelGGL(march, dim3(nblocks), dim3(256), 0, 0, in, out, a4, n1, n1, n1, chunk); }));
    }
    return 0;
}
