// The access pattern of a BRICK walk with no arithmetic (companion of list_taps.hip): a 256-thread workgroup takes a brick
// of 8 x 8 x 16 voxels (z, y, x) that holds band voxels, copies brick + one-voxel shell (10 x 10 x 18 float4) into LDS
// with unit-stride loads, compacts the brick's band voxels into wave-units and every unit reads its 19 taps out of LDS
// (ds_read_b128), adds canonical and stores one float4.  STAGES = 2: the next brick's shell is fetched into registers
// while the current one is worked on.
// hipcc -O3 --offload-arch=gfx950 tools/probe/brick_taps.hip -o /tmp/brick_taps && /tmp/brick_taps [n] [blocks per CU]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));
constexpr int BX = 16, BY = 8, BZ = 8, HX = BX + 2, HY = BY + 2, HZ = BZ + 2, SHELL = HX * HY * HZ, VOX = BX * BY * BZ;
constexpr int STAGE = (SHELL + 255) / 256;

template <bool PREFETCH>
__global__ __launch_bounds__(256) void bricks(const vf4* __restrict__ state, const float* __restrict__ canonical,
                                              vf4* __restrict__ out, const int* __restrict__ brick_list, unsigned n_bricks,
                                              const unsigned char* __restrict__ band, int nx, int ny, int nz) {
    __shared__ vf4 tile[SHELL];
    __shared__ float cn_s[VOX];
    __shared__ unsigned short ids[VOX];
    __shared__ unsigned s_count;
    const unsigned t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nbx = nx / BX, nby = ny / BY;
    int off[STAGE];  // this thread's shell elements relative to the brick's origin (interior bricks: no clamping needed)
#pragma unroll
    for (int m = 0; m < STAGE; ++m) {
        const int k = min((int)t + m * 256, SHELL - 1);
        const int hx = k % HX, r = k / HX, hy = r % HY, hz = r / HY;
        off[m] = ((hz - 1) * ny + (hy - 1)) * nx + hx - 1;
    }
    auto origin = [&](unsigned b) {
        const int brick = brick_list[b];
        const int bxy = brick / nbx;
        return (((bxy / nby) * BZ) * ny + (bxy % nby) * BY) * nx + (brick - bxy * nbx) * BX;
    };
    vf4 staged[STAGE];
    unsigned b = blockIdx.x;
    int o = b < n_bricks ? origin(b) : 0;
    if (PREFETCH && b < n_bricks) {
#pragma unroll
        for (int m = 0; m < STAGE; ++m) staged[m] = state[o + off[m]];
    }
    for (; b < n_bricks; b += gridDim.x) {
        if (t == 0) s_count = 0u;
        if (!PREFETCH) {
#pragma unroll
            for (int m = 0; m < STAGE; ++m) staged[m] = state[o + off[m]];
        }
        float cn4[4];
        bool in4[4];
        int gi4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v = j * 256 + t;
            gi4[j] = o + ((int)(v / (BX * BY)) * ny + (int)((v / BX) & (BY - 1))) * nx + (int)(v & (BX - 1));
            cn4[j] = canonical[gi4[j]];
            in4[j] = band[gi4[j]] != 0;
        }
        __syncthreads();  // the previous brick's units are done with tile / ids / cn_s
#pragma unroll
        for (int m = 0; m < STAGE; ++m)
            if (t + m * 256 < (unsigned)SHELL) tile[t + m * 256] = staged[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v = j * 256 + t;
            cn_s[v] = cn4[j];
            const unsigned long long mask = __ballot(in4[j]);
            if (mask) {
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(&s_count, (unsigned)__popcll(mask));
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                if (in4[j])
                    ids[base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] =
                        (unsigned short)v;
            }
        }
        __syncthreads();
        const unsigned next = b + gridDim.x;
        const int o_next = next < n_bricks ? origin(next) : o;
        if (PREFETCH && next < n_bricks) {
#pragma unroll
            for (int m = 0; m < STAGE; ++m) staged[m] = state[o_next + off[m]];
        }
        const unsigned count = s_count;
        for (unsigned u = wave; u * 64 < count; u += 4) {
            const unsigned k = u * 64 + lane;
            const unsigned v = ids[k < count ? k : count - 1];
            const int lx = v & (BX - 1), ly = (v / BX) & (BY - 1), lz = v / (BX * BY);
            const int c0 = ((lz + 1) * HY + ly + 1) * HX + lx + 1;
            vf4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        if ((dx != 0) + (dy != 0) + (dz != 0) > 2) continue;
                        acc += tile[c0 + (dz * HY + dy) * HX + dx];
                    }
            acc.y += cn_s[v];
            if (k < count) out[o + (lz * ny + ly) * nx + lx] = acc;
        }
        o = o_next;
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 3;
    const long long N = (long long)n * n * n;
    std::vector<unsigned char> band(N, 0);
    const float r = 0.3f * n, c = n / 2.0f;
    unsigned count = 0;
    for (int z = 1; z < n - 1; ++z)
        for (int y = 1; y < n - 1; ++y)
            for (int x = 1; x < n - 1; ++x) {
                const float d = sqrtf((x - c) * (x - c) + (y - c) * (y - c) + (z - c) * (z - c));
                if (fabsf(d - r) < 11.0f) { band[((long long)z * n + y) * n + x] = 1; ++count; }
            }
    std::vector<int> list;
    const int nbx = n / BX, nby = n / BY, nbz = n / BZ;
    for (int bz = 0; bz < nbz; ++bz)
        for (int by = 0; by < nby; ++by)
            for (int bx = 0; bx < nbx; ++bx) {
                // bricks on a face of the array are left out (their shell would leave it): the probe's band does not reach them
                bool any = false;
                for (int z = 0; z < BZ && !any; ++z)
                    for (int y = 0; y < BY && !any; ++y)
                        for (int x = 0; x < BX && !any; ++x)
                            any = band[((long long)(bz * BZ + z) * n + by * BY + y) * n + bx * BX + x] != 0;
                if (any && bz > 0 && by > 0 && bx > 0 && bz < nbz - 1 && by < nby - 1 && bx < nbx - 1) list.push_back((bz * nby + by) * nbx + bx);
            }
    vf4 *state, *out;
    float* canonical;
    int* d_list;
    unsigned char* d_band;
    hipMalloc(&state, N * 16); hipMalloc(&out, N * 16); hipMalloc(&canonical, N * 4);
    hipMalloc(&d_list, list.size() * 4); hipMalloc(&d_band, N);
    hipMemset(state, 0, N * 16); hipMemset(out, 0, N * 16); hipMemset(canonical, 0, N * 4);
    hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_band, band.data(), N, hipMemcpyHostToDevice);
    printf("%d^3: %u band voxels in %zu bricks (%.0f per brick), %d workgroups per CU\n", n, count, list.size(),
           (double)count / list.size(), per_cu);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kernel) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i)
                hipLaunchKernelGGL(kernel, dim3(256 * per_cu), dim3(256), 0, 0, i % 2 ? out : state, canonical,
                                   i % 2 ? state : out, d_list, (unsigned)list.size(), d_band, n, n, n);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms / 20 < best ? ms / 20 : best;
        }
        printf("%-44s %.2f us per launch\n", name, best * 1e3);
    };
    run("bricks, shell staged then used", bricks<false>);
    run("bricks, next shell fetched during the units", bricks<true>);
    return 0;
}
