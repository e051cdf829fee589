// What does the ACCESS PATTERN of the fused list kernel reach on its own (no arithmetic)?  A band list of a sphere shell
// (the 256^3 sphere pair's band: ~1.65 M voxels in short x-runs), one CU-sized workgroup per CU, every wave takes 64
// consecutive entries at a time: list entry -> NT float4 taps of the state around the voxel + canonical -> one float4
// store.  NT = 19 (the kernel's 3^3 neighbourhood without corners), 7 (faces), 1 (centre); also the 19 taps as dwords.
// hipcc -O3 --offload-arch=gfx950 tools/probe/list_taps.hip -o /tmp/list_taps && /tmp/list_taps [n]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int NT, bool DWORD, bool CONTIG = false, bool XCD = false>
__global__ __launch_bounds__(1024) void walk(const vf4* __restrict__ state, const float* __restrict__ canonical,
                                             vf4* __restrict__ out, const int* __restrict__ list, unsigned count, int nx,
                                             int ny) {
    const unsigned units = (count + 63) / 64, waves = gridDim.x * 16;
    const unsigned wave = blockIdx.x * 16 + threadIdx.x / 64, lane = threadIdx.x & 63;
    const int sy = nx, sz = nx * ny;
    // CONTIG: a workgroup owns a contiguous range of the list and its 16 waves go through it side by side
    const unsigned per_block = (units + gridDim.x - 1) / gridDim.x;
    unsigned first = CONTIG ? blockIdx.x * per_block + threadIdx.x / 64 : wave;
    unsigned last = CONTIG ? min(units, (blockIdx.x + 1) * per_block) : units;
    unsigned step = CONTIG ? 16u : waves;
    if (XCD) {  // the fused kernel's dealing: XCD k owns the k-th eighth of the list, its workgroups sweep it side by side
        const unsigned xcd = blockIdx.x % 8, q = blockIdx.x / 8;
        first = (unsigned)((unsigned long long)units * xcd / 8) + q * 16 + threadIdx.x / 64;
        last = (unsigned)((unsigned long long)units * (xcd + 1) / 8);
        step = gridDim.x / 8 * 16;
    }
    for (unsigned u = first; u < last; u += step) {
        const unsigned k = u * 64 + lane;
        const int i = list[k < count ? k : count - 1];
        vf4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        int taps = 0;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nzr = (dx != 0) + (dy != 0) + (dz != 0);
                    if (nzr > 2 || (NT == 7 && nzr > 1) || (NT == 1 && nzr > 0)) continue;
                    const int j = i + dx + dy * sy + dz * sz;
                    if (DWORD) acc.x += reinterpret_cast<const float*>(state)[4ll * j];
                    else acc += state[j];
                    ++taps;
                }
        acc.y += canonical[i];
        if (k < count) out[i] = acc;
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    // list order: patches of PZ slices x PY rows (x whole), the patches in z-major order, inside a patch z, y, x ascending
    // (PZ = PY = 1: the plain ascending list of the package)
    const int PZ = argc > 2 ? atoi(argv[2]) : 1, PY = argc > 3 ? atoi(argv[3]) : 1;
    // STRIPS > 0: strip-major order instead -- the rows are cut into STRIPS bands of n / STRIPS rows, every band is swept
    // through ALL slices before the next one starts (inside a band z, y, x ascending): the three uses of a state line
    // (as the z + 1, the centre and the z - 1 row) then lie a band-slice apart instead of a whole slice
    const int STRIPS = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<int> host;
    const float r = 0.3f * n, c = n / 2.0f;
    auto in_band = [&](int x, int y, int z) {
        const float d = sqrtf((x - c) * (x - c) + (y - c) * (y - c) + (z - c) * (z - c));
        return fabsf(d - r) < 11.0f;
    };
    if (STRIPS > 0) {
        const int rows = (n + STRIPS - 1) / STRIPS;
        for (int s0 = 0; s0 < n; s0 += rows)
            for (int z = 1; z < n - 1; ++z)
                for (int y = s0 > 1 ? s0 : 1; y < s0 + rows && y < n - 1; ++y)
                    for (int x = 1; x < n - 1; ++x)
                        if (in_band(x, y, z)) host.push_back((z * n + y) * n + x);
    } else
    for (int z0 = 0; z0 < n; z0 += PZ)
        for (int y0 = 0; y0 < n; y0 += PY)
            for (int z = z0; z < z0 + PZ && z < n - 1; ++z)
                for (int y = y0; y < y0 + PY && y < n - 1; ++y)
                    for (int x = 1; x < n - 1; ++x) {
                        if (z < 1 || y < 1) continue;
                        if (in_band(x, y, z)) host.push_back((z * n + y) * n + x);
                    }
    const unsigned count = (unsigned)host.size();
    const long long N = (long long)n * n * n;
    vf4 *state, *out;
    float* canonical;
    int* list;
    hipMalloc(&state, N * 16);
    hipMalloc(&out, N * 16);
    hipMalloc(&canonical, N * 4);
    hipMalloc(&list, count * 4ll);
    hipMemset(state, 0, N * 16);
    hipMemset(out, 0, N * 16);
    hipMemset(canonical, 0, N * 4);
    hipMemcpy(list, host.data(), count * 4ll, hipMemcpyHostToDevice);
    printf("%d^3, band list %u entries (%u wave-units), patches of %d slices x %d rows, %d strips\n", n, count, (count + 63) / 64, PZ, PY, STRIPS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto kernel, double bytes_per_voxel) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) {  // ping-pong like the iterations
                hipLaunchKernelGGL(kernel, dim3(256), dim3(1024), 0, 0, i % 2 ? out : state, canonical, i % 2 ? state : out,
                                   list, count, n, n);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms / 20 < best ? ms / 20 : best;
        }
        printf("%-40s %.2f us per launch  (%.0f GB/s of %.0f B per voxel through the vector L1)\n", name, best * 1e3,
               bytes_per_voxel * count / best / 1e6, bytes_per_voxel);
    };
    run("19 float4 taps + canonical + store", walk<19, false>, 19 * 16 + 4 + 16 + 4);
    run("19 float4 taps, contiguous range per CU", walk<19, false, true>, 19 * 16 + 4 + 16 + 4);
    run("19 float4 taps, an eighth per XCD", walk<19, false, false, true>, 19 * 16 + 4 + 16 + 4);
    run(" 7 float4 taps, an eighth per XCD", walk<7, false, false, true>, 7 * 16 + 4 + 16 + 4);
    run(" 1 float4 tap,  an eighth per XCD", walk<1, false, false, true>, 1 * 16 + 4 + 16 + 4);
    run(" 7 float4 taps + canonical + store", walk<7, false>, 7 * 16 + 4 + 16 + 4);
    run(" 1 float4 tap  + canonical + store", walk<1, false>, 1 * 16 + 4 + 16 + 4);
    run("19 dword  taps + canonical + store", walk<19, true>, 19 * 4 + 4 + 16 + 4);
    return 0;
}
