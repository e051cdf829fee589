// What would the fused list kernel's ACCESS PATTERN cost if a wave staged its neighbourhoods through LDS instead of
// loading 18 float4 taps per voxel through the vector L1?  (tools/probe/list_taps.hip is the pattern the kernel has now.)
//
// A wave owns a BOX of BX x BY x BZ = 64 voxels (one lane per voxel, a 64-bit mask says which of them are band voxels).
// Per box: the box plus a one-voxel shell -- (BX+2)(BY+2)(BZ+2) float4 -- is fetched with ceil(V / 64) coalesced 16-byte
// wave-loads whose per-lane offsets are constants of the kernel (computed once per wave), written to the wave's PRIVATE
// LDS image (no workgroup barrier anywhere), and the 19 taps are ds_read_b128 at compile-time immediate offsets from a
// per-lane base that is constant too.  Software pipeline: the shell of box n + 1 is in flight (in registers) while box n's
// taps are read and summed.  No arithmetic beyond the sum: this is the bare pattern.
//   vector-memory instructions per 64 voxels: ceil(V / 64) + canonical + store   (list walk: 18 + own + canonical + list + store)
// hipcc -O3 --offload-arch=gfx950 tools/probe/box_taps.hip -o tools/probe/bin/box_taps && tools/probe/bin/box_taps [n]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));

struct Box {
    int origin;  // voxel index of the box's lowest corner (x0, y0, z0)
    int pad;
    unsigned long long mask;  // lane l = (lz * BY + ly) * BX + lx
};

template <int BX, int BY, int BZ, int NT>
__global__ __launch_bounds__(1024) void walk(const vf4* __restrict__ state, const float* __restrict__ canonical,
                                             vf4* __restrict__ out, const Box* __restrict__ boxes, unsigned count, int nx,
                                             int ny) {
    constexpr int PX = BX + 2, PY = BY + 2, PZ = BZ + 2, V = PX * PY * PZ, LOADS = (V + 63) / 64;
    extern __shared__ vf4 lds[];
    const unsigned wave_in_block = threadIdx.x / 64, lane = threadIdx.x & 63;
    vf4* image = lds + wave_in_block * (LOADS * 64);
    const int sy = nx, sz = nx * ny;
    // constants of the lane: where its staging loads read (relative to the shell's lowest corner) ...
    int goff[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        int k = (int)lane + 64 * j;
        k = k < V ? k : V - 1;
        const int s = k / (PY * PX), r = (k / PX) % PY, c = k % PX;
        goff[j] = s * sz + r * sy + c;
    }
    // ... and where its own voxel sits in the image
    const int lx = lane % BX, ly = (lane / BX) % BY, lz = lane / (BX * BY);
    const int centre = ((lz + 1) * PY + (ly + 1)) * PX + lx + 1;
    const int voxel_off = lz * sz + ly * sy + lx;
    // box headers are fetched two boxes ahead, the shell one box ahead (as the list walk does with entries and states).
    // XCD-aware dealing, as the fused kernel's list walk: XCD k (workgroups k, k + 8, ...) owns the k-th eighth of the
    // boxes (a contiguous z-range -> its own L2) and its workgroups sweep through that eighth side by side
    const unsigned xcd = blockIdx.x % 8, q = blockIdx.x / 8, per_xcd = gridDim.x / 8;
    const unsigned lo = (unsigned)((unsigned long long)count * xcd / 8), hi = (unsigned)((unsigned long long)count * (xcd + 1) / 8);
    const unsigned waves = per_xcd * 16;
    unsigned u = lo + q * 16 + wave_in_block;
    if (u >= hi) return;
    const unsigned lastb = hi - 1;
    count = hi;
    Box b = boxes[u];
    Box b1 = boxes[min(u + waves, lastb)];
    vf4 staged[LOADS];
    {
        const int corner = b.origin - 1 - sy - sz;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) staged[j] = state[corner + goff[j]];
    }
    float sink = 0.0f;
    while (u < count) {
        const Box b2 = boxes[min(u + 2 * waves, lastb)];
        const int i = b.origin + voxel_off;
        const bool mine = (b.mask >> lane) & 1ull;
        const float cn = canonical[i];  // before the staging loads: vmcnt counts in order
        __builtin_amdgcn_sched_barrier(0);
        // the staged shell -> the wave's image (the previous box's tap reads are done: they were consumed into acc)
#pragma unroll
        for (int j = 0; j < LOADS; ++j) image[lane + 64 * j] = staged[j];
        // next box's shell in flight while this one is read (the last round re-reads the last box: exact vmcnt waits)
        {
            const int corner = b1.origin - 1 - sy - sz;
#pragma unroll
            for (int j = 0; j < LOADS; ++j) staged[j] = state[corner + goff[j]];
        }
        __builtin_amdgcn_sched_barrier(0);
        vf4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nzr = (dx != 0) + (dy != 0) + (dz != 0);
                    if (nzr > 2 || (NT == 7 && nzr > 1) || (NT == 1 && nzr > 0)) continue;
                    acc += image[centre + (dz * PY + dy) * PX + dx];
                }
        acc.y += cn;
        sink += cn;  // used outside the masked store: the compiler must not sink the load into it
        if (mine) out[i] = acc;
        b = b1;
        b1 = b2;
        u += waves;
    }
    if (sink == 1234.5f) out[0] = staged[0];
}

template <int BX, int BY, int BZ>
void measure(int n, const std::vector<int>& voxels, const vf4* state, const float* canonical, vf4* out, int blocks) {
    constexpr int PX = BX + 2, PY = BY + 2, PZ = BZ + 2, V = PX * PY * PZ, LOADS = (V + 63) / 64;
    std::map<long long, unsigned long long> m;
    for (int i : voxels) {
        const int x = i % n, y = (i / n) % n, z = i / (n * n);
        const long long key = ((long long)(z / BZ) * (n / BY) + y / BY) * (n / BX) + x / BX;
        m[key] |= 1ull << (((z % BZ) * BY + y % BY) * BX + x % BX);
    }
    std::vector<Box> host;
    for (auto& kv : m) {
        const long long key = kv.first;
        const int bx = (int)(key % (n / BX)), by = (int)((key / (n / BX)) % (n / BY)), bz = (int)(key / ((long long)(n / BX) * (n / BY)));
        host.push_back({((bz * BZ) * n + by * BY) * n + bx * BX, 0, kv.second});
    }
    const unsigned count = (unsigned)host.size();
    Box* boxes;
    hipMalloc(&boxes, count * sizeof(Box));
    hipMemcpy(boxes, host.data(), count * sizeof(Box), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds_bytes = 16ull * LOADS * 64 * 16;
    auto run = [&](const char* name, auto kernel) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i)
                hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds_bytes, 0, i % 2 ? out : state, canonical,
                                   i % 2 ? const_cast<vf4*>(state) : out, boxes, count, n, n);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms / 20 < best ? ms / 20 : best;
        }
        printf("  %-34s %7.2f us per launch\n", name, best * 1e3);
    };
    printf("boxes %2d x %d x %d: %u boxes, fill %.1f %%, shell %d float4 = %d wave-loads per box, LDS %zu KB per CU\n", BX, BY,
           BZ, count, 100.0 * voxels.size() / (64.0 * count), V, LOADS, lds_bytes / 1024);
    run("19 LDS taps + canonical + store", walk<BX, BY, BZ, 19>);
    run(" 7 LDS taps + canonical + store", walk<BX, BY, BZ, 7>);
    run(" 1 LDS tap  + canonical + store", walk<BX, BY, BZ, 1>);
    hipFree(boxes);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    // workgroups of 16 waves: 256 = one per CU (4 waves per SIMD, the fused kernel's occupancy at 128 VGPRs), 512 = two
    // per CU (8 waves per SIMD: what a kernel of <= 64 VGPRs could have)
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<int> voxels;
    const float r = 0.3f * n, c = n / 2.0f;
    for (int z = 4; z < n - 4; ++z)
        for (int y = 4; y < n - 4; ++y)
            for (int x = 4; x < n - 4; ++x) {
                const float d = sqrtf((x - c) * (x - c) + (y - c) * (y - c) + (z - c) * (z - c));
                if (fabsf(d - r) < 11.0f) voxels.push_back((z * n + y) * n + x);
            }
    const long long N = (long long)n * n * n;
    vf4 *state, *out;
    float* canonical;
    hipMalloc(&state, N * 16);
    hipMalloc(&out, N * 16);
    hipMalloc(&canonical, N * 4);
    hipMemset(state, 0, N * 16);
    hipMemset(out, 0, N * 16);
    hipMemset(canonical, 0, N * 4);
    printf("%d^3, %zu band voxels (%zu wave-units of a compacted list), %d workgroups of 16 waves\n", n, voxels.size(), (voxels.size() + 63) / 64, blocks);
    measure<16, 4, 1>(n, voxels, state, canonical, out, blocks);
    measure<8, 8, 1>(n, voxels, state, canonical, out, blocks);
    measure<16, 2, 2>(n, voxels, state, canonical, out, blocks);
    measure<8, 4, 2>(n, voxels, state, canonical, out, blocks);
    measure<4, 4, 4>(n, voxels, state, canonical, out, blocks);
    measure<32, 2, 1>(n, voxels, state, canonical, out, blocks);
    return 0;
}
