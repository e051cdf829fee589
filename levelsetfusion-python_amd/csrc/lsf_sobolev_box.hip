// The SobolevFusion iteration behind its first filter pass, walking BOXES instead of list entries (3-D, whole volumes):
// y pass + z pass + update + re-warp in ONE launch.  On lists these are two launches (convolve_list4_kernel and
// sobolev_state_update_kernel, lsf_sobolev_state.hip) that between them issue 7 + 7 + 1 + 8 wave-loads per 64 voxels, write
// the y-filtered gradient and read it back seven times.  Here a wave owns a box of 4 x 4 x 4 band voxels (lsf_band_box of
// the LSF_BAND_ALL subset) and copies by LDS-DMA, straight into its own LDS image,
//   * the x-filtered gradient of the box's filter footprint: 4 (x) by 4 + 2c (y) by 4 + 2c (z) float4 (c = taps / 2;
//     400 of them for seven taps: seven wave-loads) -- out of a gradient buffer laid out in BRICKS of 4 x 4 x 4 voxels
//     (lsf_sobolev_state_gradient_x writes it so): the footprint is the box's own brick and slabs of its 8 y / z
//     neighbours, 23 contiguous pieces of 192 B to 1 KB instead of 100 rows of 64 B a row's pitch (4 KB at 256^3) apart.
//     The kernel is bound by the NUMBER of cache-line requests a box makes, not by their bytes,
//   * the state's 6 x 6 x 6 shell of the box (four wave-loads) for the re-warp's cell,
// runs the y pass for the 4 x 4 x (4 + 2c) voxels the z pass will read (their results stay in LDS), the z pass for its own
// 64, and the update + re-warp of sobolev_state_update_kernel.  Every tap is an LDS read at a compile-time offset from a
// per-lane base.  The filter is zero-preserving per pass (math_utils/convolution.py:118,123,127): the verdicts of the raw
// gradient travel as bits in the fourth component (lsf_sobolev_state.hip), and bit 8 there says "this voxel is listed" --
// an unlisted voxel's y-filtered value is the zero the list path never overwrites, not the sum of its neighbours.
// The arithmetic is filtered_at's and the update kernel's, tap for tap: the same bits.
// Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:208-236, math_utils/convolution.py:94-132,
// nonrigid_opt/field_warping.py:112-151.
#include "lsf_slavcheva_state_taps.h"

using namespace lsf;
using namespace lsf::slav;

namespace {

constexpr int kBoxEdge = 4, kShellEdge = kBoxEdge + 2, kShell = kShellEdge * kShellEdge * kShellEdge;  // 216
constexpr int kShellLoads = (kShell + kWave - 1) / kWave;                                                // 4
// Waves per workgroup = per CU: a wave stages, waits and computes in turn (no second image to stage into while it
// computes), so what hides the staging latency is the number of waves -- i.e. of images the CU's LDS holds.  An image is kept
// small for that: the y results overwrite the footprint's planes they were computed from (those are dead by then), the
// shell is staged as the 216 LIVE values only (four bytes per slot): 9.2 KB per wave with seven taps, sixteen waves per CU
// (8 waves of 15.3 KB images -- float4 shell, separate y buffer -- took 49 us per 256^3 launch against 45.6 for the two list
// kernels this replaces).
#ifndef LSF_SOB_PROBE
#define LSF_SOB_PROBE 0  // measurements only: 1: no y pass, 2: the z pass reads its centre tap seven times, 4: no footprint staged, 8: a made-up gradient, 16: nothing stored, 32: no shell staged
#endif
#ifndef LSF_SOBOLEV_BOX_WAVES
#define LSF_SOBOLEV_BOX_WAVES 16
#endif
constexpr int kSobWaves = LSF_SOBOLEV_BOX_WAVES;

#ifdef LSF_SOB_TRACE  // measurement builds only (tools/sobolev_trace.py): shader-clock totals per wave of the box walk
__device__ unsigned long long* g_sob_trace = nullptr;  // [block][wave 16][8]: rounds, cycles in stage issue / wait / y / z+update / stores, total, entry stamp
#define LSF_SOB_T(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define LSF_SOB_T(var) do {} while (0)
#endif

template <int NT>
struct Footprint {
    static constexpr int c = NT / 2;
    static constexpr int edge = kBoxEdge + 2 * c;                                // y and z extent of the staged gradient
    static constexpr int slots = kBoxEdge * edge * edge;                         // 400 for seven taps
    static constexpr int loads = (slots + kWave - 1) / kWave;                    // 7
    static constexpr int y_rounds = (kBoxEdge * kBoxEdge * edge + kWave - 1) / kWave;  // 3: 160 voxels get a y pass
    // float4 slots of a wave's image: the footprint (+ the reach of the last y round's idle lanes), then the shell's live
    // values (kShellLoads * kWave floats).  The y results of plane dz go where its first 16 footprint slots were.
    static constexpr int gx_slots = loads * kWave > (y_rounds * 4 * edge + 2 * c) * 4 + 4 ? loads * kWave
                                                                                          : (y_rounds * 4 * edge + 2 * c) * 4 + 4;
    static constexpr int plane = 4 * edge;  // slots of one z-plane of the footprint
    static constexpr int shell_at = gx_slots, image = shell_at + kShellLoads * kWave / 4;
    // as many waves as images fit the CU's 160 KB (less what the workgroup declares statically), whole waves per SIMD
    static constexpr int fit = (160 * 1024 - 8192) / (image * 16);
    static constexpr int waves = fit >= kSobWaves ? kSobWaves : fit / 4 * 4;  // 16 up to seven taps, 12 for nine
    static constexpr int threads = waves * kWave;
};

// the re-warp cell out of the shell image: rewarp_from_taps (lsf_slavcheva_state_taps.h) with all eight live values read
// at computed offsets.  Lanes that are not listed vote "fine".  Active lanes must have their 3^3 neighbourhood inside the
// array (the shell's slots beyond a face hold clamped addresses' data) and stand at coordinates >= 2 (NearFar).
__device__ inline bool rewarp_from_shell(const float* __restrict__ shell, int centre, const Grid& g, int x, int y, int z,
                                         const float (&wv)[3], bool active, Rewarp& rw) {
    const NearFar ax((float)x, wv[0]), ay((float)y, wv[1]), az((float)z, wv[2]);
    const unsigned lowest = min(min((unsigned)x, (unsigned)y), (unsigned)z);
    const bool inside = x < g.nx - 1 && y < g.ny - 1 && z < g.nz - 1;
    if (!__all(!active || (ax.near && ay.near && az.near && lowest >= 2u && inside))) return false;
    const int ox = ax.below ? -1 : 1, oy = ay.below ? -kShellEdge : kShellEdge;
    const int oz = az.below ? -kShellEdge * kShellEdge : kShellEdge * kShellEdge;
    const float t000 = shell[centre];
    const float t100 = shell[centre + oz], t010 = shell[centre + oy], t001 = shell[centre + ox];
    const float t110 = shell[centre + oz + oy], t101 = shell[centre + oz + ox], t011 = shell[centre + oy + ox];
    rw.corner = shell[centre + oz + oy + ox];
    const float c00 = t000 * az.wn + t100 * az.wf;
    const float c10 = t010 * az.wn + t110 * az.wf;
    const float c01 = t001 * az.wn + t101 * az.wf;
    const float iy0 = c00 * ay.wn + c10 * ay.wf;
    rw.P = t011 * az.wn;
    rw.Q = c01 * ay.wn;
    rw.R = iy0 * ax.wn;
    rw.wfz = az.wf;
    rw.wfy = ay.wf;
    rw.wfx = ax.wf;
    rw.lerp = true;
    return true;
}

template <int NT, bool FMA>
__global__ __launch_bounds__(Footprint<NT>::threads) void sobolev_state_box_kernel(const vf4* __restrict__ gx,
                                                                        const vf4* __restrict__ state_in,
                                                                        vf4* __restrict__ state_out, vf4* __restrict__ g_out,
                                                                        Grid g, Params p, TapsN<NT> taps, lsf_gate gate,
                                                                        lsf_iteration_record* record,
                                                                        const lsf_band_box* __restrict__ boxes,
                                                                        unsigned box_count) {
    using F = Footprint<NT>;
    constexpr int c = F::c, E = F::edge;
    if (gate_closed(gate)) return;
    extern __shared__ vf4 lds[];  // [wave][F::image]
    __shared__ unsigned s_next_unit;
    __shared__ int s_goff[kWave][F::loads + kShellLoads];  // the lanes' staging offsets (the same for every wave)
    const unsigned waves = blockDim.x / kWave, wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    vf4* const image = lds + wave * F::image;
    float* const shell = reinterpret_cast<float*>(image + F::shell_at);
    const int sy = g.nx, sz = g.nx * g.ny;
    const int last_voxel = g.nz * sz - 1;
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < F::loads; ++j) {
            // footprint slot k = (dz * E + dy) * 4 + dx holds voxel (x0 + dx, y0 - c + dy, z0 - c + dz); the gradient lies in
            // bricks of 4 x 4 x 4 (lsf_sobolev_state_gradient_x, out_bricks): offset relative to the box's own brick
            int k = (int)lane + kWave * j;
            k = k < F::slots ? k : F::slots - 1;
            const int ry = (k / 4) % E - c, rz = k / (4 * E) - c;  // relative to the box's corner: -c .. 3 + c
            s_goff[lane][j] = (((rz >> 2) * (g.ny >> 2) + (ry >> 2)) * (g.nx >> 2)) * 64 + (((rz & 3) << 4) | ((ry & 3) << 2) | (k % 4));
        }
#pragma unroll
        for (int j = 0; j < kShellLoads; ++j) {
            int k = (int)lane + kWave * j;
            k = k < kShell ? k : kShell - 1;
            s_goff[lane][F::loads + j] = (k / (kShellEdge * kShellEdge)) * sz + ((k / kShellEdge) % kShellEdge) * sy + k % kShellEdge;
        }
    }
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    const int centre = ((lz + 1) * kShellEdge + ly + 1) * kShellEdge + lx + 1;  // the lane's voxel in the shell
    const int voxel_off = lz * sz + ly * sy + lx;
    // y pass, round r: voxel q = lane + 64 r of the 4 x 4 x E block = (dz = lz + 4 r, yy = ly, dx = lx); its centre tap in
    // the footprint: ((dz * E) + yy + c) * 4 + dx
    const int y_base = ((lz * E) + ly + c) * 4 + lx;

    unsigned long long best = 0ull;
#ifdef LSF_SOB_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, __builtin_amdgcn_s_memrealtime()};
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    const WaveWalk w = wave_list_walk(box_count * kWave, g.list_group);
    if (threadIdx.x == 0) s_next_unit = waves;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t state_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        state_out, 0, (int)((unsigned)(last_voxel + 1) * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        g_out ? g_out : state_out, 0, g_out ? (int)((unsigned)(last_voxel + 1) * 16u) : 0, 0x00020000);

    // a box's header is fetched one round ahead, by scalar loads (the unit number is wave-uniform and is made to look so):
    // nothing in a round waits for memory but the staged data
    auto header = [&](unsigned unit) { return boxes[unit < box_count ? unit : box_count - 1u]; };
    unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)w.unit(wave));
    lsf_band_box b = {0, 0, 0ull};
    if (box_count) b = header(u);
    while (u < w.x_end && u < box_count) {
        // the next box of this wave: an LDS atomic and its header's scalar loads, issued BEFORE the staging loads (the
        // compiler makes every LDS operation behind an LDS-DMA load wait for the whole vector-memory counter)
        unsigned next = 0u;
        if (lane == 0) next = atomicAdd(&s_next_unit, 1u);
        next = w.unit((unsigned)__builtin_amdgcn_readfirstlane((int)next));
        const lsf_band_box b_next = header(next);
        LSF_SOB_T(t0);
        int x0, y0, z0;
        decode_voxel(g, (unsigned)b.origin, x0, y0, z0);
        {
            const int corner = (((z0 >> 2) * (g.ny >> 2) + (y0 >> 2)) * (g.nx >> 2) + (x0 >> 2)) * 64;  // the box's brick
#pragma unroll
            for (int j = 0; j < (LSF_SOB_PROBE & 4 ? 0 : F::loads); ++j) {
                int v = corner + s_goff[lane][j];
                v = v < 0 ? 0 : (v > last_voxel ? last_voxel : v);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx + v),
                                                 (__attribute__((address_space(3))) void*)(image + j * kWave), 16, 0, 0);
            }
            const int shell_corner = b.origin - 1 - sy - sz;
#pragma unroll
            for (int j = 0; j < (LSF_SOB_PROBE & 32 ? 0 : kShellLoads); ++j) {
                int v = shell_corner + s_goff[lane][F::loads + j];
                v = v < 0 ? 0 : (v > last_voxel ? last_voxel : v);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(state_in + v),
                                                 (__attribute__((address_space(3))) void*)(shell + j * kWave), 4, 0, 0);
            }
        }
        LSF_SOB_T(t1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSF_SOB_T(t2);
        // a footprint that sticks out of the array along y or z: those taps count as zero (np.convolve(mode='same')); the
        // clamped addresses brought other voxels' data (wave-uniform test; boxes at the faces only)
        if (y0 < c || y0 + kBoxEdge + c > g.ny || z0 < c || z0 + kBoxEdge + c > g.nz) {
#pragma unroll
            for (int j = 0; j < F::loads; ++j) {
                const int k = (int)lane + kWave * j;
                const int yy = y0 - c + (k / 4) % E, zz = z0 - c + k / (4 * E);
                if (k < F::slots && (yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz)) {
                    const vf4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                    image[k] = zero;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // ---- y pass of the 4 x 4 x E voxels the z pass reads (filtered_at: float64 sums in tap order, one rounding) ----------
#pragma unroll
        for (int r = 0; r < (LSF_SOB_PROBE & 1 ? 0 : F::y_rounds); ++r) {
            const vf4* ctr = image + y_base + r * (4 * E * 4);
            vf4 v[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) v[j] = ctr[4 * (c - j)];
            double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[0] = mac<FMA>(acc[0], taps.k[j], (double)v[j].x);
                acc[1] = mac<FMA>(acc[1], taps.k[j], (double)v[j].y);
                acc[2] = mac<FMA>(acc[2], taps.k[j], (double)v[j].z);
            }
            const unsigned bits = __float_as_uint(v[c].w);
            const bool is_listed = (bits & 8u) != 0u;
            vf4 o;
            o.x = (!is_listed || (bits & 1u)) ? 0.0f : (float)acc[0];
            o.y = (!is_listed || (bits & 2u)) ? 0.0f : (float)acc[1];
            o.z = (!is_listed || (bits & 4u)) ? 0.0f : (float)acc[2];
            o.w = __uint_as_float(bits);
            // over the first slots of the planes this round read (every lane's taps are in registers by now)
            image[(lz + 4 * r) * F::plane + (lane & 15)] = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // other lanes' y results (one wave: LDS executes in order)
        LSF_SOB_T(t3);
        // ---- z pass of the lane's own voxel ------------------------------------------------------------------------------
        float gv[3] = {0.0f, 0.0f, 0.0f};
        if (!(LSF_SOB_PROBE & 8)) {
            const vf4* ctr = image + (lz + c) * F::plane + (lane & 15);
            vf4 v[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) v[j] = ctr[F::plane * (LSF_SOB_PROBE & 2 ? 0 : c - j)];
            double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[0] = mac<FMA>(acc[0], taps.k[j], (double)v[j].x);
                acc[1] = mac<FMA>(acc[1], taps.k[j], (double)v[j].y);
                acc[2] = mac<FMA>(acc[2], taps.k[j], (double)v[j].z);
            }
            const unsigned bits = __float_as_uint(v[c].w);
            gv[0] = (bits & 1u) ? 0.0f : (float)acc[0];
            gv[1] = (bits & 2u) ? 0.0f : (float)acc[1];
            gv[2] = (bits & 4u) ? 0.0f : (float)acc[2];
        }
        if (LSF_SOB_PROBE & 8) gv[0] = gv[1] = gv[2] = 1e-3f * (float)(lane & 7);
        // ---- update + re-warp (sobolev_state_update_kernel) ---------------------------------------------------------------
        const bool listed = ((b.mask >> lane) & 1ull) != 0ull;
        const int i = b.origin + voxel_off;
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        float wv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) wv[k] = listed ? (-gv[k]) * p.rate : 0.0f;
        const float len_w = vec_length<3>(wv);
        const float l = shell[centre];
        Rewarp rw;
        if (!rewarp_from_shell(shell, centre, g, x, y, z, wv, listed, rw)) {
            rw.lerp = false;
            rw.R = state_gather<3>(state_in, g, (float)x + wv[0], (float)y + wv[1], (float)(z + g.z_global_offset) + wv[2]);
        }
        if (wv[0] == 0.0f && wv[1] == 0.0f && wv[2] == 0.0f) {  // zero displacement: the gather returns live[p] bit for bit
            rw.lerp = false;
            rw.R = l;
        }
        float v = rw.value();
        if (1.0f - fabsf(v) < 1e-6f) {  // field_warping.py:138-141
            v = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v);
            wv[0] = wv[1] = wv[2] = 0.0f;
            if (p.zero_gradient_on_snap) gv[0] = gv[1] = gv[2] = 0.0f;
        }
        vf4 o, go;
        o.x = v; o.y = wv[0]; o.z = wv[1]; o.w = wv[2];
        go.x = gv[0]; go.y = gv[1]; go.z = gv[2]; go.w = 0.0f;
        // a lane that has nothing to store names an offset behind the buffer and the hardware drops it (range-checked raw
        // buffer stores; a null g_out is a buffer of zero bytes): no branch around the stores
        LSF_SOB_T(t4);
        const int offset = listed && !(LSF_SOB_PROBE & 16) ? i * 16 : (int)0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), state_rsrc, offset, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, go), g_rsrc, offset, 0, 0);
        const unsigned long long q = listed ? pack_max(len_w, linear_index(g, x, y, z)) : 0ull;
        best = q > best ? q : best;
        u = next;
        b = b_next;
#ifdef LSF_SOB_TRACE
        {
            const unsigned long long t5 = __builtin_readcyclecounter();
            tr[0] += 1; tr[1] += t1 - t0; tr[2] += t2 - t1; tr[3] += t3 - t2; tr[4] += t4 - t3; tr[5] += t5 - t4;
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LSF_SOB_TRACE
    if (g_sob_trace && lane == 0) {
        tr[6] = __builtin_readcyclecounter() - t_entry;
        unsigned long long* row = g_sob_trace + ((unsigned long long)blockIdx.x * 16 + wave) * 8;
        for (int k = 0; k < 8; ++k) row[k] = tr[k];
    }
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the unlisted voxels: zero update, smallest index
        const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
        best = q > best ? q : best;
    }
    const double sums[1] = {0.0};
    double* dst[1] = {nullptr};
    block_reduce_commit<0>(best, sums, record_max(record), dst);
}

struct SobBoxLaunch {
    unsigned blocks;
    hipStream_t s;
    const vf4* gx;
    const vf4* state_in;
    vf4* state_out;
    vf4* g_out;
    Grid g;
    Params p;
    lsf_gate gate;
    lsf_iteration_record* record;
    const lsf_band_box* boxes;
    unsigned box_count;
    const double* taps_host;
};

template <int NT, bool FMA>
int launch_sob_box_one(const SobBoxLaunch& a) {
    auto kernel = sobolev_state_box_kernel<NT, FMA>;
    constexpr size_t lds_bytes = (size_t)Footprint<NT>::waves * Footprint<NT>::image * sizeof(vf4);
    static_assert(lds_bytes + 8192 <= 160 * 1024, "a workgroup's images must fit the CU's LDS");
    static bool configured[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipGetLastError();
    if (!configured[dev]) {  // more than 64 KiB of dynamic LDS has to be asked for, once per device and instantiation
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return (int)hipGetLastError();
        configured[dev] = true;
    }
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = a.taps_host[j];
    hipLaunchKernelGGL(kernel, dim3(a.blocks), dim3(Footprint<NT>::threads), lds_bytes, a.s, a.gx, a.state_in, a.state_out, a.g_out, a.g,
                       a.p, taps, a.gate, a.record, a.boxes, a.box_count);
    return 0;
}

template <int NT>
int launch_sob_box(const SobBoxLaunch& a) {
    return taps_are_float32(a.taps_host, NT) ? launch_sob_box_one<NT, true>(a) : launch_sob_box_one<NT, false>(a);
}

inline unsigned sob_box_compute_units() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return (unsigned)cached[dev];
}

}  // namespace

#ifdef LSF_SOB_TRACE
extern "C" int lsf_debug_set_sobolev_trace(void* rows) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sob_trace), &rows, sizeof(rows));
}
#endif

extern "C" int lsf_sobolev_state_update_boxes(const float* in4, const float* state_in, float* state_out, float* g_out4,
                                              const lsf_grid* grid, const lsf_slavcheva_params* params,
                                              const double* taps_host, int32_t n_taps, const lsf_gate* gate,
                                              lsf_iteration_record* record, const lsf_band_box* boxes, int64_t box_count,
                                              void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in4 || !state_in || !state_out || state_out == state_in || g_out4 == in4 || !params || !record || !taps_host ||
        !boxes || box_count < 0 || box_count > 0x3ffffffll)
        return LSF_ERR_BAD_ARGUMENT;
    if (!(n_taps == 3 || n_taps == 5 || n_taps == 7 || n_taps == 9)) return LSF_ERR_KERNEL_TOO_LONG;
    // whole 3-D volumes of whole boxes whose float4 fields fit 32-bit voxel arithmetic
    if (grid->dims != 3 || grid->nx % kBoxEdge || grid->ny % kBoxEdge || grid->nz % kBoxEdge || grid->z_begin != 0 ||
        grid->z_end != grid->nz || (long long)grid->nx * grid->ny * grid->nz > 0x07ffffffll)
        return LSF_ERR_BAD_DIMS;
    Grid g = make_grid(grid, 4);
    g.fast_ok = g.plane * 16 < 0xffffffffll;
    g.wide_ok = true;
    Params p;
    p.lambda64 = params->isomorphic_enforcement_factor_f64;
    p.rate = params->rate;
    p.w_data = params->data_term_weight;
    p.w_smooth = params->smoothing_term_weight;
    p.w_level_set = params->level_set_term_weight;
    p.lambda32 = params->isomorphic_enforcement_factor;
    p.killing_c1 = params->killing_c1;
    p.zero_gradient_on_snap = params->zero_gradient_on_snap;
    // (an empty band still has the unlisted voxels' zero update to record: one workgroup)
    SobBoxLaunch a{cu_list_blocks((unsigned)box_count * kWave, sob_box_compute_units()), as_stream(stream),
                   reinterpret_cast<const vf4*>(in4), reinterpret_cast<const vf4*>(state_in),
                   reinterpret_cast<vf4*>(state_out), reinterpret_cast<vf4*>(g_out4), g, p, gate_or_open(gate), record, boxes,
                   (unsigned)box_count, taps_host};
    int status;
    switch (n_taps) {
        case 3: status = launch_sob_box<3>(a); break;
        case 5: status = launch_sob_box<5>(a); break;
        case 7: status = launch_sob_box<7>(a); break;
        default: status = launch_sob_box<9>(a); break;
    }
    return status ? status : launch_status();
}
