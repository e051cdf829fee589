// The fused warp-update kernel of lsf_slavcheva_state.hip walking BOXES instead of list entries (3-D, INTERIOR band voxels):
// a wave owns a box of 4 x 4 x 4 voxels -- one lane per voxel, a 64-bit mask says which of them are band voxels --, the box
// and its one-voxel shell (6 x 6 x 6 float4 = 216 of the state) are copied by FOUR coalesced 16-byte wave-loads straight
// into the wave's own LDS image (LDS-DMA: no destination registers, no workgroup barrier anywhere), and the 3^3
// neighbourhood of every lane -- the 19 taps of the stencils and the far corner of the re-warp cell -- is read from that
// image at compile-time offsets from a per-lane base that never changes.  Per 64 lanes: 4 vector-memory loads through the
// L1 instead of 20, ~60 cache-line accesses instead of ~530 (profiles/r05_pmc_l2_tcp.txt: the list walk keeps the L1 at 0.9
// line accesses per clock and its 64 B/clk data path at 73 %).  The price: the lanes of a box that are not band voxels idle
// (17-20 % on a narrow band).  The arithmetic is the list walk's, term for term (lsf_slavcheva_state_taps.h): the same bits.
// Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330 with data_term.py, smoothing_term.py,
// level_set_term.py and field_warping.warp_field_advanced (:112-151).
#include "lsf_slavcheva_state_taps.h"

using namespace lsf;
using namespace lsf::slav;

namespace {

constexpr int kBoxEdge = 4, kShellEdge = kBoxEdge + 2, kShell = kShellEdge * kShellEdge * kShellEdge;  // 216
constexpr int kShellLoads = (kShell + kWave - 1) / kWave;                                                // 4
// Waves per workgroup = per CU: 12, three per SIMD with up to 170 registers each (the kernel takes 145).  At 16 waves (128
// registers) it spills inside the loop -- a scratch reload is a vector-memory operation the LDS-DMA loads then queue behind:
// 124.8 against 119.8 us per 512^3 launch, 30.8 us both at 256^3.  Built to fit 128 registers without spilling -- the taps
// fetched in four groups that follow the arithmetic, the re-warp cell read from the image by seven computed offsets instead
// of selected among the taps' live components, the energy sums kept in LDS between rounds: 120 -> 128 registers, bit-identical
// -- it measured 31.5 us against 30.5 for the list walk at 256^3 and 125.7 against 144.4 at 512^3: no better than this form,
// with twice the code.  (variant builds: tools/build_variant.sh NAME - -DLSF_BOX_WAVES=16)
#ifndef LSF_BOX_WAVES
#define LSF_BOX_WAVES 12
#endif
constexpr int kBoxWaves = LSF_BOX_WAVES, kBoxThreads = kBoxWaves * kWave;
constexpr int kImage = kShellLoads * kWave + kWave / 4;  // float4 slots of one image: the shell (the last load's surplus
                                                         // lanes land behind it), then the box's 64 canonical values

// the 3^3 neighbourhood out of the wave's LDS image: tap (dx, dy, dz) at centre + (dz * 6 + dy) * 6 + dx
struct NbhStateLds : TapsBase<3> {
    __device__ inline void load(const vf4* __restrict__ image, int centre) {
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if ((dx != 0) + (dy != 0) + (dz != 0) > 2) continue;
                    this->t[dz + 1][dy + 1][dx + 1] = image[centre + (dz * kShellEdge + dy) * kShellEdge + dx];
                }
    }
};

// rewarp_from_taps (lsf_slavcheva_state_taps.h) with the far corner read from the image; lanes that are not band voxels of
// the box vote "fine" (their update is zero)
__device__ inline bool rewarp_from_image(const TapsBase<3>& n, const vf4* __restrict__ image, int centre, const Grid& g,
                                         int x, int y, int z, const float (&wv)[3], bool active, Rewarp& rw) {
    const NearFar ax((float)x, wv[0]), ay((float)(y + g.y_global_offset), wv[1]), az((float)(z + g.z_global_offset), wv[2]);
    const unsigned lowest = min(min((unsigned)x, (unsigned)y), (unsigned)z);
    if (!__all(!active || (ax.near && ay.near && az.near && lowest >= 2u))) return false;
    const int ci = centre + (ax.below ? -1 : 1) + (ay.below ? -kShellEdge : kShellEdge) +
                   (az.below ? -kShellEdge * kShellEdge : kShellEdge * kShellEdge);
    rw.corner = image[ci].x;
    auto L = [&](int dz, int dy, int dx) { return n.t[dz + 1][dy + 1][dx + 1].x; };
    auto pick2 = [&](bool b, float m, float p) { return b ? m : p; };
    const float t000 = L(0, 0, 0);
    const float t100 = pick2(az.below, L(-1, 0, 0), L(1, 0, 0));
    const float t010 = pick2(ay.below, L(0, -1, 0), L(0, 1, 0));
    const float t001 = pick2(ax.below, L(0, 0, -1), L(0, 0, 1));
    const float t110 = pick2(az.below, pick2(ay.below, L(-1, -1, 0), L(-1, 1, 0)), pick2(ay.below, L(1, -1, 0), L(1, 1, 0)));
    const float t101 = pick2(az.below, pick2(ax.below, L(-1, 0, -1), L(-1, 0, 1)), pick2(ax.below, L(1, 0, -1), L(1, 0, 1)));
    const float t011 = pick2(ay.below, pick2(ax.below, L(0, -1, -1), L(0, -1, 1)), pick2(ax.below, L(0, 1, -1), L(0, 1, 1)));
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

template <int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__global__ __launch_bounds__(kBoxThreads) __attribute__((amdgpu_waves_per_eu(kBoxWaves / 4, kBoxWaves / 4)))
void slavcheva_state_box_kernel(const vf4* __restrict__ state_in, const float* __restrict__ canonical_boxed,
                                vf4* __restrict__ state_out, Grid g, Params p, lsf_gate gate, lsf_iteration_record* record,
                                const lsf_band_box* __restrict__ boxes, unsigned box_count) {
    g.y_global_offset = 0;  // whole volumes and z-slabs: every row is a row of the volume, every row's energies count
    g.ny_global = g.ny;
    g.y_cut = 0;
    if (gate_closed(gate)) return;
    extern __shared__ vf4 lds[];  // [wave][2 images][kImage]
    __shared__ unsigned s_next_unit;
    __shared__ int s_goff[kWave][kShellLoads];  // the lanes' staging offsets (the same for every wave)
    const unsigned waves = blockDim.x / kWave, wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    vf4* const images = lds + wave * (2 * kImage);
    const int sy = g.nx, sz = g.nx * g.ny;
    const int last_voxel = g.nz * sz - 1;
    // constants of the lane: where its four staging loads read, relative to the shell's lowest corner -- kept in LDS and
    // fetched where a shell is staged (four registers that would otherwise stay live through the arithmetic) ...
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < kShellLoads; ++j) {
            int k = (int)lane + kWave * j;
            k = k < kShell ? k : kShell - 1;
            s_goff[lane][j] = (k / (kShellEdge * kShellEdge)) * sz + ((k / kShellEdge) % kShellEdge) * sy + k % kShellEdge;
        }
    }
    // ... and where its own voxel sits: in the box, in the image
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    const int centre = ((lz + 1) * kShellEdge + ly + 1) * kShellEdge + lx + 1;
    const int voxel_off = lz * sz + ly * sy + lx;

    unsigned long long best = 0ull;
    double en[3] = {0.0, 0.0, 0.0};
    const WaveWalk w = wave_list_walk(box_count * kWave, g.list_group);
    if (threadIdx.x == 0) s_next_unit = 2u * waves;
    __syncthreads();
    auto grab = [&]() {
        unsigned v = 0u;
        if (lane == 0) v = atomicAdd(&s_next_unit, 1u);
        return w.unit((unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    auto header = [&](unsigned unit, bool& listed) {  // wave-uniform: scalar loads
        listed = unit < w.x_end && unit < box_count;
        return boxes[unit < box_count ? unit : box_count - 1u];
    };
    // the shell of a box into an image: four LDS-DMA wave-loads; addresses clamped into the array (the shell of a box on a
    // face of the array sticks out: those slots only ever serve lanes that are not INTERIOR band voxels)
    auto stage = [&](int origin, unsigned unit, vf4* image) {
        // (the canonical values of the box travel the same way, four bytes per lane: no ordinary vector load is left in the
        // loop, so nothing makes the compiler wait for the vector-memory counter behind the LDS-DMA loads.  They come from
        // a copy gathered box by box once per call, lsf_band_boxes_canonical: 256 contiguous bytes, two cache lines, where the
        // box's sixteen rows of four floats in the [z][y][x] array are sixteen lines and, beyond the Infinity Cache,
        // sixteen 64-byte sectors of HBM traffic for 256 bytes: 119.0 -> 112.2 us per 512^3 launch)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(
                                             canonical_boxed + (size_t)(unit < box_count ? unit : box_count - 1u) * kWave + lane),
                                         (__attribute__((address_space(3))) void*)(image + kShellLoads * kWave), 4, 0, 0);
        const int corner = origin - 1 - sy - sz;
        int goff[kShellLoads];
#pragma unroll
        for (int j = 0; j < kShellLoads; ++j) goff[j] = s_goff[lane][j];
#pragma unroll
        for (int j = 0; j < kShellLoads; ++j) {
            int v = corner + goff[j];
            v = v < 0 ? 0 : (v > last_voxel ? last_voxel : v);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(state_in + v),
                                             (__attribute__((address_space(3))) void*)(image + j * kWave), 16, 0, 0);
        }
    };
    // second half of a voxel: the re-warped live value and the snap of field_warping.py:138-141 -- at once, the far corner
    // comes out of the image -- and, one round later (behind the wait at the top of the loop), the store
    auto new_state = [&](const Deferred& d) {
        float v = d.rw.value();
        float wv[3] = {d.wv[0], d.wv[1], d.wv[2]};
        if (1.0f - fabsf(v) < 1e-6f) {
            v = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v);
            wv[0] = wv[1] = wv[2] = 0.0f;
        }
        vf4 o;
        o.x = v; o.y = wv[0]; o.z = wv[1]; o.w = wv[2];
        return o;
    };
    // a lane that has nothing to store names an offset behind the buffer and the hardware drops it (raw buffer store,
    // range-checked): no branch around the store.  (Measured and rejected: the store issued one round LATE, right behind the
    // wait at the top of the loop, so that it has a whole round to complete before the next wait -- 31.9 against 30.9 us at
    // 256^3 relative to the list walk on the same box: five more registers live across the arithmetic cost more than the wait.)
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        state_out, 0, (int)((unsigned)(last_voxel + 1) * 16u), 0x00020000);
    auto store = [&](const vf4& o, int i) {
        const int offset = i < 0 ? (int)0xfffffff0u : i * 16;
        if (g.list_store_nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), out_rsrc, offset, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), out_rsrc, offset, 0, 0);
    };

    unsigned u = box_count ? w.unit(wave) : w.x_end, u1 = w.unit(wave + waves);
    bool in0 = false, in1 = false, in2 = false;
    lsf_band_box b0 = {0, 0, 0ull}, b1 = b0;
    int parity = 0;
    if (u < w.x_end) {
        b0 = header(u, in0);
        b1 = header(u1, in1);
        stage(b0.origin, u, images);
    }
    while (u < w.x_end) {
        // what the previous round issued has to have landed: this box's shell and canonical values (LDS-DMA counts as vector
        // memory; the compiler makes every LDS read behind an LDS-DMA wait for the whole counter anyway, the previous box's
        // store included)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned u2 = grab();
        const lsf_band_box b2 = header(u2, in2);
        const vf4* image = images + parity * kImage;
        const float cn = reinterpret_cast<const float*>(image + kShellLoads * kWave)[lane];
        stage(b1.origin, u1, images + (parity ^ 1) * kImage);
        __builtin_amdgcn_sched_barrier(0);
        NbhStateLds n;
        n.load(image, centre);
        const unsigned i = (unsigned)(b0.origin + voxel_off);
        const bool listed = in0 && ((b0.mask >> lane) & 1ull);
        int x0, y0, z0;
        decode_voxel(g, (unsigned)b0.origin, x0, y0, z0);  // wave-uniform: scalar arithmetic
        const int x = x0 + lx, y = y0 + ly, z = z0 + lz;
        Deferred d;
        d.i = listed ? (int)i : -1;
        const float l = n.t[1][1][1].x;
        const bool in_band = listed && !(fabsf(l) == 1.0f && fabsf(cn) == 1.0f);
        float gv[3] = {0.0f, 0.0f, 0.0f};
        double e[3] = {0.0, 0.0, 0.0};
        band_voxel_gradient_taps<SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, e);
        const bool counted = in_band && z >= g.e_begin && z < g.e_end;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d.wv[c] = in_band ? (-gv[c]) * p.rate : 0.0f;
            en[c] += counted ? e[c] : 0.0;
        }
        const float len = vec_length<3>(d.wv);
        const bool moved = !(d.wv[0] == 0.0f && d.wv[1] == 0.0f && d.wv[2] == 0.0f);
        if (!rewarp_from_image(n, image, centre, g, x, y, z, d.wv, listed, d.rw)) {
            d.rw.lerp = false;
            d.rw.R = state_gather<3>(state_in, g, (float)x + d.wv[0], (float)y + d.wv[1], (float)(z + g.z_global_offset) + d.wv[2]);
        }
        if (!moved) {  // zero displacement: the gather returns live[p] bit for bit (every lerp is a*1 + b*0)
            d.rw.lerp = false;
            d.rw.R = l;
        }
        const unsigned long long q = listed ? pack_max(len, i + g.index_offset) : 0ull;
        best = q > best ? q : best;
        store(new_state(d), d.i);
        u = u1; u1 = u2;
        b0 = b1; b1 = b2;
        in0 = in1; in1 = in2;
        parity ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last round staged the last box once more: let it land before the wave ends)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
        best = q > best ? q : best;
    }
    double* dst[3] = {ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->data_energy : nullptr,
                      ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->smoothing_energy : nullptr,
                      ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->level_set_energy : nullptr};
    block_reduce_commit<3>(best, en, record_max(record), dst);
}

struct BoxLaunch {
    unsigned blocks;
    hipStream_t s;
    const vf4* state_in;
    const float* canonical;
    vf4* state_out;
    Grid g;
    Params p;
    lsf_gate gate;
    lsf_iteration_record* record;
    const lsf_band_box* boxes;
    unsigned box_count;
};

constexpr size_t kBoxLdsBytes = (size_t)kBoxWaves * 2 * kImage * sizeof(vf4);  // 136 KiB with 16 waves

template <int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
int launch_box(const BoxLaunch& a) {
    auto kernel = slavcheva_state_box_kernel<SMOOTH, LEVELSET, DATA, ENERGY>;
    static bool configured[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipGetLastError();
    if (!configured[dev]) {  // more than 64 KiB of dynamic LDS has to be asked for, once per device and instantiation
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kBoxLdsBytes) != hipSuccess)
            return (int)hipGetLastError();
        configured[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(a.blocks), dim3(kBoxThreads), kBoxLdsBytes, a.s, a.state_in, a.canonical, a.state_out, a.g,
                       a.p, a.gate, a.record, a.boxes, a.box_count);
    return 0;
}

template <int SMOOTH, bool LEVELSET, int DATA>
int pick_box_energy(int energy, const BoxLaunch& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: return launch_box<SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT>(a);
        case LSF_ENERGY_VECTORIZED: return launch_box<SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED>(a);
        default: return launch_box<SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE>(a);
    }
}

Params box_params(const lsf_slavcheva_params* q) {
    Params p;
    p.lambda64 = q->isomorphic_enforcement_factor_f64;
    p.rate = q->rate;
    p.w_data = q->data_term_weight;
    p.w_smooth = q->smoothing_term_weight;
    p.w_level_set = q->level_set_term_weight;
    p.lambda32 = q->isomorphic_enforcement_factor;
    p.killing_c1 = q->killing_c1;
    p.zero_gradient_on_snap = q->zero_gradient_on_snap;
    return p;
}

inline unsigned box_compute_units() {
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

// the canonical values of the boxes, box by box: out[64 b + lane] = canonical[box b's voxel (lane)]
__global__ __launch_bounds__(kBlock) void box_canonical_kernel(const float* __restrict__ canonical,
                                                               const lsf_band_box* __restrict__ boxes, unsigned box_count,
                                                               int sy, int sz, float* __restrict__ out) {
    const unsigned b = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (b >= box_count) return;
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    out[(size_t)b * kWave + lane] = canonical[boxes[b].origin + lz * sz + ly * sy + lx];
}

extern "C" int lsf_band_boxes_canonical(const float* canonical, const lsf_grid* grid, const lsf_band_box* boxes,
                                        int64_t box_count, float* canonical_boxed, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!canonical || !boxes || !canonical_boxed || box_count < 0 || box_count > 0x3ffffffll) return LSF_ERR_BAD_ARGUMENT;
    if (grid->dims != 3 || grid->nx % kBoxEdge || grid->ny % kBoxEdge || grid->nz % kBoxEdge ||
        (long long)grid->nx * grid->ny * grid->nz > 0x0fffffffll)
        return LSF_ERR_BAD_DIMS;
    if (box_count == 0) return 0;
    hipLaunchKernelGGL(box_canonical_kernel, dim3((unsigned)((box_count + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock),
                       0, as_stream(stream), canonical, boxes, (unsigned)box_count, grid->nx, grid->nx * grid->ny, canonical_boxed);
    return launch_status();
}

extern "C" int lsf_slavcheva_state_iteration_boxes(const float* state_in, const float* canonical_boxed, float* state_out,
                                                   const lsf_grid* grid, const lsf_slavcheva_params* params,
                                                   const lsf_gate* gate, lsf_iteration_record* record,
                                                   const lsf_band_box* boxes, int64_t box_count, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state_in || !canonical_boxed || !state_out || state_out == state_in || !params || !record || !boxes || box_count < 0 ||
        box_count > 0x3ffffffll)
        return LSF_ERR_BAD_ARGUMENT;
    // 3-D volumes of whole boxes whose float4 state fits 32-bit voxel arithmetic
    if (grid->dims != 3 || grid->nx % kBoxEdge || grid->ny % kBoxEdge || grid->nz % kBoxEdge ||
        (long long)grid->nx * grid->ny * grid->nz > 0x0fffffffll)
        return LSF_ERR_BAD_DIMS;
    if (box_count == 0) return 0;
    Grid g = make_grid(grid, 4);
    g.list_store_nt = box_count * 64ll * 32ll > 200ll * 1000 * 1000 * 5 / 4;  // as the list walk: lists too long for the Infinity Cache
    BoxLaunch a{cu_list_blocks((unsigned)box_count * kWave, box_compute_units()), as_stream(stream),
                reinterpret_cast<const vf4*>(state_in), canonical_boxed, reinterpret_cast<vf4*>(state_out), g, box_params(params),
                gate_or_open(gate), record, boxes, (unsigned)box_count};
    const bool killing = params->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = params->level_set_enabled != 0;
    const bool fdm = params->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = params->energy_mode;
    int status;
#define LSF_BOX(S, L, DM) status = pick_box_energy<S, L, DM>(e, a)
    if (killing) {
        if (ls) { if (fdm) LSF_BOX(LSF_SMOOTHING_KILLING, true, LSF_DATA_THRESHOLDED_FDM); else LSF_BOX(LSF_SMOOTHING_KILLING, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_BOX(LSF_SMOOTHING_KILLING, false, LSF_DATA_THRESHOLDED_FDM); else LSF_BOX(LSF_SMOOTHING_KILLING, false, LSF_DATA_BASIC); }
    } else {
        if (ls) { if (fdm) LSF_BOX(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_THRESHOLDED_FDM); else LSF_BOX(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_BOX(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_THRESHOLDED_FDM); else LSF_BOX(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_BASIC); }
    }
#undef LSF_BOX
    return status ? status : launch_status();
}
