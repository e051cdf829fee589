// Slavcheva-style (KillingFusion / SobolevFusion) optimizer iteration kernels, D = 2, 3
// (SURVEY 8a rows a3, a12-a18).  Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:163-330 with
// data_term.py, smoothing_term.py, level_set_term.py and field_warping.warp_field_advanced.
//
// This file: the planar-layout kernels of the SobolevFusion path -- gradient (+ energies), then the caller's
// zero-preserving separable filter (lsf_convolve_axis), then update + truncation-aware re-warp -- and the band-list
// builders.  Without a Sobolev filter the optimizers run ONE fused kernel per iteration on the float4 state layout:
// lsf_slavcheva_state.hip.
#include "lsf_device.h"

using namespace lsf;

#include "lsf_slavcheva_terms.h"

using namespace lsf::slav;

namespace {


// gradient of the energy at one voxel (a12-a17); zero outside the narrow-band union
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__device__ inline void voxel_gradient(const float* __restrict__ live, const float* __restrict__ canonical,
                                      const float* __restrict__ warp_prev, const Grid& g, const Params& p, int x,
                                      int y, int z, int i, float (&gv)[3], double (&en)[3]) {
    gv[0] = gv[1] = gv[2] = 0.0f;
    const float l = live[i], cn = canonical[i];
    if (fabsf(l) == 1.0f && fabsf(cn) == 1.0f) return;  // outside the narrow-band union (tsdf_set_routines.py:19-52)
    const bool interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && (D == 2 || (z > 0 && z < g.nz - 1));
    if (g.fast_ok && __all(interior)) {  // wave-uniform: every band lane has its whole neighbourhood
        const NbhFast<D> n(live, warp_prev, g, x, y, z);
        band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, en);
    } else {
        const Nbh<D> n(live, warp_prev, g, x, y, z);
        band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, en);
    }
}

// the re-warp's D-linear gather (OOB -> 1).  When the 2^D-voxel cell of every active lane lies inside the array (wave
// vote) the taps are buffer loads at one per-lane offset (the cell's lowest corner) + scalar tap offsets and need no
// OOB selects; the lerp order (z, then y, then x) and therefore the result is that of sample_linear.
template <int D>
__device__ inline float rewarp_gather(const float* __restrict__ live, const Grid& g, float px, float py, float pz) {
    const AxisTaps ax = axis_taps(px, g.nx, 0), ay = axis_taps(py, g.ny, 0);
    AxisTaps az = ax;
    if (D == 3) az = axis_taps(pz, g.nz, g.z_global_offset);
    const bool cell_inside = ax.v0 && ax.v1 && ay.v0 && ay.v1 && (D == 2 || (az.v0 && az.v1));
    if (!(g.fast_ok && __all(cell_inside))) return sample_linear_taps<D>(live, g, ax, ay, az, 1.0f);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(live), 0, -1, 0x00020000);
    const unsigned sy = (unsigned)g.nx * 4u, sz = (unsigned)(g.nx * g.ny) * 4u;
    const unsigned corner = (unsigned)(((D == 3 ? az.c0 : 0) * g.ny + ay.c0) * g.nx + ax.c0) * 4u;
    auto tap = [&](int dx, int dy, int dz) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             rsrc, (int)(corner + (unsigned)dx * 4u), (int)((unsigned)dy * sy + (unsigned)dz * sz), 0));
    };
    if (D == 2) {
        const float i0 = tap(0, 0, 0) * ay.i + tap(0, 1, 0) * ay.r;
        const float i1 = tap(1, 0, 0) * ay.i + tap(1, 1, 0) * ay.r;
        return i0 * ax.i + i1 * ax.r;
    }
    float c[2][2];
#pragma unroll
    for (int ox = 0; ox < 2; ++ox)
#pragma unroll
        for (int oy = 0; oy < 2; ++oy) c[ox][oy] = tap(ox, oy, 0) * az.i + tap(ox, oy, 1) * az.r;
    const float i0 = c[0][0] * ay.i + c[0][1] * ay.r;
    const float i1 = c[1][0] * ay.i + c[1][1] * ay.r;
    return i0 * ax.i + i1 * ax.r;
}

// warp = -g*rate, |warp| for the arg-max, truncation-aware re-warp of the live field (a18 + a3)
template <int D>
__device__ inline unsigned long long update_and_rewarp(const float* __restrict__ live, const Grid& g,
                                                       const Params& p, int x, int y, int z, int i,
                                                       float (&gv)[3], float* __restrict__ warp_out,
                                                       float* __restrict__ live_out, float* __restrict__ g_out) {
    float wv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < D; ++c) wv[c] = (-gv[c]) * p.rate;
    const float len = vec_length<D>(wv);
    const unsigned lin = linear_index(g, x, y, z);
    float v;
    if (wv[0] == 0.0f && wv[1] == 0.0f && wv[2] == 0.0f) {
        // zero displacement (every voxel outside the narrow-band union): all ratios are 0, each lerp is
        // a*1 + b*0 = a exactly, so the D-linear gather returns live[p] bit for bit -- skip its 2^D loads
        v = live[i];
    } else {
        const float px = (float)x + wv[0], py = (float)y + wv[1];
        const float pz = D == 3 ? (float)(z + g.z_global_offset) + wv[2] : 0.0f;
        v = rewarp_gather<D>(live, g, px, py, pz);
    }
    if (1.0f - fabsf(v) < 1e-6f) {  // field_warping.py:138-141
        v = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v);
        wv[0] = wv[1] = wv[2] = 0.0f;
        if (p.zero_gradient_on_snap) gv[0] = gv[1] = gv[2] = 0.0f;
    }
    live_out[i] = v;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        warp_out[c * g.plane + i] = wv[c];
        if (g_out) g_out[c * g.plane + i] = gv[c];
    }
    return pack_max(len, lin);
}

// gradient + energies of one iteration (the Sobolev path filters the gradient before the update)
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__global__ __launch_bounds__(kBlock) void slavcheva_gradient_kernel(
    const float* __restrict__ live, const float* __restrict__ canonical, const float* __restrict__ warp_prev,
    float* __restrict__ g_out, Grid g, Params p, lsf_gate gate, lsf_iteration_record* record,
    const int* __restrict__ band_list, unsigned band_count) {
    if (gate_closed(gate)) return;
    double en[3] = {0.0, 0.0, 0.0};
    // band_list != nullptr: only the listed voxels (ALL band voxels of the z-range); everywhere else the gradient is
    // zero and the caller's zero-initialised g_out already says so
    for_each_listed_voxel(g, band_list, band_count, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float gv[3];
        double e[3] = {0.0, 0.0, 0.0};
        voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(live, canonical, warp_prev, g, p, x, y, z, i, gv, e);
        en[0] += e[0];
        en[1] += e[1];
        en[2] += e[2];
#pragma unroll
        for (int c = 0; c < D; ++c) g_out[c * g.plane + i] = gv[c];
    });
    if (ENERGY != LSF_ENERGY_NONE) {
        double* dst[3] = {&record_slot(record)->data_energy, &record_slot(record)->smoothing_energy,
                          &record_slot(record)->level_set_energy};
        block_reduce_commit<3>(0ull, en, nullptr, dst);
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void slavcheva_update_rewarp_kernel(const float* __restrict__ live,
                                                                         float* __restrict__ gfield,
                                                                         float* __restrict__ warp_out,
                                                                         float* __restrict__ live_out, Grid g,
                                                                         Params p, lsf_gate gate,
                                                                         lsf_iteration_record* record,
                                                                         const int* __restrict__ band_list,
                                                                         unsigned band_count) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    for_each_listed_voxel(g, band_list, band_count, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float gv[3] = {gfield[i], gfield[g.plane + i], D == 3 ? gfield[2 * g.plane + i] : 0.0f};
        unsigned long long q = update_and_rewarp<D>(live, g, p, x, y, z, i, gv, warp_out, live_out,
                                                    p.zero_gradient_on_snap ? gfield : nullptr);
        best = q > best ? q : best;
    });
    if (band_list && blockIdx.x == 0 && threadIdx.x == 0) {  // the unlisted voxels: zero update, smallest index
        const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
        best = q > best ? q : best;
    }
    const double sums[1] = {0.0};
    double* dst[1] = {nullptr};
    block_reduce_commit<0>(best, sums, record_max(record), dst);
}

// The LAST pass of the zero-preserving filter (3-D: z, 2-D: x -- math_utils/convolution.py:94-111) at the voxels of a
// band list, fused with the update and the truncation-aware re-warp (slavcheva_optimizer2d.py:208-236): the filtered
// gradient of a voxel is all the update needs, so it never travels through memory in between and one of the five launches
// of a SobolevFusion iteration goes away.  Same arithmetic as convolve_list_kernel + slavcheva_update_rewarp_kernel:
// float64 products and sums in tap order, one float32 rounding, the mask of the RAW gradient, then update_and_rewarp.
// g_out receives the final gradient (zeroed where the live value snapped when the mode says so).
template <int D, int NT, bool FMA>
__global__ __launch_bounds__(kBlock) void slavcheva_filter_update_rewarp_kernel(
    const float* __restrict__ in, const float* __restrict__ mask_src, const float* __restrict__ live,
    float* __restrict__ g_out, float* __restrict__ warp_out, float* __restrict__ live_out, Grid g, Params p,
    TapsN<NT> taps, int axis, lsf_gate gate, lsf_iteration_record* record, const int* __restrict__ band_list,
    unsigned band_count) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    constexpr int c = NT / 2;
    const int len = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
    const int stride = axis == 0 ? 1 : (axis == 1 ? g.nx : g.nx * g.ny);
    for_each_listed_voxel(g, band_list, band_count, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        const int a = axis == 0 ? x : (axis == 1 ? y : z);
        float gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int plane = 0; plane < D; ++plane) {
            const long long base = (long long)plane * g.plane;
            const float* __restrict__ src = in + base + i;
            const float m = mask_src[base + i];
            float v[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {  // out[a] = sum_j k[j] * in[a + c - j], zero outside [0, len)
                const int d = c - j, q = a + d;
                const bool inside = q >= 0 && q < len;
                const float t = src[inside ? d * stride : 0];
                v[j] = inside ? t : 0.0f;
            }
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], (double)v[j]);
            gv[plane] = fabsf(m) < 1e-6f ? 0.0f : (float)acc;
        }
        const unsigned long long q = update_and_rewarp<D>(live, g, p, x, y, z, i, gv, warp_out, live_out, g_out);
        best = q > best ? q : best;
    });
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the unlisted voxels: zero update, smallest index
        const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
        best = q > best ? q : best;
    }
    const double sums[1] = {0.0};
    double* dst[1] = {nullptr};
    block_reduce_commit<0>(best, sums, record_max(record), dst);
}

Params make_params(const lsf_slavcheva_params* q) {
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

struct LaunchArgs {
    unsigned blocks;
    hipStream_t s;
    const float *live, *canonical, *warp_prev;
    float* g_out;
    Grid g;
    Params p;
    lsf_gate gate;
    lsf_iteration_record* record;
    const int* band_list;
    unsigned band_count;
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
void launch_one(const LaunchArgs& a) {
    hipLaunchKernelGGL((slavcheva_gradient_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY>), dim3(a.blocks), dim3(kBlock), 0,
                       a.s, a.live, a.canonical, a.warp_prev, a.g_out, a.g, a.p, a.gate, a.record, a.band_list,
                       a.band_count);
}

template <int D, int SMOOTH, bool LEVELSET, int DATA>
void pick_energy(int energy, const LaunchArgs& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT>(a); break;
        case LSF_ENERGY_VECTORIZED: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED>(a); break;
        default: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE>(a); break;
    }
}

template <int D>
void pick_terms(const lsf_slavcheva_params* q, const LaunchArgs& a) {
    const bool killing = q->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = q->level_set_enabled != 0;
    const bool fdm = q->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = q->energy_mode;
#define LSF_PICK(S, L, DM) pick_energy<D, S, L, DM>(e, a)
    if (killing) {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_BASIC); }
    } else {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_BASIC); }
    }
#undef LSF_PICK
}

// ------------------------------------------------------------------------------------------------------
// band list: ascending indices of the voxels inside the narrow-band union, by count -> scan -> fill
// ------------------------------------------------------------------------------------------------------
constexpr unsigned kBandChunk = 4 * kBlock;  // voxels per block of the count / fill kernels

// FILL == false: block_sums[b] = number of band voxels in chunk b.  FILL == true: block_sums holds the exclusive
// prefix sums; writes the chunk's band voxels, in ascending order, to list[block_sums[b] ...].
// subset: LSF_BAND_ALL, or only the band voxels whose whole 3^D neighbourhood lies inside the array (INTERIOR) / the
// others (BOUNDARY).
template <bool FILL>
__global__ __launch_bounds__(kBlock) void band_list_kernel(const float* __restrict__ live,
                                                           const float* __restrict__ canonical, unsigned first,
                                                           unsigned n, Grid g, int dims, int subset,
                                                           int* __restrict__ block_sums, int* __restrict__ list) {
    __shared__ int part[16];  // [pass j][wave]
    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    unsigned long long masks[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned v = blockIdx.x * kBandChunk + j * kBlock + t;
        bool in_band = false;
        if (v < n) in_band = !(fabsf(live[first + v]) == 1.0f && fabsf(canonical[first + v]) == 1.0f);
        if (in_band && subset != LSF_BAND_ALL) {
            const unsigned i = first + v;
            const unsigned zy = fast_div(i, g.div_nx);
            const int x = (int)(i - zy * (unsigned)g.nx);
            const int z = (int)fast_div(zy, g.div_ny);
            const int y = (int)zy - z * g.ny;
            const bool interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && (dims == 2 || (z > 0 && z < g.nz - 1));
            in_band = interior == (subset == LSF_BAND_INTERIOR);
        }
        masks[j] = __ballot(in_band);
        if (lane == 0) part[j * 4 + wave] = __popcll(masks[j]);
    }
    __syncthreads();
    if (!FILL) {
        if (t == 0) {
            int sum = 0;
            for (int k = 0; k < 16; ++k) sum += part[k];
            block_sums[blockIdx.x] = sum;
        }
        return;
    }
    int at = block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
            if (wv == wave && ((masks[j] >> lane) & 1ull))
                list[at + __popcll(masks[j] & ((1ull << lane) - 1ull))] =
                    (int)(first + blockIdx.x * kBandChunk + j * kBlock + t);
            at += part[j * 4 + wv];
        }
    }
}

// totals[2] = number of unlisted voxels with live = -canonical, totals[3] = the first of them (or -1); one block
__device__ inline void reduce_opposite(const int* __restrict__ opposite, unsigned chunks, long long* __restrict__ totals) {
    __shared__ long long s_sum[1024 / kWave];
    __shared__ int s_first[1024 / kWave];
    long long sum = 0;
    int first = 0x7fffffff;
    for (unsigned c = threadIdx.x; c < chunks; c += 1024) {
        sum += opposite[2 * c];
        first = min(first, opposite[2 * c + 1]);
    }
    for (int d = kWave / 2; d > 0; d >>= 1) {
        sum += (long long)shfl_down_u64((unsigned long long)sum, d);
        first = min(first, __shfl_down(first, d, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_sum[threadIdx.x / kWave] = sum;
        s_first[threadIdx.x / kWave] = first;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 1024 / kWave; ++k) {
            sum += s_sum[k];
            first = min(first, s_first[k]);
        }
        totals[2] = sum;
        totals[3] = first == 0x7fffffff ? -1 : first;
    }
}

// exclusive prefix sums of sums[0..n) in place, one block per array (blockIdx.x selects the array: arrays are
// `stride` ints apart, totals consecutive).  Tiles of 4096 ints: four consecutive ints per thread, a shuffle scan per
// wave, the sixteen wave totals through LDS, the running total carried from tile to tile.  A block past the arrays
// (`opposite` given) reduces lsf_state_prepare's per-chunk counts of the unlisted voxels instead.
__global__ __launch_bounds__(1024) void band_scan_kernel(int* __restrict__ sums_base, unsigned n, unsigned stride,
                                                         unsigned arrays, long long* totals,
                                                         const int* __restrict__ opposite) {
    if (blockIdx.x >= arrays) {
        reduce_opposite(opposite, n, totals);
        return;
    }
    __shared__ int wave_total[1024 / kWave];
    int* __restrict__ sums = sums_base + (size_t)blockIdx.x * stride;
    const unsigned t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    int carry = 0;
    for (unsigned base = 0; base < n; base += 4096u) {
        const unsigned i = base + 4u * t;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + k < n ? sums[i + k] : 0;
        const int mine = v[0] + v[1] + v[2] + v[3];
        int inclusive = mine;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int up = __shfl_up(inclusive, d, kWave);
            if ((int)lane >= d) inclusive += up;
        }
        if (lane == kWave - 1) wave_total[wave] = inclusive;
        __syncthreads();
        int before = 0, tile = 0;
#pragma unroll
        for (int w = 0; w < 1024 / kWave; ++w) {
            const int x = wave_total[w];
            before += w < (int)wave ? x : 0;
            tile += x;
        }
        int run = carry + before + inclusive - mine;  // exclusive prefix of this thread's four
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < n) sums[i + k] = run;
            run += v[k];
        }
        carry += tile;
        __syncthreads();  // wave_total is rewritten by the next tile
    }
    if (t == 0) totals[blockIdx.x] = carry;
}

// The same scan with ONE BLOCK PER TILE of 4096 counts (round 6: lsf_state_prepare's two arrays of per-chunk counts are
// 131 072 long at 512^3 -- 32 tiles that one block walked one after the other, two barriers and a dependent load each: 78 us
// where the whole counting pass takes 240).  Two launches: (1) every tile's total, and per tile the (count, first) of the
// unlisted voxels with live = -canonical; (2) every tile scans itself in place behind the sum of the totals in front of it.
// tile_out: [0, tiles) totals of array 0, [tiles, 2 tiles) of array 1, [2 tiles, 3 tiles) opposite counts, then their firsts.
constexpr unsigned kScanTile = 4096u;

__global__ __launch_bounds__(1024) void band_tile_totals_kernel(const int* __restrict__ sums_base, unsigned n,
                                                                unsigned stride, const int* __restrict__ opposite,
                                                                int* __restrict__ tile_out, unsigned tiles) {
    __shared__ int s_sum[1024 / kWave], s_first[1024 / kWave];
    const unsigned tile = blockIdx.x, a = blockIdx.y, t = threadIdx.x;
    const unsigned i = tile * kScanTile + 4u * t;
    int sum = 0, first = 0x7fffffff;
    if (a < 2u) {
        const int* __restrict__ sums = sums_base + (size_t)a * stride;
#pragma unroll
        for (unsigned k = 0; k < 4u; ++k) sum += i + k < n ? sums[i + k] : 0;
    } else {
#pragma unroll
        for (unsigned k = 0; k < 4u; ++k)
            if (i + k < n) {
                sum += opposite[2 * (i + k)];
                first = min(first, opposite[2 * (i + k) + 1]);
            }
    }
    for (int d = kWave / 2; d > 0; d >>= 1) {
        sum += __shfl_down(sum, d, kWave);
        first = min(first, __shfl_down(first, d, kWave));
    }
    if ((t & (kWave - 1)) == 0) {
        s_sum[t / kWave] = sum;
        s_first[t / kWave] = first;
    }
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 1024 / kWave; ++k) {
            sum += s_sum[k];
            first = min(first, s_first[k]);
        }
        tile_out[(size_t)a * tiles + tile] = sum;
        if (a == 2u) tile_out[(size_t)3 * tiles + tile] = first;
    }
}

__global__ __launch_bounds__(1024) void band_scan_tiles_kernel(int* __restrict__ sums_base, unsigned n, unsigned stride,
                                                               const int* __restrict__ tile_out, unsigned tiles,
                                                               long long* __restrict__ totals) {
    const unsigned tile = blockIdx.x, a = blockIdx.y, t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    if (a == 2u) {  // the unlisted voxels with live = -canonical: their number and the first of them (one thread: <= 256 tiles)
        if (tile == 0u && t == 0u) {
            long long sum = 0;
            int first = 0x7fffffff;
            for (unsigned k = 0; k < tiles; ++k) {
                sum += tile_out[(size_t)2 * tiles + k];
                first = min(first, tile_out[(size_t)3 * tiles + k]);
            }
            totals[2] = sum;
            totals[3] = first == 0x7fffffff ? -1 : first;
        }
        return;
    }
    __shared__ int wave_total[1024 / kWave];
    int* __restrict__ sums = sums_base + (size_t)a * stride;
    int carry = 0;  // the totals of the tiles in front of this one (every thread adds them up for itself: L2 hits)
    for (unsigned k = 0; k < tile; ++k) carry += tile_out[(size_t)a * tiles + k];
    const unsigned i = tile * kScanTile + 4u * t;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i + k < n ? sums[i + k] : 0;
    const int mine = v[0] + v[1] + v[2] + v[3];
    int inclusive = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int up = __shfl_up(inclusive, d, kWave);
        if ((int)lane >= d) inclusive += up;
    }
    if (lane == kWave - 1) wave_total[wave] = inclusive;
    __syncthreads();
    int before = 0, whole = 0;
#pragma unroll
    for (int w = 0; w < 1024 / kWave; ++w) {
        const int x = wave_total[w];
        before += w < (int)wave ? x : 0;
        whole += x;
    }
    int run = carry + before + inclusive - mine;  // exclusive prefix of this thread's four
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + k < n) sums[i + k] = run;
        run += v[k];
    }
    if (tile == tiles - 1u && t == 0u) totals[a] = (long long)carry + whole;
}

typedef float vf4 __attribute__((ext_vector_type(4)));

// start of an optimize() call on the fused path in ONE pass over live and canonical: the ping-pong states = (live, 0)
// (lsf_state_pack; `b` may be null -- the caller then fills the second state while the host waits for the list sizes)
// and the per-chunk counts of the INTERIOR and the BOUNDARY band voxels (lsf_band_count twice)
__global__ __launch_bounds__(kBlock) void state_prepare_kernel(const float* __restrict__ live,
                                                               const float* __restrict__ canonical,
                                                               vf4* __restrict__ a, vf4* __restrict__ b, unsigned n,
                                                               Grid g, int dims, int* __restrict__ sums_interior,
                                                               int* __restrict__ sums_boundary,
                                                               unsigned long long* __restrict__ masks,
                                                               int* __restrict__ opposite, int* __restrict__ nonempty) {
    __shared__ int part[4][16];  // [INTERIOR count, BOUNDARY count, opposite count, first opposite][pass j * 4 + wave]
    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned v = blockIdx.x * kBandChunk + j * kBlock + t;
        bool in_band = false, interior = false, opposed = false;
        if (v < n) {
            const float l = live[v], cn = canonical[v];
            in_band = !(fabsf(l) == 1.0f && fabsf(cn) == 1.0f);
            // outside the band |canonical - live| is 0 or -- live = -canonical -- exactly 2, now and for ever: what the
            // end-of-call statistics need to know about the voxels no list holds (lsf_state_finalize_listed)
            opposed = !in_band && l != cn;
            vf4 o;
            o.x = l; o.y = 0.0f; o.z = 0.0f; o.w = 0.0f;
            if (a) a[v] = o;  // null: the states are written by lsf_state_pack_needed, only where an iteration can read them
            if (b) b[v] = o;
            const unsigned zy = fast_div(v, g.div_nx);
            const int x = (int)(v - zy * (unsigned)g.nx);
            const int z = (int)fast_div(zy, g.div_ny);
            const int y = (int)zy - z * g.ny;
            interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && (dims == 2 || (z > 0 && z < g.nz - 1));
        }
        const unsigned long long mi = __ballot(in_band && interior), mb = __ballot(in_band && !interior);
        const unsigned long long mo = __ballot(opposed);
        if (lane == 0) {
            part[0][j * 4 + wave] = __popcll(mi);
            part[1][j * 4 + wave] = __popcll(mb);
            part[2][j * 4 + wave] = __popcll(mo);
            part[3][j * 4 + wave] = mo ? (int)(blockIdx.x * kBandChunk + j * kBlock + wave * kWave) + __ffsll((long long)mo) - 1
                                       : 0x7fffffff;
            // the ballots themselves: the fill step reads these 16 bytes per 64 voxels instead of live and canonical again
            masks[((size_t)blockIdx.x * 16 + j * 4 + wave) * 2 + 0] = mi;
            masks[((size_t)blockIdx.x * 16 + j * 4 + wave) * 2 + 1] = mb;
        }
    }
    __syncthreads();
    if (t < 2) {
        int sum = 0;
        for (int k = 0; k < 16; ++k) sum += part[t][k];
        (t == 0 ? sums_interior : sums_boundary)[blockIdx.x] = sum;
    } else if (t == 2) {
        int sum = 0, first = 0x7fffffff;
        for (int k = 0; k < 16; ++k) {
            sum += part[2][k];
            first = min(first, part[3][k]);
        }
        opposite[2 * blockIdx.x] = sum;
        opposite[2 * blockIdx.x + 1] = first;
    } else if (t == 3) {
        int any = 0;
        for (int k = 0; k < 16; ++k) any |= part[0][k] | part[1][k];
        nonempty[blockIdx.x] = any != 0;
    }
}

// The same pass with 16-byte loads: a thread takes FOUR consecutive voxels (rows of a multiple of four voxels, 16-byte
// aligned fields: the launcher checks), a block four chunks with the eight loads of a thread issued together -- a quarter
// of the load instructions and four times the bytes in flight per block (the dword version streams 128 MB at 3 TB/s at
// 256^3: 16 384 blocks of 8 KB each).  Outputs are those of state_prepare_kernel, bit for bit: the ballot words are per 64
// CONSECUTIVE voxels, i.e. per 16 lanes here -- a lane contributes a nibble at bit 4 * (lane % 16) and the sixteen lanes of
// a DPP row OR theirs together (a butterfly of four lane exchanges on either half).
constexpr int kPrepareChunksPerBlock = 4;
__device__ inline unsigned long long row16_or(unsigned nibble, int lane) {
    const int pos = lane & 15;
    unsigned lo = pos < 8 ? nibble << (4 * pos) : 0u, hi = pos < 8 ? 0u : nibble << (4 * (pos - 8));
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        lo |= (unsigned)__shfl_xor((int)lo, d, kWave);
        hi |= (unsigned)__shfl_xor((int)hi, d, kWave);
    }
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kBlock) void state_prepare4_kernel(const vf4* __restrict__ live4,
                                                                const vf4* __restrict__ canonical4,
                                                                vf4* __restrict__ a, vf4* __restrict__ b, unsigned n,
                                                                unsigned chunks, Grid g, int dims,
                                                                int* __restrict__ sums_interior,
                                                                int* __restrict__ sums_boundary,
                                                                unsigned long long* __restrict__ masks,
                                                                int* __restrict__ opposite, int* __restrict__ nonempty) {
    __shared__ int part[kPrepareChunksPerBlock][4][16];  // [chunk][INTERIOR, BOUNDARY, opposite count, first opposite][word]
    const int t = threadIdx.x, wave = t / kWave, lane = t % kWave;
    const unsigned chunk0 = blockIdx.x * kPrepareChunksPerBlock;
    vf4 l[kPrepareChunksPerBlock], c[kPrepareChunksPerBlock];
#pragma unroll
    for (int k = 0; k < kPrepareChunksPerBlock; ++k) {
        const unsigned v = (chunk0 + k) * kBandChunk + 4u * t;  // n is a multiple of four: a group is inside or outside
        const vf4 one = {1.0f, 1.0f, 1.0f, 1.0f};
        l[k] = v < n ? live4[v / 4] : one;
        c[k] = v < n ? canonical4[v / 4] : one;
    }
#pragma unroll
    for (int k = 0; k < kPrepareChunksPerBlock; ++k) {
        const unsigned chunk = chunk0 + k, v = chunk * kBandChunk + 4u * t;
        unsigned ni = 0u, nb = 0u, no = 0u;
        if (v < n) {
            const unsigned zy = fast_div(v, g.div_nx);
            const int x0 = (int)(v - zy * (unsigned)g.nx);  // the four voxels share a row (nx % 4 == 0)
            const int z = (int)fast_div(zy, g.div_ny);
            const int y = (int)zy - z * g.ny;
            const bool row_inside = y > 0 && y < g.ny - 1 && (dims == 2 || (z > 0 && z < g.nz - 1));
            const float lv[4] = {l[k].x, l[k].y, l[k].z, l[k].w}, cv[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in_band = !(fabsf(lv[j]) == 1.0f && fabsf(cv[j]) == 1.0f);
                const bool interior = row_inside && x0 + j > 0 && x0 + j < g.nx - 1;
                ni |= (in_band && interior ? 1u : 0u) << j;
                nb |= (in_band && !interior ? 1u : 0u) << j;
                no |= (!in_band && lv[j] != cv[j] ? 1u : 0u) << j;
                vf4 o;
                o.x = lv[j]; o.y = 0.0f; o.z = 0.0f; o.w = 0.0f;
                if (a) a[v + j] = o;
                if (b) b[v + j] = o;
            }
        }
        const unsigned long long mi = row16_or(ni, lane), mb = row16_or(nb, lane), mo = row16_or(no, lane);
        if ((lane & 15) == 0 && chunk < chunks) {
            const int word = wave * 4 + lane / 16;  // 64 consecutive voxels: chunk * 1024 + word * 64 ...
            part[k][0][word] = __popcll(mi);
            part[k][1][word] = __popcll(mb);
            part[k][2][word] = __popcll(mo);
            part[k][3][word] = mo ? (int)(chunk * kBandChunk + word * kWave) + __ffsll((long long)mo) - 1 : 0x7fffffff;
            masks[((size_t)chunk * 16 + word) * 2 + 0] = mi;
            masks[((size_t)chunk * 16 + word) * 2 + 1] = mb;
        }
    }
    __syncthreads();
    if (t < 4 * kPrepareChunksPerBlock) {
        const int k = t / 4, what = t % 4;
        const unsigned chunk = chunk0 + k;
        if (chunk < chunks) {
            if (what < 2) {
                int sum = 0;
                for (int w = 0; w < 16; ++w) sum += part[k][what][w];
                (what == 0 ? sums_interior : sums_boundary)[chunk] = sum;
            } else if (what == 2) {
                int sum = 0, first = 0x7fffffff;
                for (int w = 0; w < 16; ++w) {
                    sum += part[k][2][w];
                    first = min(first, part[k][3][w]);
                }
                opposite[2 * chunk] = sum;
                opposite[2 * chunk + 1] = first;
            } else {
                int any = 0;
                for (int w = 0; w < 16; ++w) any |= part[k][0][w] | part[k][1][w];
                nonempty[chunk] = any != 0;
            }
        }
    }
}

// The ping-pong states only where an iteration can READ them (lsf_state_pack_needed): chunk c (1024 consecutive voxels) is
// needed when some chunk that holds band voxels has a voxel within `reach` voxels of it along every axis -- the 3^D
// stencils reach 1, the re-warp gather of an update shorter than `reach` voxels floor(|w|) + 1 <= reach.  In linear
// indices: the voxels of chunk c shifted by dz * slice + dy * row, |dz|, |dy| <= reach, and widened by `reach` along x,
// fall into at most three consecutive chunks per (dz, dy) -- row ends only ever add candidates.  Two steps: the verdicts
// from the chunks' non-empty flags (written by lsf_state_prepare), then the needed chunks written as (live, 0) to both
// states; needed[c] keeps the verdict for whoever has to complete a state later (invert != 0: write exactly the chunks
// NOT needed, from the kept verdicts).
// step 1: the verdicts.  One WAVE per chunk, the (dz, dy) offsets dealt to its lanes (25 of them at reach 2: one round; a
// thread per chunk walked them one after the other, 12 us at 256^3 for a few dozen dependent flag loads each)
__global__ __launch_bounds__(kBlock) void chunk_needed_kernel(unsigned chunks, long long row, long long slice, int dims,
                                                              int reach, const int* __restrict__ nonempty,
                                                              int* __restrict__ needed) {
    const unsigned c = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (c >= chunks) return;
    const int lane = threadIdx.x % kWave;
    const long long c0 = (long long)c * kBandChunk;
    const int span = 2 * reach + 1, zs = dims == 3 ? span : 1;
    bool hit = false;
    for (int k = lane; k < zs * span; k += kWave) {
        const int dy = k % span - reach, dz = dims == 3 ? k / span - reach : 0;
        const long long lo = c0 + dz * slice + dy * row - reach, hi = c0 + (long long)kBandChunk - 1 + dz * slice + dy * row + reach;
        // floor division of possibly negative voxel indices; chunks lo / 1024 .. hi / 1024 (at most three)
        long long first = lo >= 0 ? lo / kBandChunk : -((-lo + kBandChunk - 1) / kBandChunk);
        long long last = hi >= 0 ? hi / kBandChunk : -((-hi + kBandChunk - 1) / kBandChunk);
        first = first < 0 ? 0 : first;
        last = last >= (long long)chunks ? (long long)chunks - 1 : last;
        for (long long cand = first; cand <= last; ++cand) hit |= nonempty[cand] != 0;
    }
    const bool any = __any(hit);
    if (lane == 0) needed[c] = any ? 1 : 0;
}

// step 2: (live, 0) into both states for the chunks whose verdict equals `want`: one chunk per block (four per block
// measured slower, 58 against 46 us at 256^3: the needed chunks come in runs, and a block that draws four of them keeps
// its CU's other slots waiting), two chunks in three leave after one load on a narrow band
constexpr int kPackChunks = 1;
__global__ __launch_bounds__(kBlock) void state_pack_needed_kernel(const float* __restrict__ live, vf4* __restrict__ a,
                                                                   vf4* __restrict__ b, unsigned n, unsigned chunks,
                                                                   const int* __restrict__ needed, int want) {
    const int t = threadIdx.x;
    int verdict[kPackChunks];
#pragma unroll
    for (int k = 0; k < kPackChunks; ++k) {
        const unsigned c = blockIdx.x * kPackChunks + k;
        verdict[k] = c < chunks ? needed[c] : -1;
    }
#pragma unroll
    for (int k = 0; k < kPackChunks; ++k) {
        if (verdict[k] != want) continue;
        const unsigned c = blockIdx.x * kPackChunks + k;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v = c * kBandChunk + j * kBlock + t;
            if (v < n) {
                vf4 o;
                o.x = live[v]; o.y = 0.0f; o.z = 0.0f; o.w = 0.0f;
                if (a) a[v] = o;
                if (b) b[v] = o;
            }
        }
    }
}



// ordered fill of one subset's list from the ballots lsf_state_prepare kept (which = 0 INTERIOR, 1 BOUNDARY).  One WAVE
// per 1024-voxel chunk: its sixteen ballots arrive in one load (lane w holds word w), a chunk without entries -- nine
// out of ten on a narrow band -- ends there, otherwise the words are handed round and every lane writes its own voxel.
__global__ __launch_bounds__(kBlock) void band_fill_from_masks_kernel(const unsigned long long* __restrict__ masks,
                                                                      const int* __restrict__ block_sums, int which,
                                                                      int* __restrict__ list, unsigned chunks) {
    const int lane = threadIdx.x % kWave;
    const unsigned chunk = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (chunk >= chunks) return;
    const unsigned long long mine = lane < 16 ? masks[((size_t)chunk * 16 + lane) * 2 + which] : 0ull;
    if (__ballot(mine != 0ull) == 0ull) return;
    int at = block_sums[chunk];
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned lo = (unsigned)__shfl((int)(unsigned)mine, w, kWave);
        const unsigned hi = (unsigned)__shfl((int)(unsigned)(mine >> 32), w, kWave);
        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
        if ((m >> lane) & 1ull)
            list[at + __popcll(m & ((1ull << lane) - 1ull))] = (int)(chunk * kBandChunk + w * kWave + lane);
        at += __popcll(m);
    }
}

}  // namespace

static inline bool band_subset_ok(int32_t subset) {
    return subset == LSF_BAND_ALL || subset == LSF_BAND_INTERIOR || subset == LSF_BAND_BOUNDARY;
}

extern "C" int lsf_slavcheva_gradient(const float* live, const float* canonical, const float* warp_prev_planar,
                                      float* g_out_planar, const lsf_grid* grid, const lsf_slavcheva_params* params,
                                      const lsf_gate* gate, lsf_iteration_record* record, const int32_t* band_list,
                                      int64_t band_count, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !canonical || !warp_prev_planar || !params || !record || !g_out_planar ||
        g_out_planar == warp_prev_planar || (band_list && (band_count < 0 || band_count > 0x7fffffffll)))
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    LaunchArgs a{band_list ? band_list_blocks((unsigned)band_count) : launch_blocks(t.total), as_stream(stream), live,
                 canonical, warp_prev_planar, g_out_planar, g, make_params(params), gate_or_open(gate), record, band_list,
                 (unsigned)band_count};
    if (grid->dims == 2) pick_terms<2>(params, a);
    else pick_terms<3>(params, a);
    return launch_status();
}

extern "C" int lsf_slavcheva_update_rewarp(const float* live, const float* canonical, float* g_planar,
                                           float* warp_out_planar, float* live_out, const lsf_grid* grid,
                                           const lsf_slavcheva_params* params, const lsf_gate* gate,
                                           lsf_iteration_record* record, const int32_t* band_list, int64_t band_count,
                                           void* stream) {
    (void)canonical;
    if (int e = check_grid(grid)) return e;
    if (!live || !g_planar || !warp_out_planar || !live_out || live_out == live || !params || !record ||
        (band_list && (band_count < 0 || band_count > 0x7fffffffll)))
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    Params p = make_params(params);
    lsf_gate gt = gate_or_open(gate);
    const unsigned blocks = band_list ? band_list_blocks((unsigned)band_count) : launch_blocks(t.total);
    if (grid->dims == 2)
        hipLaunchKernelGGL(slavcheva_update_rewarp_kernel<2>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), live,
                           g_planar, warp_out_planar, live_out, g, p, gt, record, band_list, (unsigned)band_count);
    else
        hipLaunchKernelGGL(slavcheva_update_rewarp_kernel<3>, dim3(blocks), dim3(kBlock), 0, as_stream(stream), live,
                           g_planar, warp_out_planar, live_out, g, p, gt, record, band_list, (unsigned)band_count);
    return launch_status();
}

template <int D, int NT>
static void launch_filter_update(bool fma, unsigned blocks, hipStream_t s, const float* in, const float* mask,
                                 const float* live, float* g_out, float* warp_out, float* live_out, const Grid& g,
                                 const Params& p, const double* taps_host, int axis, const lsf_gate& gt,
                                 lsf_iteration_record* record, const int* list, unsigned count) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    if (fma)
        hipLaunchKernelGGL((slavcheva_filter_update_rewarp_kernel<D, NT, true>), dim3(blocks), dim3(kBlock), 0, s, in, mask,
                           live, g_out, warp_out, live_out, g, p, taps, axis, gt, record, list, count);
    else
        hipLaunchKernelGGL((slavcheva_filter_update_rewarp_kernel<D, NT, false>), dim3(blocks), dim3(kBlock), 0, s, in, mask,
                           live, g_out, warp_out, live_out, g, p, taps, axis, gt, record, list, count);
}

extern "C" int lsf_slavcheva_filter_update_rewarp(const float* in_planar, const float* zero_mask_source,
                                                  const float* live, float* g_out_planar, float* warp_out_planar,
                                                  float* live_out, const lsf_grid* grid,
                                                  const lsf_slavcheva_params* params, int32_t axis,
                                                  const double* taps_host, int32_t n_taps, const lsf_gate* gate,
                                                  lsf_iteration_record* record, const int32_t* band_list,
                                                  int64_t band_count, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in_planar || !zero_mask_source || !live || !g_out_planar || !warp_out_planar || !live_out ||
        live_out == live || g_out_planar == in_planar || !params || !record || !taps_host || !band_list ||
        band_count < 0 || band_count > 0x7fffffffll || axis < 0 || axis >= grid->dims)
        return LSF_ERR_BAD_ARGUMENT;
    if (n_taps != 3 && n_taps != 5 && n_taps != 7 && n_taps != 9) return LSF_ERR_KERNEL_TOO_LONG;
    Grid g = make_grid(grid);
    if (g.z_end == g.z_begin) return 0;
    const Params p = make_params(params);
    const lsf_gate gt = gate_or_open(gate);
    const unsigned blocks = band_list_blocks((unsigned)band_count);
    hipStream_t s = as_stream(stream);
    bool fma = true;
    for (int j = 0; j < n_taps; ++j) fma = fma && (double)(float)taps_host[j] == taps_host[j];
    const int* list = band_list;
    const unsigned count = (unsigned)band_count;
#define LSF_FUR(D, NT) launch_filter_update<D, NT>(fma, blocks, s, in_planar, zero_mask_source, live, g_out_planar, \
                                                 warp_out_planar, live_out, g, p, taps_host, axis, gt, record, list, count)
    if (grid->dims == 2) {
        switch (n_taps) { case 3: LSF_FUR(2, 3); break; case 5: LSF_FUR(2, 5); break; case 7: LSF_FUR(2, 7); break; default: LSF_FUR(2, 9); break; }
    } else {
        switch (n_taps) { case 3: LSF_FUR(3, 3); break; case 5: LSF_FUR(3, 5); break; case 7: LSF_FUR(3, 7); break; default: LSF_FUR(3, 9); break; }
    }
#undef LSF_FUR
    return launch_status();
}

// ---- band list (see include/lsf_hip.h)
static inline bool band_range(const lsf_grid* grid, unsigned& first, unsigned& n, unsigned& chunks) {
    const long long slice = (long long)grid->ny * grid->nx;
    first = (unsigned)(slice * grid->z_begin);
    n = (unsigned)(slice * (grid->z_end - grid->z_begin));
    chunks = (n + kBandChunk - 1) / kBandChunk;
    return n > 0;
}

extern "C" int64_t lsf_band_scratch_elements(const lsf_grid* grid) {
    if (check_grid(grid)) return 0;
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    return (int64_t)chunks + 1;
}

extern "C" int lsf_band_count(const float* live, const float* canonical, const lsf_grid* grid, int32_t subset,
                              int32_t* scratch, int64_t* count_out, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !canonical || !scratch || !count_out || !band_subset_ok(subset)) return LSF_ERR_BAD_ARGUMENT;
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    hipStream_t s = as_stream(stream);
    if (chunks > 0)
        hipLaunchKernelGGL(band_list_kernel<false>, dim3(chunks), dim3(kBlock), 0, s, live, canonical, first, n,
                           make_grid(grid), grid->dims, subset, scratch, (int*)nullptr);
    hipLaunchKernelGGL(band_scan_kernel, dim3(1), dim3(1024), 0, s, scratch, chunks, 0u, 1u, (long long*)count_out,
                       (const int*)nullptr);
    return launch_status();
}

// scratch of lsf_state_prepare: [chunks + 1] INTERIOR sums, [chunks + 1] BOUNDARY sums, then (8-byte aligned) the two
// ballots of every 64-voxel group: 32 uint64 per 1024-voxel chunk, then per chunk (count, first) of the unlisted voxels
// with live = -canonical
static inline unsigned long long* prepare_masks(int32_t* scratch, unsigned chunks) {
    return reinterpret_cast<unsigned long long*>(scratch + 2 * (size_t)(chunks + 1) + (2 * (chunks + 1)) % 2);
}

static inline int* prepare_opposite(int32_t* scratch, unsigned chunks) {
    return reinterpret_cast<int*>(prepare_masks(scratch, chunks) + (size_t)chunks * 32);
}

// ... then one int per chunk: does the chunk hold band voxels; then one per chunk: lsf_state_pack_needed's verdict
static inline int* prepare_nonempty(int32_t* scratch, unsigned chunks) { return prepare_opposite(scratch, chunks) + 2 * (size_t)chunks; }
static inline int* prepare_needed(int32_t* scratch, unsigned chunks) { return prepare_nonempty(scratch, chunks) + chunks; }
// ... then the scan's per-tile totals (band_tile_totals_kernel): 4 ints per tile of 4096 chunks
static inline unsigned prepare_scan_tiles(unsigned chunks) { return (chunks + kScanTile - 1u) / kScanTile; }
static inline int* prepare_tile_totals(int32_t* scratch, unsigned chunks) { return prepare_needed(scratch, chunks) + chunks; }

extern "C" int64_t lsf_state_prepare_scratch_elements(const lsf_grid* grid) {
    if (check_grid(grid, true)) return 0;
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    return 2 * (int64_t)(chunks + 1) + 64 * (int64_t)chunks + 2 + 2 * (int64_t)chunks + 2 * (int64_t)chunks +
           4 * ((int64_t)prepare_scan_tiles(chunks) + 1);
}

extern "C" int lsf_band_list_fill_prepared(const lsf_grid* grid, int32_t subset, const int32_t* scratch, int32_t* list,
                                           void* stream) {
    if (int e = check_grid(grid, true)) return e;
    if (!scratch || !list || (subset != LSF_BAND_INTERIOR && subset != LSF_BAND_BOUNDARY) || grid->z_begin != 0 ||
        grid->z_end != grid->nz)
        return LSF_ERR_BAD_ARGUMENT;
    unsigned first, n, chunks;
    if (!band_range(grid, first, n, chunks)) return 0;
    const int which = subset == LSF_BAND_INTERIOR ? 0 : 1;
    hipLaunchKernelGGL(band_fill_from_masks_kernel, dim3((chunks + kBlock / kWave - 1) / (kBlock / kWave)), dim3(kBlock),
                       0, as_stream(stream), prepare_masks(const_cast<int32_t*>(scratch), chunks),
                       scratch + which * (size_t)(chunks + 1), which, list, chunks);
    return launch_status();
}

// ---- boxes for lsf_slavcheva_state_iteration_boxes (lsf_slavcheva_box.hip), from the ballots lsf_state_prepare kept ------
// Box B = (bz * ny/4 + by) * nx/4 + bx covers voxels (4bx .. 4bx+3, 4by .. 4by+3, 4bz .. 4bz+3); its mask takes four bits
// from each of its sixteen x-rows out of the INTERIOR ballot word of the 64 consecutive voxels the row starts in (rows of a
// multiple of four voxels never straddle a word).  One thread per box; sixteen boxes share a word.
namespace {
constexpr int kBoxGroup = kBlock;  // boxes per block = per count of the scan

__device__ inline unsigned long long box_mask_of(const unsigned long long* __restrict__ masks, unsigned box, int nbx,
                                                 int nby, int nx, int ny, bool all) {
    const int bx = (int)(box % (unsigned)nbx), by = (int)((box / (unsigned)nbx) % (unsigned)nby);
    const int bz = (int)(box / ((unsigned)nbx * (unsigned)nby));
    unsigned long long m = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned v = (unsigned)(((4 * bz + k) * ny + 4 * by + j) * nx + 4 * bx);
            // [2 W + 0]: the INTERIOR ballot of voxels 64 W ..., [2 W + 1]: the BOUNDARY ballot (all: their union)
            const unsigned long long word = masks[2 * (size_t)(v >> 6)] | (all ? masks[2 * (size_t)(v >> 6) + 1] : 0ull);
            m |= ((word >> (v & 63u)) & 0xFull) << (4 * (k * 4 + j));
        }
    return m;
}

template <bool FILL>
__global__ __launch_bounds__(kBlock) void band_boxes_kernel(const unsigned long long* __restrict__ masks, unsigned n_boxes,
                                                            int nbx, int nby, int nx, int ny, bool all,
                                                            int* __restrict__ group_sums, lsf_band_box* __restrict__ boxes) {
    __shared__ int wave_count[kBlock / kWave];
    const unsigned box = blockIdx.x * kBoxGroup + threadIdx.x;
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    const unsigned long long m = box < n_boxes ? box_mask_of(masks, box, nbx, nby, nx, ny, all) : 0ull;
    const unsigned long long any = __ballot(m != 0ull);
    if (lane == 0) wave_count[wave] = __popcll(any);
    __syncthreads();
    if (!FILL) {
        if (threadIdx.x == 0) {
            int sum = 0;
            for (int k = 0; k < kBlock / kWave; ++k) sum += wave_count[k];
            group_sums[blockIdx.x] = sum;
        }
        return;
    }
    if (m == 0ull) return;
    int at = group_sums[blockIdx.x];  // exclusive prefix after the scan
    for (int k = 0; k < wave; ++k) at += wave_count[k];
    at += __popcll(any & ((1ull << lane) - 1ull));
    const int bx = (int)(box % (unsigned)nbx), by = (int)((box / (unsigned)nbx) % (unsigned)nby);
    const int bz = (int)(box / ((unsigned)nbx * (unsigned)nby));
    lsf_band_box b;
    b.origin = ((4 * bz) * ny + 4 * by) * nx + 4 * bx;
    b.reserved = 0;
    b.mask = m;
    boxes[at] = b;
}

inline bool boxes_ok(const lsf_grid* g) {
    return g->dims == 3 && g->nx % 4 == 0 && g->ny % 4 == 0 && g->nz % 4 == 0 && g->z_begin == 0 && g->z_end == g->nz &&
           (long long)g->nx * g->ny * g->nz <= 0x0fffffffll;
}
}  // namespace

extern "C" int64_t lsf_band_boxes_scratch_elements(const lsf_grid* grid) {
    if (check_grid(grid) || !boxes_ok(grid)) return 0;
    const long long n_boxes = (long long)grid->nx * grid->ny * grid->nz / 64;
    return (n_boxes + kBoxGroup - 1) / kBoxGroup + 1;
}

extern "C" int lsf_band_boxes_count(const lsf_grid* grid, int32_t subset, const int32_t* prepare_scratch,
                                    int32_t* box_scratch, int64_t* count_out, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!boxes_ok(grid)) return LSF_ERR_BAD_DIMS;
    if (!prepare_scratch || !box_scratch || !count_out || (subset != LSF_BAND_INTERIOR && subset != LSF_BAND_ALL))
        return LSF_ERR_BAD_ARGUMENT;
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    const unsigned n_boxes = n / 64, groups = (n_boxes + kBoxGroup - 1) / kBoxGroup;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(band_boxes_kernel<false>, dim3(groups), dim3(kBlock), 0, s,
                       prepare_masks(const_cast<int32_t*>(prepare_scratch), chunks), n_boxes, grid->nx / 4, grid->ny / 4,
                       grid->nx, grid->ny, subset == LSF_BAND_ALL, box_scratch, (lsf_band_box*)nullptr);
    hipLaunchKernelGGL(band_scan_kernel, dim3(1), dim3(1024), 0, s, box_scratch, groups, 0u, 1u, (long long*)count_out,
                       (const int*)nullptr);
    return launch_status();
}

extern "C" int lsf_band_boxes_fill(const lsf_grid* grid, int32_t subset, const int32_t* prepare_scratch,
                                   const int32_t* box_scratch, lsf_band_box* boxes, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!boxes_ok(grid)) return LSF_ERR_BAD_DIMS;
    if (!prepare_scratch || !box_scratch || !boxes || (subset != LSF_BAND_INTERIOR && subset != LSF_BAND_ALL))
        return LSF_ERR_BAD_ARGUMENT;
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    const unsigned n_boxes = n / 64, groups = (n_boxes + kBoxGroup - 1) / kBoxGroup;
    hipLaunchKernelGGL(band_boxes_kernel<true>, dim3(groups), dim3(kBlock), 0, as_stream(stream),
                       prepare_masks(const_cast<int32_t*>(prepare_scratch), chunks), n_boxes, grid->nx / 4, grid->ny / 4,
                       grid->nx, grid->ny, subset == LSF_BAND_ALL, const_cast<int32_t*>(box_scratch), boxes);
    return launch_status();
}

extern "C" int lsf_state_prepare(const float* live, const float* canonical, float* state_a, float* state_b,
                                 const lsf_grid* grid, int32_t* scratch, int64_t* counts_out, void* stream) {
    if (int e = check_grid(grid, true)) return e;
    // state_b: optional; state_a AND state_b NULL: the states are written by lsf_state_pack_needed
    if (!live || !canonical || (!state_a && state_b) || !scratch || !counts_out) return LSF_ERR_BAD_ARGUMENT;
    if (grid->z_begin != 0 || grid->z_end != grid->nz) return LSF_ERR_BAD_ARGUMENT;  // whole arrays only
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    hipStream_t s = as_stream(stream);
    int* sums_interior = scratch;
    int* sums_boundary = scratch + (chunks + 1);
    const bool wide = grid->nx % 4 == 0 && ((uintptr_t)live | (uintptr_t)canonical) % 16 == 0;
    if (wide)  // 16-byte loads, four chunks per block
        hipLaunchKernelGGL(state_prepare4_kernel, dim3((chunks + kPrepareChunksPerBlock - 1) / kPrepareChunksPerBlock),
                           dim3(kBlock), 0, s, reinterpret_cast<const vf4*>(live), reinterpret_cast<const vf4*>(canonical),
                           reinterpret_cast<vf4*>(state_a), reinterpret_cast<vf4*>(state_b), n, chunks, make_grid(grid),
                           grid->dims, sums_interior, sums_boundary, prepare_masks(scratch, chunks),
                           prepare_opposite(scratch, chunks), prepare_nonempty(scratch, chunks));
    else
        hipLaunchKernelGGL(state_prepare_kernel, dim3(chunks), dim3(kBlock), 0, s, live, canonical,
                           reinterpret_cast<vf4*>(state_a), reinterpret_cast<vf4*>(state_b), n, make_grid(grid), grid->dims,
                           sums_interior, sums_boundary, prepare_masks(scratch, chunks), prepare_opposite(scratch, chunks),
                           prepare_nonempty(scratch, chunks));
    // the scan of the two count arrays and the reduction of the unlisted voxels' counts: a block per tile of 4096 chunks
    const unsigned tiles = prepare_scan_tiles(chunks) > 0u ? prepare_scan_tiles(chunks) : 1u;
    hipLaunchKernelGGL(band_tile_totals_kernel, dim3(tiles, 3), dim3(1024), 0, s, scratch, chunks, chunks + 1,
                       (const int*)prepare_opposite(scratch, chunks), prepare_tile_totals(scratch, chunks), tiles);
    hipLaunchKernelGGL(band_scan_tiles_kernel, dim3(tiles, 3), dim3(1024), 0, s, scratch, chunks, chunks + 1,
                       (const int*)prepare_tile_totals(scratch, chunks), tiles, (long long*)counts_out);
    return launch_status();
}

extern "C" int lsf_state_pack_needed(const float* live, float* state_a, float* state_b, const lsf_grid* grid,
                                     int32_t* scratch, int32_t reach, int32_t invert, void* stream) {
    if (int e = check_grid(grid, true)) return e;  // whole local arrays, also of slabs cut along y (as lsf_state_prepare)
    if (!live || (!state_a && !state_b) || !scratch || reach < 1 || reach > 8) return LSF_ERR_BAD_ARGUMENT;
    if (grid->z_begin != 0 || grid->z_end != grid->nz) return LSF_ERR_BAD_ARGUMENT;  // whole arrays only
    unsigned first, n, chunks;
    band_range(grid, first, n, chunks);
    if (!invert)
        hipLaunchKernelGGL(chunk_needed_kernel, dim3((chunks + kBlock / kWave - 1) / (kBlock / kWave)), dim3(kBlock), 0, as_stream(stream),
                           chunks, (long long)grid->nx, (long long)grid->nx * grid->ny, grid->dims, reach,
                           prepare_nonempty(scratch, chunks), prepare_needed(scratch, chunks));
    hipLaunchKernelGGL(state_pack_needed_kernel, dim3((chunks + kPackChunks - 1) / kPackChunks), dim3(kBlock), 0,
                       as_stream(stream), live, reinterpret_cast<vf4*>(state_a), reinterpret_cast<vf4*>(state_b), n, chunks,
                       prepare_needed(scratch, chunks), invert ? 0 : 1);
    return launch_status();
}

extern "C" int lsf_band_list_fill(const float* live, const float* canonical, const lsf_grid* grid, int32_t subset,
                                  const int32_t* scratch, int32_t* list, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !canonical || !scratch || !list || !band_subset_ok(subset)) return LSF_ERR_BAD_ARGUMENT;
    unsigned first, n, chunks;
    if (!band_range(grid, first, n, chunks)) return 0;
    hipLaunchKernelGGL(band_list_kernel<true>, dim3(chunks), dim3(kBlock), 0, as_stream(stream), live, canonical, first, n,
                       make_grid(grid), grid->dims, subset, const_cast<int32_t*>(scratch), list);
    return launch_status();
}
