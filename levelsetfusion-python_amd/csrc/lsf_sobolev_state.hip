// The SobolevFusion iteration (zero-preserving separable filter between gradient and update) on the float4 layouts, at
// the voxels of band lists.  Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:163-236 (VECTORIZED) / :238-330
// (DIRECT) with math_utils/convolution.py:114-132.
//
// Why another set of kernels: the list passes of this path cost per vector-memory INSTRUCTION, not per byte (DESIGN.md
// section 7, round 3: one plane of the planar filter pass takes 13 us at 256^3, three planes 17 / 22 / 35 us) -- and on
// planar fields every tap of every component is an instruction of its own.  Here
//   state   float4 [z][y][x] = (live, u, v, w)   the fused path's layout: live and warp of a neighbour in ONE load
//   g4      float4 [z][y][x] = (g_x, g_y, g_z, 0)  gradient, filter intermediates: the three components of a tap in ONE load
// so that per listed voxel the gradient reads 7 (Tikhonov) to 19 (Killing / level set) neighbours instead of 28 to 76
// dwords, a filter pass 8 float4s instead of 24 dwords, and the last pass + update + re-warp writes two float4s instead
// of seven dwords.  Same per-voxel arithmetic as the planar kernels (lsf_slavcheva.hip) and as the fused kernel
// (lsf_slavcheva_state_taps.h): results are bit-identical to both.
#include "lsf_slavcheva_state_taps.h"

using namespace lsf;
using namespace lsf::slav;

namespace {

__device__ inline void decode_listed(const Grid& g, unsigned i, int& x, int& y, int& z) {
    const unsigned zy = fast_div(i, g.div_nx);
    x = (int)(i - zy * (unsigned)g.nx);
    z = (int)fast_div(zy, g.div_ny);
    y = (int)zy - z * g.ny;
}

// raw gradient (and energy terms) of ONE listed voxel; zero where the voxel has left the narrow-band union
// (tsdf_set_routines.py:19-52: a listed voxel may have snapped to +-1)
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__device__ inline void listed_voxel_gradient(const vf4* __restrict__ state, const float* __restrict__ canonical,
                                             const Grid& g, const Params& p, int i, int x, int y, int z, float (&gv)[3],
                                             double (&e)[3]) {
    const vf4 sc = state[i];
    const float l = sc.x, cn = canonical[i];
    gv[0] = gv[1] = gv[2] = 0.0f;
    e[0] = e[1] = e[2] = 0.0;
    if (!(fabsf(l) == 1.0f && fabsf(cn) == 1.0f)) {
        const bool interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && (D == 2 || (z > 0 && z < g.nz - 1));
        if (g.wide_ok && __all(interior) && wave_span_ok(g, i)) {
            NbhStateFast<D> n;
            n.load(state, g, (unsigned)i, sc, !g.fast_ok);
            band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, e);
        } else {
            const NbhState<D> n(state, g, x, y, z, sc);
            band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, e);
        }
    }
}

// gradient (+ energies) of the listed voxels -> g_raw4; everything else in g_raw4 keeps the caller's zeros
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__global__ __launch_bounds__(kBlock) void sobolev_state_gradient_kernel(const vf4* __restrict__ state,
                                                                        const float* __restrict__ canonical,
                                                                        vf4* __restrict__ g_raw, Grid g, Params p,
                                                                        lsf_gate gate, lsf_iteration_record* record,
                                                                        const int* __restrict__ band_list,
                                                                        unsigned band_count) {
    if (gate_closed(gate)) return;
    double en[3] = {0.0, 0.0, 0.0};
    for_each_listed_voxel(g, band_list, band_count, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float gv[3];
        double e[3];
        listed_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(state, canonical, g, p, i, x, y, z, gv, e);
        if (z >= g.e_begin && z < g.e_end) {
            en[0] += e[0];
            en[1] += e[1];
            en[2] += e[2];
        }
        vf4 o;
        o.x = gv[0]; o.y = gv[1]; o.z = D == 3 ? gv[2] : 0.0f; o.w = 0.0f;
        g_raw[i] = o;
    });
    if (ENERGY != LSF_ENERGY_NONE) {
        double* dst[3] = {&record_slot(record)->data_energy, &record_slot(record)->smoothing_energy,
                          &record_slot(record)->level_set_energy};
        block_reduce_commit<3>(0ull, en, nullptr, dst);
    }
}

// Gradient AND the first filter pass (along x, 3-D: math_utils/convolution.py:94-105 filters x first) in one launch: the
// raw gradient never goes to memory.  Why this is possible on a band list: the raw gradient is ZERO at every voxel that
// is not listed (the filter's input buffers are zero-initialised and only listed voxels are ever written), so the x -/+ d
// taps of a listed voxel are either entries of the same ascending list a few positions away -- same row, consecutive x --
// or zero.  A workgroup takes 256 consecutive entries, every thread computes its entry's raw gradient into LDS, and the
// threads c = n_taps / 2 entries away from either end of the tile filter theirs (tiles overlap by 2c entries: 2.4 % of
// the gradient work twice, for a 7-tap filter).  Same float64 sums in the same tap order as filtered_at (a missing
// tap adds k * 0 exactly as a stored zero does), same mask bits in the fourth component: bit-identical to gradient ->
// raw -> lsf_convolve_axis_listed4(axis 0).  Saves per listed voxel and iteration one 16-byte store, eight 16-byte loads
// and a launch (21.4 + 15.9 us of kernels at 256^3, profiles/r04_sobolev_pmc_hbm_traffic.csv), and per call one
// zero-filled gradient buffer.
constexpr int kMaxTaps = 9;
// workgroups per XCD of the fused gradient + x launch: two to three tiles each at 256^3 (a tile is a chain of dependent
// round trips with two barriers; 7 / 4 / 3 / 2 / 1 tiles per workgroup: 79 / 73.7 / 71.0 / 70.4-71.5 / 74.5 us per iteration)
constexpr unsigned kGradientXBlocksPerXcd = 512u;

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, bool FMA>
__global__ __launch_bounds__(kBlock) void sobolev_state_gradient_x_kernel(const vf4* __restrict__ state,
                                                                          const float* __restrict__ canonical,
                                                                          vf4* __restrict__ out, Grid g, Params p,
                                                                          TapsN<kMaxTaps> taps, int n_taps, int bricks,
                                                                          lsf_gate gate, lsf_iteration_record* record,
                                                                          const int* __restrict__ band_list,
                                                                          unsigned band_count) {
    if (gate_closed(gate)) return;
    __shared__ int s_index[kBlock];
    __shared__ vf4 s_raw[kBlock];
    const int c = n_taps / 2, t = (int)threadIdx.x;
    const unsigned per_tile = (unsigned)(kBlock - 2 * c);  // entries a tile FILTERS; it computes c more on either side
    const unsigned tiles = (band_count + per_tile - 1) / per_tile;
    // the list walk of for_each_listed_voxel: XCD k owns the k-th eighth of the tiles (blocks b, b + 8, ... share an XCD)
    unsigned first = blockIdx.x, step = gridDim.x, end = tiles;
    if (gridDim.x % kXcds == 0) {
        const unsigned per_xcd = (tiles + kXcds - 1) / kXcds, xcd = blockIdx.x % kXcds;
        first = xcd * per_xcd + blockIdx.x / kXcds;
        step = gridDim.x / kXcds;
        end = (xcd + 1) * per_xcd < tiles ? (xcd + 1) * per_xcd : tiles;
    }
    double en[3] = {0.0, 0.0, 0.0};
    for (unsigned u = first; u < end; u += step) {  // block-uniform bounds: every thread meets every barrier
        const long long k = (long long)u * per_tile - c + t;
        const bool valid = k >= 0 && k < (long long)band_count;
        const bool owner = valid && t >= c && t < kBlock - c;
        int i = -0x40000000, x = 0, y = 0, z = 0;  // an index no tap address (i + d >= -kMaxTaps / 2) can equal
        vf4 raw = {0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) {
            i = band_list[k];
            decode_listed(g, (unsigned)i, x, y, z);
            float gv[3];
            double e[3];
            listed_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(state, canonical, g, p, i, x, y, z, gv, e);
            if (owner && z >= g.e_begin && z < g.e_end) {
                en[0] += e[0];
                en[1] += e[1];
                en[2] += e[2];
            }
            raw.x = gv[0]; raw.y = gv[1]; raw.z = D == 3 ? gv[2] : 0.0f;
        }
        s_index[t] = i;
        s_raw[t] = raw;
        __syncthreads();
        if (owner) {
            double acc[3] = {0.0, 0.0, 0.0};
            // tap j = c - d reads voxel i + d: j ascending = d from +c down to -c (filtered_at's order)
#pragma unroll
            for (int d = kMaxTaps / 2; d >= -(kMaxTaps / 2); --d) {
                if (d <= c && d >= -c) {
                    const int q = x + d;
                    vf4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (d == 0) {
                        v = raw;
                    } else if (q >= 0 && q < g.nx) {
                        // voxel i + d of the same row, if listed, sits 1 .. |d| entries away (the list ascends)
                        const int sign = d > 0 ? 1 : -1, reach = d > 0 ? d : -d;
                        int at = -1;
#pragma unroll
                        for (int s = 1; s <= reach; ++s)
                            if (s_index[t + sign * s] == i + d) at = t + sign * s;
                        if (at >= 0) v = s_raw[at];
                    }
                    const double k = taps.k[c - d];
                    acc[0] = mac<FMA>(acc[0], k, (double)v.x);
                    acc[1] = mac<FMA>(acc[1], k, (double)v.y);
                    acc[2] = mac<FMA>(acc[2], k, (double)v.z);
                }
            }
            // (8: a listed voxel -- what lsf_sobolev_state_update_boxes tells from the zeros nobody ever wrote)
            const unsigned bits = (fabsf(raw.x) < 1e-6f ? 1u : 0u) | (fabsf(raw.y) < 1e-6f ? 2u : 0u) |
                                  (fabsf(raw.z) < 1e-6f ? 4u : 0u) | 8u;
            vf4 o;
            o.x = (bits & 1u) ? 0.0f : (float)acc[0];
            o.y = (bits & 2u) ? 0.0f : (float)acc[1];
            o.z = (bits & 4u) ? 0.0f : (float)acc[2];
            o.w = __uint_as_float(bits);
            // bricks: the output in boxes of 4 x 4 x 4 voxels, 1 KB each, box (bz, by, bx) at ((bz * ny/4 + by) * nx/4 + bx)
            // * 64, voxel (lz, ly, lx) of it at (lz * 4 + ly) * 4 + lx -- what lsf_sobolev_state_update_boxes stages from
            const long long at = bricks ? ((((long long)(z >> 2) * (g.ny >> 2) + (y >> 2)) * (g.nx >> 2) + (x >> 2)) << 6) +
                                              (((z & 3) << 4) | ((y & 3) << 2) | (x & 3))
                                        : (long long)i;
            out[at] = o;
        }
        __syncthreads();  // the next tile overwrites the LDS arrays
    }
    if (ENERGY != LSF_ENERGY_NONE) {
        double* dst[3] = {&record_slot(record)->data_energy, &record_slot(record)->smoothing_energy,
                          &record_slot(record)->level_set_energy};
        block_reduce_commit<3>(0ull, en, nullptr, dst);
    }
}

// out[a] = sum_j k[j] * in[a + c - j] per component, zero outside [0, len), float64 in tap order, one float32 rounding,
// forced to 0 where the RAW gradient's component is below 1e-6 (math_utils/convolution.py:118,123,127)
// The mask travels WITH the data from the second pass on: the first pass (mask_src = the raw gradient, which is also its
// input) leaves the three verdicts as bits in the unused fourth component of its output, every later pass (mask_src ==
// nullptr) finds them in the centre tap it loads anyway and hands them on -- one 16-byte load per voxel and pass less, and
// 26 MB per pass at 256^3 that no longer come through the fabric (profiles/r04_sobolev_pmc_hbm_traffic.csv).
template <int NT, bool FMA>
__device__ inline vf4 filtered_at(const vf4* __restrict__ in, const vf4* __restrict__ mask_src, const TapsN<NT>& taps,
                                  unsigned i, int a, int len, int stride) {
    constexpr int c = NT / 2;
    vf4 m = {0.0f, 0.0f, 0.0f, 0.0f};
    if (mask_src) m = mask_src[i];
    vf4 v[NT];
    const vf4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int d = c - j, q = a + d;
        const bool inside = q >= 0 && q < len;
        const vf4 t = in[(long long)i + (inside ? d * stride : 0)];
        v[j] = inside ? t : zero;
    }
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        acc[0] = mac<FMA>(acc[0], taps.k[j], (double)v[j].x);
        acc[1] = mac<FMA>(acc[1], taps.k[j], (double)v[j].y);
        acc[2] = mac<FMA>(acc[2], taps.k[j], (double)v[j].z);
    }
    unsigned bits;
    if (mask_src) bits = (fabsf(m.x) < 1e-6f ? 1u : 0u) | (fabsf(m.y) < 1e-6f ? 2u : 0u) | (fabsf(m.z) < 1e-6f ? 4u : 0u) | 8u;
    else bits = __float_as_uint(v[c].w);  // the centre tap of the previous pass's output
    vf4 o;
    o.x = (bits & 1u) ? 0.0f : (float)acc[0];
    o.y = (bits & 2u) ? 0.0f : (float)acc[1];
    o.z = (bits & 4u) ? 0.0f : (float)acc[2];
    o.w = __uint_as_float(bits);
    return o;
}


// one zero-preserving pass at listed voxels (one thread per voxel; convolve_list_kernel on the float4 layout)
template <int NT, bool FMA>
__global__ __launch_bounds__(kBlock) void convolve_list4_kernel(const vf4* __restrict__ in, vf4* __restrict__ out,
                                                                const vf4* __restrict__ mask_src, Grid g,
                                                                TapsN<NT> taps, int axis, const int* __restrict__ list,
                                                                unsigned count, lsf_gate gate) {
    if (gate_closed(gate)) return;
    // XCD-aware tile order (blocks b, b + 8, ... share an XCD under round-robin dispatch): XCD k takes the k-th eighth of
    // the list -- a contiguous z-range --, so that the y -/+ 3 rows a block reads are rows its own XCD's other blocks read
    // or wrote into the same L2.  In plain block order the seven rows of the y pass came from seven other XCDs' L2s, i.e.
    // through the fabric: 143 MB per launch against 79 MB for the x pass (profiles/r04_sobolev_pmc_hbm_traffic.csv)
    unsigned tile = blockIdx.x;
    const unsigned tiles = gridDim.x;
    if (tiles >= 64u) {
        const unsigned per_xcd = (tiles + kXcds - 1) / kXcds, xcd = blockIdx.x % kXcds;
        tile = xcd * per_xcd + blockIdx.x / kXcds;
        if (tile >= tiles || blockIdx.x / kXcds >= per_xcd) return;  // grid padded to a multiple of 8 (see launch_pass4)
    }
    const unsigned k = tile * kBlock + threadIdx.x;
    if (k >= count) return;
    const unsigned i = (unsigned)list[k];
    int x, y, z;
    decode_listed(g, i, x, y, z);
    const int a = axis == 0 ? x : (axis == 1 ? y : z);
    const int len = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
    const int stride = axis == 0 ? 1 : (axis == 1 ? g.nx : g.nx * g.ny);
    out[i] = filtered_at<NT, FMA>(in, mask_src, taps, i, a, len, stride);
}

// the LAST pass + warp = -g * rate + the truncation-aware re-warp (slavcheva_optimizer2d.py:208-236,
// field_warping.py:112-151): state' = (re-warped live, warp), g_out4 = the final gradient (zeroed where the live value
// snapped, DIRECT only), record max = the longest update
template <int D, int NT, bool FMA>
__global__ __launch_bounds__(kBlock) void sobolev_state_update_kernel(const vf4* __restrict__ in,
                                                                      const vf4* __restrict__ mask_src,
                                                                      const vf4* __restrict__ state_in,
                                                                      vf4* __restrict__ state_out, vf4* __restrict__ g_out,
                                                                      Grid g, Params p, TapsN<NT> taps, int axis,
                                                                      lsf_gate gate, lsf_iteration_record* record,
                                                                      const int* __restrict__ band_list,
                                                                      unsigned band_count, int first_list) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    const int len = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
    const int stride = axis == 0 ? 1 : (axis == 1 ? g.nx : g.nx * g.ny);
    for_each_listed_voxel(g, band_list, band_count, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        const int a = axis == 0 ? x : (axis == 1 ? y : z);
        const vf4 gf = filtered_at<NT, FMA>(in, mask_src, taps, (unsigned)i, a, len, stride);
        float gv[3] = {gf.x, gf.y, D == 3 ? gf.z : 0.0f};
        float wv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < D; ++c) wv[c] = (-gv[c]) * p.rate;
        const float len_w = vec_length<D>(wv);
        const float l = state_in[i].x;
        float v;
        if (wv[0] == 0.0f && wv[1] == 0.0f && wv[2] == 0.0f) {
            v = l;  // zero displacement: every lerp is a * 1 + b * 0 = a exactly, the gather returns live[p] bit for bit
        } else {
            v = state_gather<D>(state_in, g, (float)x + wv[0], (float)y + wv[1],
                                D == 3 ? (float)(z + g.z_global_offset) + wv[2] : 0.0f);
        }
        if (1.0f - fabsf(v) < 1e-6f) {  // field_warping.py:138-141
            v = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v);
            wv[0] = wv[1] = wv[2] = 0.0f;
            if (p.zero_gradient_on_snap) gv[0] = gv[1] = gv[2] = 0.0f;
        }
        vf4 o, go;
        o.x = v; o.y = wv[0]; o.z = wv[1]; o.w = wv[2];
        go.x = gv[0]; go.y = gv[1]; go.z = gv[2]; go.w = 0.0f;
        state_out[i] = o;
        if (g_out) g_out[i] = go;  // null: nobody will read this iteration's gradient (16 B per voxel less to write)
        const unsigned long long q = pack_max(len_w, linear_index(g, x, y, z));
        best = q > best ? q : best;
    });
    if (first_list && blockIdx.x == 0 && threadIdx.x == 0) {  // the unlisted voxels: zero update, smallest index
        const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
        best = q > best ? q : best;
    }
    const double sums[1] = {0.0};
    double* dst[1] = {nullptr};
    block_reduce_commit<0>(best, sums, record_max(record), dst);
}

Params params_of(const lsf_slavcheva_params* q) {
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

// the state kernels' addressing limits (lsf_slavcheva_state_iteration)
Grid state_grid(const lsf_grid* grid) {
    Grid g = make_grid(grid, 4);
    g.fast_ok = g.plane * 16 < 0xffffffffll;
    g.wide_ok = 16ll * (2ll * grid->nx * grid->ny + 2ll * grid->nx + 3) < 0x7fffffffll;
    return g;
}

struct GradArgs {
    unsigned blocks;
    hipStream_t s;
    const vf4* state;
    const float* canonical;
    vf4* g_raw;  // the raw gradient -- or, with n_taps > 0, the gradient after the x pass (sobolev_state_gradient_x_kernel)
    Grid g;
    Params p;
    lsf_gate gate;
    lsf_iteration_record* record;
    const int* list;
    unsigned count;
    TapsN<kMaxTaps> taps;
    int n_taps;  // 0: gradient only
    bool fma;
    int bricks;  // the fused gradient + x pass writes its output in boxes of 4 x 4 x 4 voxels
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
void grad_one(const GradArgs& a) {
    if (a.n_taps == 0) {
        hipLaunchKernelGGL((sobolev_state_gradient_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY>), dim3(a.blocks), dim3(kBlock),
                           0, a.s, a.state, a.canonical, a.g_raw, a.g, a.p, a.gate, a.record, a.list, a.count);
        return;
    }
    if constexpr (D == 3) {
        if (a.fma)
            hipLaunchKernelGGL((sobolev_state_gradient_x_kernel<3, SMOOTH, LEVELSET, DATA, ENERGY, true>), dim3(a.blocks),
                               dim3(kBlock), 0, a.s, a.state, a.canonical, a.g_raw, a.g, a.p, a.taps, a.n_taps, a.bricks,
                               a.gate, a.record, a.list, a.count);
        else
            hipLaunchKernelGGL((sobolev_state_gradient_x_kernel<3, SMOOTH, LEVELSET, DATA, ENERGY, false>), dim3(a.blocks),
                               dim3(kBlock), 0, a.s, a.state, a.canonical, a.g_raw, a.g, a.p, a.taps, a.n_taps, a.bricks,
                               a.gate, a.record, a.list, a.count);
    }
}

template <int D, int SMOOTH, bool LEVELSET, int DATA>
void grad_energy(int energy, const GradArgs& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: grad_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT>(a); break;
        case LSF_ENERGY_VECTORIZED: grad_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED>(a); break;
        default: grad_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE>(a); break;
    }
}

template <int D>
void grad_terms(const lsf_slavcheva_params* q, const GradArgs& a) {
    const bool killing = q->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = q->level_set_enabled != 0;
    const bool fdm = q->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = q->energy_mode;
#define LSF_PICK(S, L, DM) grad_energy<D, S, L, DM>(e, a)
    if (killing) {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_BASIC); }
    } else {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_BASIC); }
    }
#undef LSF_PICK
}

inline bool taps_ok(int32_t n) { return n == 3 || n == 5 || n == 7 || n == 9; }

template <int NT>
void launch_pass4(const vf4* in, vf4* out, const vf4* mask, const Grid& g, int axis, const double* taps_host,
                  const int* list, unsigned count, const lsf_gate& gt, hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    unsigned tiles = (count + kBlock - 1) / kBlock;
    if (tiles >= 64u) tiles = (tiles + kXcds - 1) / kXcds * kXcds;  // every XCD the same number of blocks (the kernel's tile order)
    const dim3 grid(tiles);
    if (taps_are_float32(taps_host, NT))
        hipLaunchKernelGGL((convolve_list4_kernel<NT, true>), grid, dim3(kBlock), 0, s, in, out, mask, g, taps, axis, list,
                           count, gt);
    else
        hipLaunchKernelGGL((convolve_list4_kernel<NT, false>), grid, dim3(kBlock), 0, s, in, out, mask, g, taps, axis, list,
                           count, gt);
}

template <int D, int NT>
void launch_update4(const vf4* in, const vf4* mask, const vf4* state_in, vf4* state_out, vf4* g_out, const Grid& g,
                    const Params& p, int axis, const double* taps_host, const lsf_gate& gt, lsf_iteration_record* record,
                    const int* list, unsigned count, int first_list, hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    const dim3 grid(band_list_blocks(count));
    if (taps_are_float32(taps_host, NT))
        hipLaunchKernelGGL((sobolev_state_update_kernel<D, NT, true>), grid, dim3(kBlock), 0, s, in, mask, state_in,
                           state_out, g_out, g, p, taps, axis, gt, record, list, count, first_list);
    else
        hipLaunchKernelGGL((sobolev_state_update_kernel<D, NT, false>), grid, dim3(kBlock), 0, s, in, mask, state_in,
                           state_out, g_out, g, p, taps, axis, gt, record, list, count, first_list);
}

}  // namespace

// ---- the ascending band list regrouped STRIP by strip (the z pass's walk order, see lsf_band_list_strip_major) ----------
// In the ascending list the entries of one (slice z, strip of rows s) are ONE run (rows ascend inside a slice): run
// boundaries by binary search, an exclusive scan of the run lengths in strip-major order, and a gather.
__device__ inline int first_not_below(const int* __restrict__ list, unsigned count, long long key) {
    unsigned lo = 0u, hi = count;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if ((long long)list[mid] < key) lo = mid + 1u;
        else hi = mid;
    }
    return (int)lo;
}

__global__ __launch_bounds__(kBlock) void strip_runs_kernel(const int* __restrict__ list, unsigned count, int nx, int ny,
                                                            int nz, int rows, int n_strips, int* __restrict__ start,
                                                            int* __restrict__ length) {
    const int r = blockIdx.x * kBlock + threadIdx.x;  // run index in strip-major order: r = s * nz + z
    if (r >= n_strips * nz) return;
    const int s = r / nz, z = r - s * nz;
    const int row0 = min(s * rows, ny), row1 = min((s + 1) * rows, ny);
    const int lo = first_not_below(list, count, ((long long)z * ny + row0) * nx);
    const int hi = first_not_below(list, count, ((long long)z * ny + row1) * nx);
    start[r] = lo;
    length[r] = hi - lo;
}

// exclusive scan of m run lengths by ONE workgroup (m = strips * nz: a few thousand)
__global__ __launch_bounds__(1024) void strip_scan_kernel(const int* __restrict__ length, int* __restrict__ first, int m) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (m + 1023) / 1024;
    int sum = 0;
    for (int k = t * per; k < min((t + 1) * per, m); ++k) sum += length[k];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele over the 1024 partial sums
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int at = t ? part[t - 1] : 0;
    for (int k = t * per; k < min((t + 1) * per, m); ++k) {
        first[k] = at;
        at += length[k];
    }
}

__global__ __launch_bounds__(kBlock) void strip_gather_kernel(const int* __restrict__ list, const int* __restrict__ start,
                                                              const int* __restrict__ first, int m, unsigned count,
                                                              int* __restrict__ out) {
    const unsigned p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= count) return;
    // the LAST run whose first output position is <= p (empty runs share their successor's position and are skipped)
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (first[mid] <= (int)p) lo = mid + 1;
        else hi = mid;
    }
    const int r = lo - 1;
    out[p] = list[start[r] + ((int)p - first[r])];
}

extern "C" int lsf_band_list_strip_major(const int32_t* band_list, int64_t band_count, const lsf_grid* grid, int32_t strips,
                                         int32_t* out, int32_t* scratch, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!band_list || !out || !scratch || out == band_list || band_count < 0 || band_count > 0x7fffffffll || strips < 1 ||
        strips > 64 || grid->dims != 3)
        return LSF_ERR_BAD_ARGUMENT;
    if (band_count == 0) return 0;
    const int rows = (grid->ny + strips - 1) / strips, n_strips = (grid->ny + rows - 1) / rows;
    const long long m = (long long)n_strips * grid->nz;
    if (m > (1 << 20)) return LSF_ERR_BAD_ARGUMENT;
    int* start = scratch;
    int* length = scratch + m;
    int* first = scratch + 2 * m;
    hipStream_t s = as_stream(stream);
    const unsigned count = (unsigned)band_count;
    hipLaunchKernelGGL(strip_runs_kernel, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, band_list, count,
                       grid->nx, grid->ny, grid->nz, rows, n_strips, start, length);
    hipLaunchKernelGGL(strip_scan_kernel, dim3(1), dim3(1024), 0, s, length, first, (int)m);
    hipLaunchKernelGGL(strip_gather_kernel, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, band_list, start, first,
                       (int)m, count, out);
    return launch_status();
}

extern "C" int lsf_sobolev_state_gradient(const float* state, const float* canonical, float* g_raw4,
                                          const lsf_grid* grid, const lsf_slavcheva_params* params, const lsf_gate* gate,
                                          lsf_iteration_record* record, const int32_t* band_list, int64_t band_count,
                                          void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state || !canonical || !g_raw4 || !params || !record || !band_list || band_count < 0 ||
        band_count > 0x7fffffffll)
        return LSF_ERR_BAD_ARGUMENT;
    const Grid g = state_grid(grid);
    if (g.z_end == g.z_begin || band_count == 0) return 0;
    GradArgs a{band_list_blocks((unsigned)band_count), as_stream(stream), reinterpret_cast<const vf4*>(state), canonical,
               reinterpret_cast<vf4*>(g_raw4), g, params_of(params), gate_or_open(gate), record, band_list,
               (unsigned)band_count, TapsN<kMaxTaps>(), 0, false, 0};
    if (grid->dims == 2) grad_terms<2>(params, a);
    else grad_terms<3>(params, a);
    return launch_status();
}

extern "C" int lsf_sobolev_state_gradient_x(const float* state, const float* canonical, float* out4, const lsf_grid* grid,
                                            const lsf_slavcheva_params* params, const double* taps_host, int32_t n_taps,
                                            const lsf_gate* gate, lsf_iteration_record* record, const int32_t* band_list,
                                            int64_t band_count, int32_t out_bricks, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state || !canonical || !out4 || !params || !record || !band_list || !taps_host || band_count < 0 ||
        band_count > 0x7fffffffll || grid->dims != 3)
        return LSF_ERR_BAD_ARGUMENT;
    if (out_bricks && (grid->nx % 4 || grid->ny % 4 || grid->nz % 4)) return LSF_ERR_BAD_DIMS;
    if (!taps_ok(n_taps)) return LSF_ERR_KERNEL_TOO_LONG;
    const Grid g = state_grid(grid);
    if (g.z_end == g.z_begin || band_count == 0) return 0;
    GradArgs a{band_list_blocks((unsigned)band_count, kGradientXBlocksPerXcd), as_stream(stream), reinterpret_cast<const vf4*>(state), canonical,
               reinterpret_cast<vf4*>(out4), g, params_of(params), gate_or_open(gate), record, band_list,
               (unsigned)band_count, TapsN<kMaxTaps>(), n_taps, taps_are_float32(taps_host, n_taps), out_bricks != 0};
    for (int j = 0; j < kMaxTaps; ++j) a.taps.k[j] = j < n_taps ? taps_host[j] : 0.0;
    grad_terms<3>(params, a);
    return launch_status();
}

// zeros at the voxels of a band list (natural layout or the bricks of lsf_sobolev_state_gradient_x): what makes a gradient
// buffer of the PREVIOUS call usable again -- every other voxel of it is zero already (268 MB of zero fill per buffer and call
// at 256^3 against 26 MB of listed voxels)
__global__ __launch_bounds__(kBlock) void zero_listed4_kernel(vf4* __restrict__ field, const int* __restrict__ list,
                                                              unsigned count, Grid g, int bricks) {
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const int i = list[k];
    long long at = i;
    if (bricks) {
        int x, y, z;
        decode_listed(g, (unsigned)i, x, y, z);
        at = ((((long long)(z >> 2) * (g.ny >> 2) + (y >> 2)) * (g.nx >> 2) + (x >> 2)) << 6) + (((z & 3) << 4) | ((y & 3) << 2) | (x & 3));
    }
    const vf4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    field[at] = zero;
}

extern "C" int lsf_zero_listed4(float* field4, const lsf_grid* grid, const int32_t* band_list, int64_t band_count,
                                int32_t bricks, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!field4 || band_count < 0 || band_count > 0x7fffffffll || (band_count && !band_list)) return LSF_ERR_BAD_ARGUMENT;
    if (bricks && (grid->dims != 3 || grid->nx % 4 || grid->ny % 4 || grid->nz % 4)) return LSF_ERR_BAD_DIMS;
    if (band_count == 0) return 0;
    hipLaunchKernelGGL(zero_listed4_kernel, dim3((unsigned)((band_count + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       as_stream(stream), reinterpret_cast<vf4*>(field4), band_list, (unsigned)band_count, state_grid(grid),
                       bricks);
    return launch_status();
}

extern "C" int lsf_convolve_axis_listed4(const float* in4, float* out4, const float* zero_mask_source4,
                                         const lsf_grid* grid, int32_t axis, const double* taps_host, int32_t n_taps,
                                         const lsf_gate* gate, const int32_t* band_list, int64_t band_count,
                                         void* stream) {
    if (int e = check_grid(grid)) return e;
    // zero_mask_source4 == NULL: the mask bits are in in4's fourth component (a pass behind the first one)
    if (!in4 || !out4 || in4 == out4 || !taps_host || !band_list || band_count < 0 ||
        band_count > 0x7fffffffll || axis < 0 || axis >= grid->dims)
        return LSF_ERR_BAD_ARGUMENT;
    if (!taps_ok(n_taps)) return LSF_ERR_KERNEL_TOO_LONG;
    const Grid g = state_grid(grid);
    if (band_count == 0) return 0;
    const vf4* in = reinterpret_cast<const vf4*>(in4);
    vf4* out = reinterpret_cast<vf4*>(out4);
    const vf4* mask = reinterpret_cast<const vf4*>(zero_mask_source4);
    const lsf_gate gt = gate_or_open(gate);
    hipStream_t s = as_stream(stream);
    const unsigned count = (unsigned)band_count;
    switch (n_taps) {
        case 3: launch_pass4<3>(in, out, mask, g, axis, taps_host, band_list, count, gt, s); break;
        case 5: launch_pass4<5>(in, out, mask, g, axis, taps_host, band_list, count, gt, s); break;
        case 7: launch_pass4<7>(in, out, mask, g, axis, taps_host, band_list, count, gt, s); break;
        default: launch_pass4<9>(in, out, mask, g, axis, taps_host, band_list, count, gt, s); break;
    }
    return launch_status();
}

extern "C" int lsf_sobolev_state_update(const float* in4, const float* zero_mask_source4, const float* state_in,
                                        float* state_out, float* g_out4, const lsf_grid* grid,
                                        const lsf_slavcheva_params* params, int32_t axis, const double* taps_host,
                                        int32_t n_taps, const lsf_gate* gate, lsf_iteration_record* record,
                                        const int32_t* band_list, int64_t band_count, int32_t first_list, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in4 || !state_in || !state_out || state_out == state_in || g_out4 == in4 ||
        !params || !record || !taps_host || !band_list || band_count < 0 || band_count > 0x7fffffffll || axis < 0 ||
        axis >= grid->dims)
        return LSF_ERR_BAD_ARGUMENT;
    if (!taps_ok(n_taps)) return LSF_ERR_KERNEL_TOO_LONG;
    const Grid g = state_grid(grid);
    if (g.z_end == g.z_begin) return 0;
    const vf4* in = reinterpret_cast<const vf4*>(in4);
    const vf4* mask = reinterpret_cast<const vf4*>(zero_mask_source4);
    const vf4* s_in = reinterpret_cast<const vf4*>(state_in);
    vf4* s_out = reinterpret_cast<vf4*>(state_out);
    vf4* g_out = reinterpret_cast<vf4*>(g_out4);
    const Params p = params_of(params);
    const lsf_gate gt = gate_or_open(gate);
    hipStream_t s = as_stream(stream);
    const unsigned count = (unsigned)band_count;
#define LSF_UPD(D, NT) launch_update4<D, NT>(in, mask, s_in, s_out, g_out, g, p, axis, taps_host, gt, record, band_list, \
                                             count, first_list, s)
    if (grid->dims == 2) {
        switch (n_taps) { case 3: LSF_UPD(2, 3); break; case 5: LSF_UPD(2, 5); break; case 7: LSF_UPD(2, 7); break; default: LSF_UPD(2, 9); break; }
    } else {
        switch (n_taps) { case 3: LSF_UPD(3, 3); break; case 5: LSF_UPD(3, 5); break; case 7: LSF_UPD(3, 7); break; default: LSF_UPD(3, 9); break; }
    }
#undef LSF_UPD
    return launch_status();
}
