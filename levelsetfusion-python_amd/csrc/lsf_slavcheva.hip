// Slavcheva-style (KillingFusion / SobolevFusion) optimizer iteration kernels, D = 2, 3
// (SURVEY 8a rows a3, a12-a18).  Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:163-330 with
// data_term.py, smoothing_term.py, level_set_term.py and field_warping.warp_field_advanced.
//
// FUSED stage (no Sobolev filter between gradient and update): ONE kernel per iteration reads
//   live (4 B, 3^D-neighbourhood through cache), canonical (4 B), previous warp (4D B, neighbourhood)
// and writes the new warp (4D B) and the re-warped live field (4 B): 36 B / voxel-update of compulsory HBM
// traffic in 3-D (SURVEY 8d books 52 B for the two-pass formulation).  The max-warp arg-max and the three
// energies are wave-shuffle + LDS block reductions ending in one atomic per block.
#include "lsf_device.h"

using namespace lsf;

namespace {

struct Params {
    double lambda64;
    float rate, w_data, w_smooth, w_level_set, lambda32, killing_c1;
    int zero_gradient_on_snap;
};

__device__ inline float np_gradient_at(const float* __restrict__ f, long long i, int coord, int n, long long stride) {
    if (n == 1) return 0.0f;
    if (coord == 0) return f[i + stride] - f[i];
    if (coord == n - 1) return f[i] - f[i - stride];
    return (f[i + stride] - f[i - stride]) * 0.5f;
}

// reads of the previous warp with "out of bounds -> centre value" (utils/sampling.py:84-88 called with
// replacement = warp[y, x], smoothing_term.py:60-63)
struct WarpReader {
    const float* __restrict__ w;
    const Grid& g;
    int x, y, z;
    const float* centre;  // [3]
    __device__ inline float at(int c, int dx, int dy, int dz) const {
        int xx = x + dx, yy = y + dy, zz = z + dz;
        return inside(g, xx, yy, zz) ? w[c * g.plane + vidx(g, xx, yy, zz)] : centre[c];
    }
};

template <int D>
__device__ inline void axis_step(int a, int s, int& dx, int& dy, int& dz) {
    dx = a == 0 ? s : 0;
    dy = a == 1 ? s : 0;
    dz = a == 2 ? s : 0;
}

// a14 (vectorised form used for both compute methods): -Laplacian, edge replicated, scipy rounding
template <int D>
__device__ inline void tikhonov_gradient(const WarpReader& r, float (&gs)[3]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float a0 = r.centre[c];
        float d2y = second_difference_f64(r.at(c, 0, -1, 0), a0, r.at(c, 0, 1, 0));
        float d2x = second_difference_f64(r.at(c, -1, 0, 0), a0, r.at(c, 1, 0, 0));
        float lap;
        if (D == 3) {
            float d2z = second_difference_f64(r.at(c, 0, 0, -1), a0, r.at(c, 0, 0, 1));
            lap = (d2z + d2y) + d2x;
        } else {
            lap = d2y + d2x;
        }
        gs[c] = -lap;
    }
}

// a15: Killing regulariser, smoothing_term.py:50-100, every quirk kept (w_yy uses the +1 neighbour twice; the
// -2(1+lambda) factor multiplies the xx term only); 3-D extension per DESIGN.md section 3.
template <int D>
__device__ inline void killing_gradient(const WarpReader& r, const Params& p, float (&gs)[3], double& energy,
                                        bool want_energy) {
    float first[3][3];   // first[a][i]  = d w_i / d a
    float second[3][3];  // second[a][i] = d2 w_i / d a2 (quirky for a == y)
    float cross[3][3];   // cross[k][i], k = 0:(x,y) 1:(x,z) 2:(y,z)
#pragma unroll
    for (int a = 0; a < D; ++a) {
        int dx, dy, dz;
        axis_step<D>(a, 1, dx, dy, dz);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float pl = r.at(i, dx, dy, dz), mi = r.at(i, -dx, -dy, -dz);
            first[a][i] = 0.5f * (pl - mi);
            float t = pl - 2.0f * r.centre[i];
            second[a][i] = a == 1 ? t + pl : t + mi;
        }
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a + 1; b < D; ++b) {
            const int k = a + b - 1;
            int ax, ay, az, bx, by, bz;
            axis_step<D>(a, 1, ax, ay, az);
            axis_step<D>(b, 1, bx, by, bz);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float pp = r.at(i, ax + bx, ay + by, az + bz);
                float pm = r.at(i, ax - bx, ay - by, az - bz);
                float mp = r.at(i, -ax + bx, -ay + by, -az + bz);
                float mm = r.at(i, -ax - bx, -ay - by, -az - bz);
                cross[k][i] = (((pp - pm) - mp) + mm) / 4.0f;
            }
        }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float g = p.killing_c1 * second[0][i];
#pragma unroll
        for (int a = 1; a < D; ++a) g = g + second[a][i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j == i) continue;
            const int k = i + j - 1;
            g = g + p.lambda32 * cross[k][j];
        }
        gs[i] = g;
    }
    if (want_energy) {
        double e = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                double jic = (double)first[c][i], jci = (double)first[i][c];
                e += jic * jic + p.lambda64 * jic * jci;
            }
        energy = e;
    }
}

// a16: level-set term, level_set_term.py:28-64 (OOB -> 1; second derivatives use the +1 neighbour twice)
template <int D>
__device__ inline void level_set_gradient(const float* __restrict__ live, const Grid& g, int x, int y, int z,
                                          float l, float (&gl)[3], double& energy) {
    float grad[3] = {0.0f, 0.0f, 0.0f};
    float hess[3][3];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        int dx, dy, dz;
        axis_step<D>(c, 1, dx, dy, dz);
        float pl = read_oob(live, g, x + dx, y + dy, z + dz, 1.0f);
        float mi = read_oob(live, g, x - dx, y - dy, z - dz, 1.0f);
        grad[c] = (0.5f * (pl - mi)) * 10.0f;
        hess[c][c] = ((pl - 2.0f * l) + pl) * 10.0f;
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a + 1; b < D; ++b) {
            int ax, ay, az, bx, by, bz;
            axis_step<D>(a, 1, ax, ay, az);
            axis_step<D>(b, 1, bx, by, bz);
            float pp = read_oob(live, g, x + ax + bx, y + ay + by, z + az + bz, 1.0f);
            float mp = read_oob(live, g, x - ax + bx, y - ay + by, z - az + bz, 1.0f);
            float pm = read_oob(live, g, x + ax - bx, y + ay - by, z + az - bz, 1.0f);
            float mm = read_oob(live, g, x - ax - bx, y - ay - by, z - az - bz, 1.0f);
            float s = (a == 0 && b == 1) ? ((pp - mp) - pm) + mm   // level_set_term.py:52-53
                                         : ((pp - pm) - mp) + mm;  // pairs with z: z difference first
            float h = (0.25f * s) * 10.0f;
            hess[a][b] = h;
            hess[b][a] = h;
        }
    float sq = grad[0] * grad[0];
#pragma unroll
    for (int c = 1; c < D; ++c) sq = sq + grad[c] * grad[c];
    const float n = sqrtf(sq);
    const float coef = (1.0f - n) / (n + 1e-5f);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float hv = hess[i][0] * grad[0];
#pragma unroll
        for (int j = 1; j < D; ++j) hv = hv + hess[i][j] * grad[j];
        gl[i] = coef * hv;
    }
    const double dn = (double)n - 1.0;
    energy = 0.5 * dn * dn;
}

// gradient of the energy at one voxel (a12-a17); returns false when the voxel is outside the band union
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__device__ inline void voxel_gradient(const float* __restrict__ live, const float* __restrict__ canonical,
                                      const float* __restrict__ warp_prev, const Grid& g, const Params& p, int x,
                                      int y, int z, long long i, float (&gv)[3], double (&en)[3]) {
    gv[0] = gv[1] = gv[2] = 0.0f;
    const float l = live[i], cn = canonical[i];
    const bool live_truncated = fabsf(l) == 1.0f;
    if (live_truncated && fabsf(cn) == 1.0f) return;  // outside the narrow-band union (tsdf_set_routines.py:19-52)
    const long long sy = g.nx, sz = (long long)g.nx * g.ny;
    // ---- data term (data_term.py:169-187 / :334-349; thresholded variant :190-227)
    const float diff = l - cn;
    float lg[3];
    lg[0] = np_gradient_at(live, i, x, g.nx, 1);
    lg[1] = np_gradient_at(live, i, y, g.ny, sy);
    lg[2] = D == 3 ? np_gradient_at(live, i, z, g.nz, sz) : 0.0f;
    if (DATA == LSF_DATA_THRESHOLDED_FDM) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            if (fabsf(lg[c]) > 0.5f) {
                int dx, dy, dz;
                axis_step<D>(c, 1, dx, dy, dz);
                float fwd = read_oob(live, g, x + dx, y + dy, z + dz, 1.0f) - l;
                float bwd = l - read_oob(live, g, x - dx, y - dy, z - dz, 1.0f);
                float alt = fabsf(fwd) < fabsf(bwd) ? fwd : bwd;
                if (fabsf(alt) > 0.5f) alt = 0.0f;
                lg[c] = alt;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) gv[c] = p.w_data * ((diff * lg[c]) * 10.0f);
    if (ENERGY != LSF_ENERGY_NONE) en[0] = 0.5 * (double)diff * (double)diff;
    // ---- level-set term (DIRECT only; skipped where live is truncated, slavcheva_optimizer2d.py:274)
    if (LEVELSET && !live_truncated) {
        float gl[3];
        double e;
        level_set_gradient<D>(live, g, x, y, z, l, gl, e);
#pragma unroll
        for (int c = 0; c < D; ++c) gv[c] = gv[c] + p.w_level_set * gl[c];
        if (ENERGY != LSF_ENERGY_NONE) en[2] = e;
    }
    // ---- smoothing term
    float wc[3];
    wc[0] = warp_prev[i];
    wc[1] = warp_prev[g.plane + i];
    wc[2] = D == 3 ? warp_prev[2 * g.plane + i] : 0.0f;
    WarpReader r{warp_prev, g, x, y, z, wc};
    float gs[3] = {0.0f, 0.0f, 0.0f};
    if (SMOOTH == LSF_SMOOTHING_KILLING) {
        double e = 0.0;
        killing_gradient<D>(r, p, gs, e, ENERGY != LSF_ENERGY_NONE);
        if (ENERGY != LSF_ENERGY_NONE) en[1] = e;
    } else {
        tikhonov_gradient<D>(r, gs);
        if (ENERGY == LSF_ENERGY_DIRECT) {
            // smoothing_term.py:134-139: 0.5 * sum_axis |0.5 (w[+1] - w[-1])|^2, OOB -> centre
            double e = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                int dx, dy, dz;
                axis_step<D>(a, 1, dx, dy, dz);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    float der = 0.5f * (r.at(c, dx, dy, dz) - r.at(c, -dx, -dy, -dz));
                    e += (double)der * (double)der;
                }
            }
            en[1] = 0.5 * e;
        } else if (ENERGY == LSF_ENERGY_VECTORIZED) {
            // smoothing_term.py:162-177: 0.5 * sum_{c,axis} np.gradient(warp_c)[axis]^2 over the band
            double e = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const float* wp = warp_prev + c * g.plane;
                float d0 = np_gradient_at(wp, i, x, g.nx, 1);
                float d1 = np_gradient_at(wp, i, y, g.ny, sy);
                float d2 = D == 3 ? np_gradient_at(wp, i, z, g.nz, sz) : 0.0f;
                e += (double)d0 * (double)d0 + (double)d1 * (double)d1 + (double)d2 * (double)d2;
            }
            en[1] = 0.5 * e;
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) gv[c] = gv[c] + p.w_smooth * gs[c];
}

// warp = -g*rate, |warp| for the arg-max, truncation-aware re-warp of the live field (a18 + a3)
template <int D>
__device__ inline unsigned long long update_and_rewarp(const float* __restrict__ live, const Grid& g,
                                                       const Params& p, int x, int y, int z, long long i,
                                                       float (&gv)[3], float* __restrict__ warp_out,
                                                       float* __restrict__ live_out, float* __restrict__ g_out) {
    float wv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < D; ++c) wv[c] = (-gv[c]) * p.rate;
    const float len = vec_length<D>(wv);
    const unsigned lin = linear_index(g, x, y, z);
    const float px = (float)x + wv[0], py = (float)y + wv[1];
    const float pz = D == 3 ? (float)(z + g.z_global_offset) + wv[2] : 0.0f;
    float v = sample_linear<D>(live, g, px, py, pz, 1.0f);
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

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, bool FUSED>
__global__ __launch_bounds__(kBlock) void slavcheva_iteration_kernel(
    const float* __restrict__ live, const float* __restrict__ canonical, const float* __restrict__ warp_prev,
    float* __restrict__ warp_out, float* __restrict__ live_out, float* __restrict__ g_out, Grid g, Params p,
    lsf_gate gate, lsf_iteration_record* record) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    double en[3] = {0.0, 0.0, 0.0};
    for_each_voxel(g, [&](int x, int y, int z) {
        const long long i = vidx(g, x, y, z);
        float gv[3];
        double e[3] = {0.0, 0.0, 0.0};
        voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(live, canonical, warp_prev, g, p, x, y, z, i, gv, e);
        en[0] += e[0];
        en[1] += e[1];
        en[2] += e[2];
        if (FUSED) {
            unsigned long long q = update_and_rewarp<D>(live, g, p, x, y, z, i, gv, warp_out, live_out, g_out);
            best = q > best ? q : best;
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) g_out[c * g.plane + i] = gv[c];
        }
    });
    if (FUSED || ENERGY != LSF_ENERGY_NONE) {
        double* dst[3] = {ENERGY != LSF_ENERGY_NONE ? &record->data_energy : nullptr,
                          ENERGY != LSF_ENERGY_NONE ? &record->smoothing_energy : nullptr,
                          ENERGY != LSF_ENERGY_NONE ? &record->level_set_energy : nullptr};
        block_reduce_commit<3>(best, en, FUSED ? record_max(record) : nullptr, dst);
    }
}

template <int D>
__global__ __launch_bounds__(kBlock) void slavcheva_update_rewarp_kernel(const float* __restrict__ live,
                                                                         float* __restrict__ gfield,
                                                                         float* __restrict__ warp_out,
                                                                         float* __restrict__ live_out, Grid g,
                                                                         Params p, lsf_gate gate,
                                                                         lsf_iteration_record* record) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    for_each_voxel(g, [&](int x, int y, int z) {
        const long long i = vidx(g, x, y, z);
        float gv[3] = {gfield[i], gfield[g.plane + i], D == 3 ? gfield[2 * g.plane + i] : 0.0f};
        unsigned long long q = update_and_rewarp<D>(live, g, p, x, y, z, i, gv, warp_out, live_out,
                                                    p.zero_gradient_on_snap ? gfield : nullptr);
        best = q > best ? q : best;
    });
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
    float *warp_out, *live_out, *g_out;
    Grid g;
    Params p;
    lsf_gate gate;
    lsf_iteration_record* record;
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, bool FUSED>
void launch_one(const LaunchArgs& a) {
    hipLaunchKernelGGL((slavcheva_iteration_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY, FUSED>), dim3(a.blocks),
                       dim3(kBlock), 0, a.s, a.live, a.canonical, a.warp_prev, a.warp_out, a.live_out, a.g_out, a.g,
                       a.p, a.gate, a.record);
}

template <int D, int SMOOTH, bool LEVELSET, int DATA, bool FUSED>
void pick_energy(int energy, const LaunchArgs& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT, FUSED>(a); break;
        case LSF_ENERGY_VECTORIZED: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED, FUSED>(a); break;
        default: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE, FUSED>(a); break;
    }
}

template <int D, bool FUSED>
void pick_terms(const lsf_slavcheva_params* q, const LaunchArgs& a) {
    const bool killing = q->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = q->level_set_enabled != 0;
    const bool fdm = q->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = q->energy_mode;
#define LSF_PICK(S, L, DM) pick_energy<D, S, L, DM, FUSED>(e, a)
    if (killing) {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_BASIC); }
    } else {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_BASIC); }
    }
#undef LSF_PICK
}

}  // namespace

extern "C" int lsf_slavcheva_iteration(int32_t stage, const float* live, const float* canonical,
                                       const float* warp_prev_planar, float* warp_out_planar, float* live_out,
                                       float* g_out_planar, const lsf_grid* grid, const lsf_slavcheva_params* params,
                                       const lsf_gate* gate, lsf_iteration_record* record, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !canonical || !warp_prev_planar || !params || !record) return LSF_ERR_BAD_ARGUMENT;
    if (stage == LSF_STAGE_FUSED) {
        if (!warp_out_planar || !live_out || live_out == live || warp_out_planar == warp_prev_planar)
            return LSF_ERR_BAD_ARGUMENT;
    } else if (stage == LSF_STAGE_GRADIENT) {
        if (!g_out_planar || g_out_planar == warp_prev_planar) return LSF_ERR_BAD_ARGUMENT;
    } else {
        return LSF_ERR_BAD_ARGUMENT;
    }
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    LaunchArgs a{launch_blocks(t.total), as_stream(stream), live, canonical, warp_prev_planar, warp_out_planar, live_out,
                 g_out_planar, g, make_params(params), gate_or_open(gate), record};
    if (grid->dims == 2) {
        if (stage == LSF_STAGE_FUSED) pick_terms<2, true>(params, a); else pick_terms<2, false>(params, a);
    } else {
        if (stage == LSF_STAGE_FUSED) pick_terms<3, true>(params, a); else pick_terms<3, false>(params, a);
    }
    return launch_status();
}

extern "C" int lsf_slavcheva_update_rewarp(const float* live, const float* canonical, float* g_planar,
                                           float* warp_out_planar, float* live_out, const lsf_grid* grid,
                                           const lsf_slavcheva_params* params, const lsf_gate* gate,
                                           lsf_iteration_record* record, void* stream) {
    (void)canonical;
    if (int e = check_grid(grid)) return e;
    if (!live || !g_planar || !warp_out_planar || !live_out || live_out == live || !params || !record)
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    Params p = make_params(params);
    lsf_gate gt = gate_or_open(gate);
    if (grid->dims == 2)
        hipLaunchKernelGGL(slavcheva_update_rewarp_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream), live,
                           g_planar, warp_out_planar, live_out, g, p, gt, record);
    else
        hipLaunchKernelGGL(slavcheva_update_rewarp_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream), live,
                           g_planar, warp_out_planar, live_out, g, p, gt, record);
    return launch_status();
}
