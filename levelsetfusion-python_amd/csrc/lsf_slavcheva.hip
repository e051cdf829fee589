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

// Neighbourhood addressing of one voxel.  Every neighbour is read from a CLAMPED offset (always a valid address,
// equal to the centre when the neighbour does not exist along that axis) and the reference's three different
// out-of-bounds rules are applied afterwards with selects -- the loads themselves are unconditional, so the
// compiler emits no exec-mask branches around them:
//   warp neighbours   OOB -> centre value  (utils/sampling.py:84-88 with replacement = warp[y, x])
//   level-set / FDM   OOB -> 1             (utils/sampling.py:35-55)
//   np.gradient       one-sided first-order difference at the array border
template <int D>
struct Nbh {
    using Field = const float*;  // base of one scalar plane
    int i;          // index of the voxel inside a plane
    int off[3][2];  // clamped element offsets of the -1 / +1 neighbours along x, y, z
    bool has[3][2]; // neighbour exists

    __device__ inline Nbh(const Grid& g, int x, int y, int z) {
        i = vidx(g, x, y, z);
        const int stride[3] = {1, g.nx, g.nx * g.ny};
        const int coord[3] = {x, y, z};
        const int extent[3] = {g.nx, g.ny, g.nz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            has[a][0] = a < D && coord[a] > 0;
            has[a][1] = a < D && coord[a] < extent[a] - 1;
            off[a][0] = has[a][0] ? -stride[a] : 0;
            off[a][1] = has[a][1] ? stride[a] : 0;
        }
    }
    __device__ static inline Field field(const float* base, const Grid& g, int plane) { return base + plane * g.plane; }
    __device__ inline bool exists(int a, int s) const { return has[a][s]; }
    __device__ inline float centre(Field f) const { return f[i]; }
    // axis neighbour of a scalar plane; clamped (== centre when missing)
    __device__ inline float axis(Field f, int a, int s) const { return f[i + off[a][s]]; }
    // diagonal neighbour in the (a, b) plane, sa/sb in {0: -1, 1: +1}; clamped
    __device__ inline float diag(Field f, int a, int sa, int b, int sb) const {
        return f[i + off[a][sa] + off[b][sb]];
    }
    __device__ inline bool diag_exists(int a, int sa, int b, int sb) const { return has[a][sa] && has[b][sb]; }
};

// The same interface for a voxel whose whole 3^D neighbourhood lies inside the array (decided per WAVE: all its
// active lanes).  Every neighbour exists, so the reference's OOB rules never fire, and the neighbour offsets are the
// same for all lanes: loads go through buffer resources with ONE per-lane byte offset (the neighbourhood's lowest
// corner) and the neighbour selected by the instruction's SCALAR offset operand -- no per-neighbour VALU address
// arithmetic, no selects.  (The generic path spends 107 of its ~950 VALU instructions per 64 voxels on 64-bit
// address adds and 100 on OOB selects.)  Buffer offsets are 32-bit: the host enables this path only when a scalar
// plane and the D planes of a vector field each stay below 4 GiB (Grid::fast_ok).
struct BufField {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned plane_bytes;  // byte offset of the addressed plane inside the resource
};

template <int D>
struct NbhFast {
    using Field = BufField;
    // byte offset of voxel (x-1, y-1[, z-1]) inside a plane = wave_base (scalar: the offset of the wave's first
    // active lane, which is its smallest -- both walks hand out ascending voxel indices by lane) + lane_delta.
    // Keeping the scalar part tied to the voxel makes every neighbour's scalar offset a one-instruction SALU add at
    // the point of use instead of ~70 loop-invariant values that would have to live in (spilled) SGPRs.
    unsigned wave_base, lane_delta;
    unsigned sy, sz;   // byte strides of y and z (uniform)

    __device__ inline NbhFast(const Grid& g, int x, int y, int z) {
        sy = (unsigned)g.nx * 4u;
        sz = (unsigned)(g.nx * g.ny) * 4u;
        const unsigned corner = (unsigned)vidx(g, x, y, z) * 4u - 4u - sy - (D == 3 ? sz : 0u);
        wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)corner);
        lane_delta = corner - wave_base;
    }
    __device__ static inline Field field(const float* base, const Grid& g, int plane) {
        // one resource spans all planes of the field (D * plane * 4 bytes < 4 GiB, see Grid::fast_ok)
        return BufField{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000),
                        (unsigned)plane * (unsigned)g.plane * 4u};
    }
    __device__ inline float at(const Field& f, int dx, int dy, int dz) const {
        const unsigned soff = wave_base + f.plane_bytes + (unsigned)(dy + 1) * sy + (D == 3 ? (unsigned)(dz + 1) * sz : 0u);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             f.rsrc, (int)(lane_delta + (unsigned)(dx + 1) * 4u), (int)soff, 0));
    }
    __device__ static constexpr bool exists(int, int) { return true; }
    __device__ static constexpr bool diag_exists(int, int, int, int) { return true; }
    __device__ inline float centre(const Field& f) const { return at(f, 0, 0, 0); }
    __device__ inline float axis(const Field& f, int a, int s) const {
        const int d = s ? 1 : -1;
        return at(f, a == 0 ? d : 0, a == 1 ? d : 0, a == 2 ? d : 0);
    }
    __device__ inline float diag(const Field& f, int a, int sa, int b, int sb) const {
        const int da = sa ? 1 : -1, db = sb ? 1 : -1;
        return at(f, (a == 0 ? da : 0) + (b == 0 ? db : 0), (a == 1 ? da : 0) + (b == 1 ? db : 0),
                  (a == 2 ? da : 0) + (b == 2 ? db : 0));
    }
};

// np.gradient along axis a from the clamped neighbours: (f+ - f-)/2 inside, one-sided at the border, 0 for n == 1
template <class NB>
__device__ inline float np_gradient_from(const NB& n, int a, float fm, float fp) {
    const float d = fp - fm;  // at a border the missing side was read as the centre
    return (n.exists(a, 0) && n.exists(a, 1)) ? d * 0.5f : d;
}

// a14 (vectorised form used for both compute methods): -Laplacian, edge replicated, scipy rounding
template <int D>
__device__ inline void tikhonov_gradient(const float (&wm)[3][3], const float (&wp)[3][3], const float (&wc)[3],
                                         float (&gs)[3]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float d2y = second_difference_f64(wm[1][c], wc[c], wp[1][c]);
        const float d2x = second_difference_f64(wm[0][c], wc[c], wp[0][c]);
        float lap;
        if (D == 3) {
            const float d2z = second_difference_f64(wm[2][c], wc[c], wp[2][c]);
            lap = (d2z + d2y) + d2x;
        } else {
            lap = d2y + d2x;
        }
        gs[c] = -lap;
    }
}

// a15: Killing regulariser, smoothing_term.py:50-100, every quirk kept (w_yy uses the +1 neighbour twice; the
// -2(1+lambda) factor multiplies the xx term only); 3-D extension per DESIGN.md section 3.
// wm/wp[a][i]: component i at the -1/+1 neighbour along axis a (missing neighbour = centre).
template <int D, class NB>
__device__ inline void killing_gradient(const NB& n, const typename NB::Field (&w)[3], const float (&wm)[3][3],
                                        const float (&wp)[3][3], const float (&wc)[3], const Params& p,
                                        float (&gs)[3], double& energy, bool want_energy) {
    float first[3][3];   // first[a][i]  = d w_i / d a
    float second[3][3];  // second[a][i] = d2 w_i / d a2 (quirky for a == y)
    float cross[3][3];   // cross[k][i], k = 0:(x,y) 1:(x,z) 2:(y,z)
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const float pl = wp[a][i], mi = wm[a][i];
            first[a][i] = 0.5f * (pl - mi);
            const float t = pl - 2.0f * wc[i];
            second[a][i] = a == 1 ? t + pl : t + mi;
        }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a + 1; b < D; ++b) {
            const int k = a + b - 1;
            const bool epp = n.diag_exists(a, 1, b, 1), epm = n.diag_exists(a, 1, b, 0);
            const bool emp = n.diag_exists(a, 0, b, 1), emm = n.diag_exists(a, 0, b, 0);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float pp = n.diag(w[i], a, 1, b, 1), pm = n.diag(w[i], a, 1, b, 0);
                float mp = n.diag(w[i], a, 0, b, 1), mm = n.diag(w[i], a, 0, b, 0);
                pp = epp ? pp : wc[i];
                pm = epm ? pm : wc[i];
                mp = emp ? mp : wc[i];
                mm = emm ? mm : wc[i];
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
                const double jic = (double)first[c][i], jci = (double)first[i][c];
                e += jic * jic + p.lambda64 * jic * jci;
            }
        energy = e;
    }
}

// a16: level-set term, level_set_term.py:28-64 (OOB -> 1; second derivatives use the +1 neighbour twice)
// lm/lp[a]: live at the -1/+1 neighbour along axis a with OOB already replaced by 1
template <int D, class NB>
__device__ inline void level_set_gradient(const NB& n, const typename NB::Field& live, const float (&lm)[3],
                                          const float (&lp)[3], float l, float (&gl)[3], double& energy) {
    float grad[3] = {0.0f, 0.0f, 0.0f};
    float hess[3][3];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        grad[c] = (0.5f * (lp[c] - lm[c])) * 10.0f;
        hess[c][c] = ((lp[c] - 2.0f * l) + lp[c]) * 10.0f;
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a + 1; b < D; ++b) {
            float pp = n.diag(live, a, 1, b, 1), mp = n.diag(live, a, 0, b, 1);
            float pm = n.diag(live, a, 1, b, 0), mm = n.diag(live, a, 0, b, 0);
            pp = n.diag_exists(a, 1, b, 1) ? pp : 1.0f;
            mp = n.diag_exists(a, 0, b, 1) ? mp : 1.0f;
            pm = n.diag_exists(a, 1, b, 0) ? pm : 1.0f;
            mm = n.diag_exists(a, 0, b, 0) ? mm : 1.0f;
            const float s = (a == 0 && b == 1) ? ((pp - mp) - pm) + mm   // level_set_term.py:52-53
                                               : ((pp - pm) - mp) + mm;  // pairs with z: z difference first
            const float h = (0.25f * s) * 10.0f;
            hess[a][b] = h;
            hess[b][a] = h;
        }
    float sq = grad[0] * grad[0];
#pragma unroll
    for (int c = 1; c < D; ++c) sq = sq + grad[c] * grad[c];
    const float nrm = sqrtf(sq);
    const float coef = (1.0f - nrm) / (nrm + 1e-5f);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        float hv = hess[i][0] * grad[0];
#pragma unroll
        for (int j = 1; j < D; ++j) hv = hv + hess[i][j] * grad[j];
        gl[i] = coef * hv;
    }
    const double dn = (double)nrm - 1.0;
    energy = 0.5 * dn * dn;
}

// gradient of the energy at one voxel of the narrow-band union (a12-a17); NB = Nbh<D> or NbhFast<D>
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, class NB>
__device__ inline void band_voxel_gradient(const float* __restrict__ live_base, const float* __restrict__ warp_prev,
                                           const Grid& g, const Params& p, int x, int y, int z, float l, float cn,
                                           float (&gv)[3], double (&en)[3]) {
    const bool live_truncated = fabsf(l) == 1.0f;
    const NB n(g, x, y, z);
    const typename NB::Field live = NB::field(live_base, g, 0);
    // ---- live neighbours (shared by np.gradient, the thresholded data term and the level-set term)
    float lmc[3], lpc[3];  // clamped: a missing neighbour reads the centre
#pragma unroll
    for (int a = 0; a < D; ++a) {
        lmc[a] = n.axis(live, a, 0);
        lpc[a] = n.axis(live, a, 1);
    }
    // ---- data term (data_term.py:169-187 / :334-349; thresholded variant :190-227)
    const float diff = l - cn;
    float lg[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < D; ++a) lg[a] = np_gradient_from(n, a, lmc[a], lpc[a]);
    if (DATA == LSF_DATA_THRESHOLDED_FDM) {
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const float fwd = (n.exists(a, 1) ? lpc[a] : 1.0f) - l;
            const float bwd = l - (n.exists(a, 0) ? lmc[a] : 1.0f);
            float alt = fabsf(fwd) < fabsf(bwd) ? fwd : bwd;
            alt = fabsf(alt) > 0.5f ? 0.0f : alt;
            lg[a] = fabsf(lg[a]) > 0.5f ? alt : lg[a];
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) gv[c] = p.w_data * ((diff * lg[c]) * 10.0f);
    if (ENERGY != LSF_ENERGY_NONE) en[0] = 0.5 * (double)diff * (double)diff;
    // ---- level-set term (DIRECT only; skipped where live is truncated, slavcheva_optimizer2d.py:274)
    if (LEVELSET && !live_truncated) {
        float lm1[3], lp1[3];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            lm1[a] = n.exists(a, 0) ? lmc[a] : 1.0f;
            lp1[a] = n.exists(a, 1) ? lpc[a] : 1.0f;
        }
        float gl[3];
        double e;
        level_set_gradient<D, NB>(n, live, lm1, lp1, l, gl, e);
#pragma unroll
        for (int c = 0; c < D; ++c) gv[c] = gv[c] + p.w_level_set * gl[c];
        if (ENERGY != LSF_ENERGY_NONE) en[2] = e;
    }
    // ---- smoothing term on the previous warp: axis neighbours with "missing -> centre" come free from clamping
    const typename NB::Field w[3] = {NB::field(warp_prev, g, 0), NB::field(warp_prev, g, 1),
                                     NB::field(warp_prev, g, D == 3 ? 2 : 0)};
    float wc[3] = {0.0f, 0.0f, 0.0f}, wm[3][3], wp[3][3];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        wc[c] = n.centre(w[c]);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            wm[a][c] = n.axis(w[c], a, 0);
            wp[a][c] = n.axis(w[c], a, 1);
        }
    }
    float gs[3] = {0.0f, 0.0f, 0.0f};
    if (SMOOTH == LSF_SMOOTHING_KILLING) {
        double e = 0.0;
        killing_gradient<D, NB>(n, w, wm, wp, wc, p, gs, e, ENERGY != LSF_ENERGY_NONE);
        if (ENERGY != LSF_ENERGY_NONE) en[1] = e;
    } else {
        tikhonov_gradient<D>(wm, wp, wc, gs);
        if (ENERGY == LSF_ENERGY_DIRECT) {
            // smoothing_term.py:134-139: 0.5 * sum_axis |0.5 (w[+1] - w[-1])|^2, OOB -> centre
            double e = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float der = 0.5f * (wp[a][c] - wm[a][c]);
                    e += (double)der * (double)der;
                }
            en[1] = 0.5 * e;
        } else if (ENERGY == LSF_ENERGY_VECTORIZED) {
            // smoothing_term.py:162-177: 0.5 * sum_{c,axis} np.gradient(warp_c)[axis]^2 over the band.
            // accumulation order (per component: x, y, z) as in oracle.smoothing_energy_vectorized is irrelevant
            // to the float64 sum at the 1e-9 level the tests ask for
            double e = 0.0;
#pragma unroll
            for (int c = 0; c < D; ++c)
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const float d = np_gradient_from(n, a, wm[a][c], wp[a][c]);
                    e += (double)d * (double)d;
                }
            en[1] = 0.5 * e;
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) gv[c] = gv[c] + p.w_smooth * gs[c];
}

// gradient of the energy at one voxel (a12-a17); zero outside the narrow-band union
// ALL_INTERIOR: the caller guarantees the neighbourhood of every voxel it visits (an INTERIOR band list): the generic
// path is not even compiled in, which keeps that kernel at 85 VGPRs / 5 waves per SIMD and ~620 VALU instructions
// per 64 voxels (the two-path kernel needs 111 VGPRs, the generic path ~850 instructions).
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, bool ALL_INTERIOR>
__device__ inline void voxel_gradient(const float* __restrict__ live, const float* __restrict__ canonical,
                                      const float* __restrict__ warp_prev, const Grid& g, const Params& p, int x,
                                      int y, int z, int i, float (&gv)[3], double (&en)[3]) {
    gv[0] = gv[1] = gv[2] = 0.0f;
    const float l = live[i], cn = canonical[i];
    if (fabsf(l) == 1.0f && fabsf(cn) == 1.0f) return;  // outside the narrow-band union (tsdf_set_routines.py:19-52)
    const bool interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && (D == 2 || (z > 0 && z < g.nz - 1));
    if (ALL_INTERIOR || (g.fast_ok && __all(interior)))  // wave-uniform: every band lane has its whole neighbourhood
        band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY, NbhFast<D>>(live, warp_prev, g, p, x, y, z, l, cn, gv, en);
    else
        band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY, Nbh<D>>(live, warp_prev, g, p, x, y, z, l, cn, gv, en);
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

// MODE: 0 = gradient only (Sobolev path), 1 = fused, 2 = fused over an all-interior band list
constexpr int kModeGradient = 0, kModeFused = 1, kModeFusedInterior = 2;

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, int MODE>
__global__ __launch_bounds__(kBlock) void slavcheva_iteration_kernel(
    const float* __restrict__ live, const float* __restrict__ canonical, const float* __restrict__ warp_prev,
    float* __restrict__ warp_out, float* __restrict__ live_out, float* __restrict__ g_out, Grid g, Params p,
    lsf_gate gate, lsf_iteration_record* record, const int* __restrict__ band_list, unsigned band_count) {
    constexpr bool FUSED = MODE != kModeGradient;
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    double en[3] = {0.0, 0.0, 0.0};
    auto voxel = [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float gv[3];
        double e[3] = {0.0, 0.0, 0.0};
        voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY, MODE == kModeFusedInterior>(live, canonical, warp_prev, g, p, x,
                                                                                      y, z, i, gv, e);
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
    };
    // Band list (lsf_band_list_fill): only voxels that can be in the narrow-band union are visited.  Every other voxel
    // has |live| == |canonical| == 1: zero gradient, zero warp, live' = live -- the caller initialised both ping-pong
    // buffer sets with exactly that, and such a voxel can never enter the band.  Their arg-max candidates are all
    // (length 0, own index); the smallest index of the launch's z-range stands for them.
    for_each_listed_voxel(g, FUSED ? band_list : nullptr, band_count, voxel);
    if (FUSED && band_list && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
        best = q > best ? q : best;
    }
    if (FUSED || ENERGY != LSF_ENERGY_NONE) {
        double* dst[3] = {ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->data_energy : nullptr,
                          ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->smoothing_energy : nullptr,
                          ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->level_set_energy : nullptr};
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
        const int i = vidx(g, x, y, z);
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
    const int* band_list;
    unsigned band_count;
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, int MODE>
void launch_one(const LaunchArgs& a) {
    hipLaunchKernelGGL((slavcheva_iteration_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY, MODE>), dim3(a.blocks),
                       dim3(kTileX * a.g.tile_y), 0, a.s, a.live, a.canonical, a.warp_prev, a.warp_out, a.live_out, a.g_out, a.g,
                       a.p, a.gate, a.record, a.band_list, a.band_count);
}

template <int D, int SMOOTH, bool LEVELSET, int DATA, int MODE>
void pick_energy(int energy, const LaunchArgs& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT, MODE>(a); break;
        case LSF_ENERGY_VECTORIZED: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED, MODE>(a); break;
        default: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE, MODE>(a); break;
    }
}

template <int D, int MODE>
void pick_terms(const lsf_slavcheva_params* q, const LaunchArgs& a) {
    const bool killing = q->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = q->level_set_enabled != 0;
    const bool fdm = q->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = q->energy_mode;
#define LSF_PICK(S, L, DM) pick_energy<D, S, L, DM, MODE>(e, a)
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

// exclusive prefix sums of sums[0..n) in place by ONE block (n <= 2^21 chunks; runs once per optimize() call)
__global__ __launch_bounds__(1024) void band_scan_kernel(int* __restrict__ sums, unsigned n, long long* total) {
    __shared__ int buf[1024];
    const unsigned t = threadIdx.x;
    int carry = 0;
    for (unsigned base = 0; base < n; base += 1024) {
        const unsigned i = base + t;
        const int v = i < n ? sums[i] : 0;
        buf[t] = v;
        __syncthreads();
        for (unsigned off = 1; off < 1024; off <<= 1) {
            const int add = t >= off ? buf[t - off] : 0;
            __syncthreads();
            buf[t] += add;
            __syncthreads();
        }
        if (i < n) sums[i] = carry + buf[t] - v;
        carry += buf[1023];
        __syncthreads();
    }
    if (t == 0) *total = carry;
}

}  // namespace

static inline bool band_subset_ok(int32_t subset) {
    return subset == LSF_BAND_ALL || subset == LSF_BAND_INTERIOR || subset == LSF_BAND_BOUNDARY;
}

extern "C" int lsf_slavcheva_iteration(int32_t stage, const float* live, const float* canonical,
                                       const float* warp_prev_planar, float* warp_out_planar, float* live_out,
                                       float* g_out_planar, const lsf_grid* grid, const lsf_slavcheva_params* params,
                                       const lsf_gate* gate, lsf_iteration_record* record, const int32_t* band_list,
                                       int64_t band_count, int32_t band_subset, void* stream) {
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
    // work unit = (64 x 4) tile, 4-wave blocks.  Finer units (1- or 2-wave blocks on 64 x 1 / 64 x 2 tiles, which
    // would average the ~4x cost difference between band and non-band units over more units per block) were measured
    // and lose: the four rows of a tile share stencil rows through L1 (256^3 all-in-band 0.39 / 0.46 / 0.59 ms for
    // tile heights 4 / 2 / 1).
    const int tile_y = 4;
    const unsigned per_xcd = blocks_per_xcd();
    Grid g = make_grid(grid, tile_y);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    const bool listed = band_list != nullptr;
    if (listed && (stage != LSF_STAGE_FUSED || g_out_planar || band_count < 0 || band_count > 0x7fffffffll ||
                   !band_subset_ok(band_subset)))
        return LSF_ERR_BAD_ARGUMENT;
    // an INTERIOR list runs the kernel without the generic neighbourhood path; it needs 32-bit buffer offsets
    const bool all_interior = listed && band_subset == LSF_BAND_INTERIOR;
    if (all_interior && !g.fast_ok) return LSF_ERR_BAD_ARGUMENT;
    const unsigned blocks = listed ? band_list_blocks((unsigned)band_count) : launch_blocks(t.total, per_xcd);
    LaunchArgs a{blocks, as_stream(stream), live, canonical, warp_prev_planar, warp_out_planar,
                 live_out, g_out_planar, g, make_params(params), gate_or_open(gate), record,
                 band_list, (unsigned)band_count};
    if (grid->dims == 2) {
        if (all_interior) pick_terms<2, kModeFusedInterior>(params, a);
        else if (stage == LSF_STAGE_FUSED) pick_terms<2, kModeFused>(params, a);
        else pick_terms<2, kModeGradient>(params, a);
    } else {
        if (all_interior) pick_terms<3, kModeFusedInterior>(params, a);
        else if (stage == LSF_STAGE_FUSED) pick_terms<3, kModeFused>(params, a);
        else pick_terms<3, kModeGradient>(params, a);
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
    hipLaunchKernelGGL(band_scan_kernel, dim3(1), dim3(1024), 0, s, scratch, chunks, (long long*)count_out);
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
