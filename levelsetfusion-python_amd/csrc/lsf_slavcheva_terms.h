// Per-voxel energy-gradient terms of the Slavcheva-style optimizer (a12-a17), shared by the planar kernels
// (lsf_slavcheva.hip: gradient stage of the Sobolev path) and the state-layout fused kernel
// (lsf_slavcheva_state.hip).  Reference: nonrigid_opt/slavcheva/{data_term,smoothing_term,level_set_term}.py.
#pragma once
#include "lsf_device.h"

namespace lsf {
namespace slav {

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
    const float *live_base, *warp_base;
    long long plane;

    __device__ inline Nbh(const float* live_field, const float* warp_planar, const Grid& g, int x, int y, int z)
        : live_base(live_field), warp_base(warp_planar), plane(g.plane) {
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
    __device__ inline Field live() const { return live_base; }
    __device__ inline Field warp(int c) const { return warp_base + c * plane; }
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
    __amdgpu_buffer_rsrc_t live_rsrc, warp_rsrc;
    unsigned plane_bytes;

    __device__ inline NbhFast(const float* live_field, const float* warp_planar, const Grid& g, int x, int y, int z) {
        // one resource spans all planes of the vector field (D * plane * 4 bytes < 4 GiB, see Grid::fast_ok)
        live_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(live_field), 0, -1, 0x00020000);
        warp_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(warp_planar), 0, -1, 0x00020000);
        plane_bytes = (unsigned)g.plane * 4u;
        sy = (unsigned)g.nx * 4u;
        sz = (unsigned)(g.nx * g.ny) * 4u;
        const unsigned corner = (unsigned)vidx(g, x, y, z) * 4u - 4u - sy - (D == 3 ? sz : 0u);
        wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)corner);
        lane_delta = corner - wave_base;
    }
    __device__ inline Field live() const { return BufField{live_rsrc, 0u}; }
    __device__ inline Field warp(int c) const { return BufField{warp_rsrc, (unsigned)c * plane_bytes}; }
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
        // sum_ic J_ic^2 + lambda * sum_ic J_ic J_ci  =  |J|_F^2 + lambda * (sum_i J_ii^2 + 2 sum_{i<c} J_ic J_ci),
        // J_ic = d w_i / d c.  The local contribution is formed in float32 like the reference's
        // (smoothing_term.py:93-98: float32 dot products of the float32 Jacobian) and widened for the sum only: the
        // float64 form of these products was 16 % of the fused kernel's time.  Order: oracle.killing_gradient.
        float sq[3][3];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int c = 0; c < D; ++c) sq[i][c] = first[c][i] * first[c][i];
        float frob = sq[0][0], diag = sq[0][0], off = 0.0f;
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                if (i + c > 0) frob = frob + sq[i][c];
                if (c == i && i > 0) diag = diag + sq[i][i];
                if (c > i) {
                    const float t = first[c][i] * first[i][c];
                    off = (i == 0 && c == 1) ? t : off + t;
                }
            }
        energy = (double)(frob + p.lambda32 * (diag + (off + off)));
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
    const float dn = nrm - 1.0f;  // level_set_term.py:63, float32 like the gradient length it is formed from
    energy = (double)(0.5f * (dn * dn));
}

// gradient of the energy at one voxel of the narrow-band union (a12-a17); NB = Nbh<D> or NbhFast<D>
template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, class NB>
__device__ inline void band_voxel_gradient(const NB& n, const Params& p, float l, float cn, float (&gv)[3],
                                           double (&en)[3]) {
    const bool live_truncated = fabsf(l) == 1.0f;
    const typename NB::Field live = n.live();
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
    if (ENERGY != LSF_ENERGY_NONE) en[0] = (double)(0.5f * (diff * diff));  // data_term.py:185, float32
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
    const typename NB::Field w[3] = {n.warp(0), n.warp(1), n.warp(D == 3 ? 2 : 0)};
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
            float e = 0.0f;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float der = 0.5f * (wp[a][c] - wm[a][c]);
                    e = (a + c == 0) ? der * der : e + der * der;
                }
            en[1] = (double)(0.5f * e);
        } else if (ENERGY == LSF_ENERGY_VECTORIZED) {
            // smoothing_term.py:162-177: 0.5 * sum_{c,axis} np.gradient(warp_c)[axis]^2 over the band.
            // float32 per voxel in oracle.smoothing_energy_vectorized's order (per component: x, y, z)
            float e = 0.0f;
#pragma unroll
            for (int c = 0; c < D; ++c)
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const float d = np_gradient_from(n, a, wm[a][c], wp[a][c]);
                    e = (a + c == 0) ? d * d : e + d * d;
                }
            en[1] = (double)(0.5f * e);
        }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) gv[c] = gv[c] + p.w_smooth * gs[c];
}

}  // namespace slav
}  // namespace lsf
