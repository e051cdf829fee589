// The per-voxel pieces of the fused warp-update kernels on the STATE layout (float4 [z][y][x] = (live, u, v, w)), shared
// by the per-iteration kernel (lsf_slavcheva_state.hip) and the multi-iteration chain kernel (lsf_slavcheva_chain.hip):
// the 3^D neighbourhood in registers, the gradient terms written on whole taps, the re-warp gather out of the
// neighbourhood.  Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330 with data_term.py,
// smoothing_term.py, level_set_term.py and field_warping.warp_field_advanced (:112-151).
#pragma once
#include "lsf_slavcheva_terms.h"

namespace lsf {
namespace slav {

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef unsigned vu4 __attribute__((ext_vector_type(4)));

// a - b on two / four floats at once.  The compiler selects v_pk_mul_f32 / v_pk_add_f32 for float2 products and sums but
// splits a float2 DIFFERENCE into two v_sub_f32; the packed add takes per-source negation modifiers, and a + (-b) is
// a - b bit for bit.
__device__ inline vf2 pk_sub(vf2 a, vf2 b) {
    vf2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ inline vf4 vsub(const vf4& a, const vf4& b) {
    const vf2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

__device__ inline float comp(const vf4& v, int f) { return f == 0 ? v.x : (f == 1 ? v.y : (f == 2 ? v.z : v.w)); }

// 3^D neighbourhood of one voxel held in registers: tap (dx, dy, dz) with at most two non-zero offsets (the 19-point
// stencil of the Killing cross derivatives; 9 points in 2-D).  Field = component of the state.
template <int D>
struct TapsBase {
    using Field = int;
    vf4 t[3][3][3];  // [dz + 1][dy + 1][dx + 1]; only the taps the stencils use are ever loaded
    __device__ inline Field live() const { return 0; }
    __device__ inline Field warp(int c) const { return 1 + c; }
    __device__ inline float centre(Field f) const { return comp(t[1][1][1], f); }
    __device__ inline float axis(Field f, int a, int s) const {
        const int d = s ? 1 : -1;
        return comp(t[1 + (a == 2 ? d : 0)][1 + (a == 1 ? d : 0)][1 + (a == 0 ? d : 0)], f);
    }
    __device__ inline float diag(Field f, int a, int sa, int b, int sb) const {
        const int da = sa ? 1 : -1, db = sb ? 1 : -1;
        return comp(t[1 + (a == 2 ? da : 0) + (b == 2 ? db : 0)][1 + (a == 1 ? da : 0) + (b == 1 ? db : 0)]
                     [1 + (a == 0 ? da : 0) + (b == 0 ? db : 0)], f);
    }
};

// generic: every neighbour read from a CLAMPED offset, the reference's OOB rules applied by the terms with selects
template <int D>
struct NbhState : TapsBase<D> {
    bool has[3][2];
    __device__ inline NbhState(const vf4* __restrict__ s, const Grid& g, int x, int y, int z, const vf4& centre_value) {
        const int i = vidx(g, x, y, z);
        const int stride[3] = {1, g.nx, g.nx * g.ny};
        const int coord[3] = {x, y, z};
        const int extent[3] = {g.nx, g.ny, g.nz};
        int off[3][3];  // [axis][d + 1]
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            has[a][0] = a < D && coord[a] > 0;
            has[a][1] = a < D && coord[a] < extent[a] - 1;
            off[a][0] = has[a][0] ? -stride[a] : 0;
            off[a][1] = 0;
            off[a][2] = has[a][1] ? stride[a] : 0;
        }
#pragma unroll
        for (int dz = (D == 3 ? -1 : 0); dz <= (D == 3 ? 1 : 0); ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nz = (dx != 0) + (dy != 0) + (dz != 0);
                    if (nz > 2) continue;
                    this->t[dz + 1][dy + 1][dx + 1] =
                        nz == 0 ? centre_value : s[i + off[0][dx + 1] + off[1][dy + 1] + off[2][dz + 1]];
                }
    }
    __device__ inline bool exists(int a, int s) const { return has[a][s]; }
    __device__ inline bool diag_exists(int a, int sa, int b, int sb) const { return has[a][sa] && has[b][sb]; }
};

// every lane's whole neighbourhood lies inside the array (wave vote or an INTERIOR band list): one per-lane byte offset
// (the neighbourhood's lowest corner relative to the wave's first lane), the tap chosen by the instruction's scalar
// offset (dy, dz) and immediate (dx) -- no per-tap VALU address arithmetic, no OOB selects.  32-bit buffer offsets:
// while 16 * nz * ny * nx < 2^32 (Grid::fast_ok) one resource spans the state and the first lane's corner is part of
// the scalar offset; `wide` (arrays of 4 GiB and more): the resource STARTS at the first lane's corner (a 64-bit
// scalar), so that the offsets only have to span one wave's voxels plus a neighbourhood -- the caller checks that per
// wave (wave_span_ok).  The INTERIOR kernel passes a literal false and keeps the first form only.
// (Measured and rejected: taking the ten x -/+ 1 taps from the neighbouring lane's own-x taps by wave-wide DPP shifts,
// with exec-masked loads only for the lanes at the ends of an x-run -- bit-identical, 27 % fewer bytes through the
// vector L1, yet 4 % SLOWER on both walks: a mostly masked buffer_load_dwordx4 occupies the address path like a full
// one, and almost every wave has some lane at a run end.  DESIGN.md section 5.)
template <int D>
struct NbhStateFast : TapsBase<D> {
    __device__ inline NbhStateFast() {}
    __device__ inline void load(const vf4* __restrict__ s, const Grid& g, unsigned i, const vf4& centre_value,
                                bool wide) {
        const unsigned sy = (unsigned)g.nx * 16u, sz = (unsigned)(g.nx * g.ny) * 16u;
        // per lane only the low 32 bits matter: the difference to the first lane's corner is < 2^32 (see above)
        const unsigned corner = (unsigned)i * 16u - 16u - sy - (D == 3 ? sz : 0u);
        const int first = __builtin_amdgcn_readfirstlane((int)i);  // smallest: ascending by lane
        const long long base = (long long)first * 16 - 16 - (long long)sy - (D == 3 ? (long long)sz : 0ll);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(s)) + (wide ? base : 0ll), 0, -1, 0x00020000);
        const unsigned wave_base = wide ? 0u : (unsigned)base;
        const unsigned lane_delta = corner - (unsigned)base;
#pragma unroll
        for (int dz = (D == 3 ? -1 : 0); dz <= (D == 3 ? 1 : 0); ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int nz = (dx != 0) + (dy != 0) + (dz != 0);
                    if (nz > 2) continue;
                    if (nz == 0) {
                        this->t[1][1][1] = centre_value;
                        continue;
                    }
                    const unsigned soff = wave_base + (unsigned)(dy + 1) * sy + (D == 3 ? (unsigned)(dz + 1) * sz : 0u);
                    this->t[dz + 1][dy + 1][dx + 1] = __builtin_bit_cast(
                        vf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_delta + (unsigned)(dx + 1) * 16u),
                                                                   (int)soff, 0));
                }
    }
    __device__ static constexpr bool exists(int, int) { return true; }
    __device__ static constexpr bool diag_exists(int, int, int, int) { return true; }
};

// band_voxel_gradient (lsf_slavcheva_terms.h) for a 3-D voxel whose whole neighbourhood is in registers and inside
// the array, written on whole taps: the finite differences of the level-set term (live) and of the Killing / Tikhonov
// terms (u, v, w) are the SAME stencils applied to the four channels of the state, so one float4 expression --
// two packed-float instructions, v_pk_add_f32 / v_pk_mul_f32 on the (live, u) and (v, w) register pairs a 16-byte
// load leaves behind -- replaces four scalar ones.  The fused kernel is bound by VALU issue (tools/state_trace.py:
// ~425 VALU instructions per 64 voxels at 4 waves per SIMD = 81 % of the issue slots), not by memory.
// Element by element these are the operations of the scalar terms in the same order (no contraction, no
// reassociation: -ffp-contract=off, IEEE vector semantics), so results are bit-identical; only where the reference
// treats live and warp differently (second differences along x and z, the x-y cross term's association) is the live
// channel computed on its own.
template <int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__device__ inline void band_voxel_gradient_taps(const TapsBase<3>& n, const Params& p, float l, float cn,
                                                float (&gv)[3], double (&en)[3]) {
    const vf4 c = n.t[1][1][1];
    const vf4 xm = n.t[1][1][0], xp = n.t[1][1][2], ym = n.t[1][0][1], yp = n.t[1][2][1];
    const vf4 zm = n.t[0][1][1], zp = n.t[2][1][1];
    // central differences of all four channels: .x = np.gradient(live) (interior), .yzw = J[., axis]
    const vf4 d[3] = {0.5f * vsub(xp, xm), 0.5f * vsub(yp, ym), 0.5f * vsub(zp, zm)};
    // ---- data term (data_term.py:169-187 / :334-349; thresholded variant :190-227)
    const float diff = l - cn;
    float lg[3] = {d[0].x, d[1].x, d[2].x};
    if (DATA == LSF_DATA_THRESHOLDED_FDM) {
        const float lm[3] = {xm.x, ym.x, zm.x}, lp[3] = {xp.x, yp.x, zp.x};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float fwd = lp[a] - l, bwd = l - lm[a];
            float alt = fabsf(fwd) < fabsf(bwd) ? fwd : bwd;
            alt = fabsf(alt) > 0.5f ? 0.0f : alt;
            lg[a] = fabsf(lg[a]) > 0.5f ? alt : lg[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) gv[a] = p.w_data * ((diff * lg[a]) * 10.0f);
    if (ENERGY != LSF_ENERGY_NONE) en[0] = (double)(0.5f * (diff * diff));
    const bool killing = SMOOTH == LSF_SMOOTHING_KILLING;
    const bool want_ls = LEVELSET && !(fabsf(l) == 1.0f);
    // ---- second differences: p - 2 c for every channel, then + m (warp along x, z) or + p (warp along y: the
    // reference's typo, smoothing_term.py:69; live along every axis, level_set_term.py:47-48)
    const vf4 c2 = 2.0f * c;
    const vf4 qx = vsub(xp, c2), qy = vsub(yp, c2), qz = vsub(zp, c2);
    const vf4 sx = qx + xm, sy = qy + yp, sz = qz + zm;  // .yzw: Killing's second differences; sy.x: live's along y
    // ---- cross differences of the three axis pairs; pairs with z: the same association for live and warp
    const vf4 pxz = vsub(vsub(n.t[2][1][2], n.t[0][1][2]), n.t[2][1][0]) + n.t[0][1][0];  // ((pp - pm) - mp) + mm, a = x, b = z
    const vf4 pyz = vsub(vsub(n.t[2][2][1], n.t[0][2][1]), n.t[2][0][1]) + n.t[0][0][1];  // a = y, b = z
    const vf4 kxz = pxz * 0.25f, kyz = pyz * 0.25f;
    const vf4 ppxy = n.t[1][2][2], pmxy = n.t[1][0][2], mpxy = n.t[1][2][0], mmxy = n.t[1][0][0];
    if (want_ls) {
        // level_set_term.py:28-64
        const float grad[3] = {d[0].x * 10.0f, d[1].x * 10.0f, d[2].x * 10.0f};
        float hess[3][3];
        hess[0][0] = (qx.x + xp.x) * 10.0f;
        hess[1][1] = sy.x * 10.0f;
        hess[2][2] = (qz.x + zp.x) * 10.0f;
        const float sxy = ((ppxy.x - mpxy.x) - pmxy.x) + mmxy.x;  // level_set_term.py:52-53
        hess[0][1] = hess[1][0] = (0.25f * sxy) * 10.0f;
        hess[0][2] = hess[2][0] = kxz.x * 10.0f;
        hess[1][2] = hess[2][1] = kyz.x * 10.0f;
        float sq = grad[0] * grad[0];
        sq = sq + grad[1] * grad[1];
        sq = sq + grad[2] * grad[2];
        const float nrm = sqrtf(sq);
        const float coef = (1.0f - nrm) / (nrm + 1e-5f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float hv = hess[i][0] * grad[0];
            hv = hv + hess[i][1] * grad[1];
            hv = hv + hess[i][2] * grad[2];
            gv[i] = gv[i] + p.w_level_set * (coef * hv);
        }
        if (ENERGY != LSF_ENERGY_NONE) {
            const float dn = nrm - 1.0f;
            en[2] = (double)(0.5f * (dn * dn));
        }
    }
    // ---- smoothing term on the previous warp
    float gs[3];
    if (killing) {
        // smoothing_term.py:50-100 with every quirk (lsf_slavcheva_terms.h::killing_gradient); DESIGN.md section 3
        const vf4 kxy = (vsub(vsub(ppxy, pmxy), mpxy) + mmxy) * 0.25f;
        const vf4 g0 = p.killing_c1 * sx;
        const vf4 g1 = (g0 + sy) + sz;  // c1 * w_xx + w_yy + w_zz per channel
        gs[0] = (g1.y + p.lambda32 * kxy.z) + p.lambda32 * kxz.w;  // u: + lambda v_xy + lambda w_xz
        gs[1] = (g1.z + p.lambda32 * kxy.y) + p.lambda32 * kyz.w;  // v: + lambda u_xy + lambda w_yz
        gs[2] = (g1.w + p.lambda32 * kxz.y) + p.lambda32 * kyz.z;  // w: + lambda u_xz + lambda v_yz
        if (ENERGY != LSF_ENERGY_NONE) {
            // |J|_F^2 + lambda (sum_i J_ii^2 + 2 sum_{i<c} J_ic J_ci), J_ic = d[c][1 + i]; float32, the order of
            // killing_gradient in lsf_slavcheva_terms.h
            const vf4 q0 = d[0] * d[0], q1 = d[1] * d[1], q2 = d[2] * d[2];  // .yzw: J_0c^2, J_1c^2, J_2c^2 for c = x, y, z
            float frob = q0.y;           // i = 0: c = 0, 1, 2
            frob = frob + q1.y;
            frob = frob + q2.y;
            frob = frob + q0.z;          // i = 1
            frob = frob + q1.z;
            frob = frob + q2.z;
            frob = frob + q0.w;          // i = 2
            frob = frob + q1.w;
            frob = frob + q2.w;
            float diag = q0.y;
            diag = diag + q1.z;
            diag = diag + q2.w;
            float off = d[1].y * d[0].z;      // J_01 J_10
            off = off + d[2].y * d[0].w;      // J_02 J_20
            off = off + d[2].z * d[1].w;      // J_12 J_21
            en[1] = (double)(frob + p.lambda32 * (diag + (off + off)));
        }
    } else {
        // -Laplacian, scipy rounding (lsf_slavcheva_terms.h::tikhonov_gradient)
        const float wc[3] = {c.y, c.z, c.w};
        const float wm[3][3] = {{xm.y, xm.z, xm.w}, {ym.y, ym.z, ym.w}, {zm.y, zm.z, zm.w}};
        const float wp[3][3] = {{xp.y, xp.z, xp.w}, {yp.y, yp.z, yp.w}, {zp.y, zp.z, zp.w}};
        tikhonov_gradient<3>(wm, wp, wc, gs);
        if (ENERGY == LSF_ENERGY_DIRECT || ENERGY == LSF_ENERGY_VECTORIZED) {
            // smoothing_term.py:134-139 / :162-177: interior voxels, both forms square the same central differences
            float e = 0.0f;
            if (ENERGY == LSF_ENERGY_DIRECT) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float der = i == 0 ? d[a].y : (i == 1 ? d[a].z : d[a].w);
                        e = (a + i == 0) ? der * der : e + der * der;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float der = i == 0 ? d[a].y : (i == 1 ? d[a].z : d[a].w);
                        e = (a + i == 0) ? der * der : e + der * der;
                    }
            }
            en[1] = (double)(0.5f * e);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) gv[i] = gv[i] + p.w_smooth * gs[i];
}

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__device__ inline void fast_voxel_gradient(const NbhStateFast<D>& n, const Params& p, float l, float cn,
                                           float (&gv)[3], double (&en)[3]) {
    if constexpr (D == 3) band_voxel_gradient_taps<SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, en);
    else band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, en);
}

// NbhStateFast's requirement on arrays of 4 GiB and more: every active lane's neighbourhood within 2^32 bytes of the
// first lane's (voxel indices ascend with the lane in every walk; a list's consecutive entries can still lie far apart)
__device__ inline bool wave_span_ok(const Grid& g, int i) {
    if (g.fast_ok) return true;
    const int first = __builtin_amdgcn_readfirstlane(i);
    const long long reach = 3ll * 16 + 2ll * 16 * g.nx + 2ll * 16 * g.nx * g.ny;
    return __all(((long long)i - first) * 16 + reach < 0xffffffffll);
}

// the re-warp's D-linear gather of the live component (OOB -> 1): lerp z, then y, then x as sample_linear does
template <int D>
__device__ inline float state_gather(const vf4* __restrict__ s, const Grid& g, float px, float py, float pz) {
    const AxisTaps ax = axis_taps(px, g.nx, 0), ay = axis_taps(py, g.ny, g.y_global_offset);  // py, pz: GLOBAL positions
    AxisTaps az = ax;
    if (D == 3) az = axis_taps(pz, g.nz, g.z_global_offset);
    const bool cell_inside = ax.v0 && ax.v1 && ay.v0 && ay.v1 && (D == 2 || (az.v0 && az.v1));
    if (!(g.fast_ok && __all(cell_inside))) {
        const float* f = reinterpret_cast<const float*>(s);
        auto rd = [&](const AxisTaps& tx, int ox, const AxisTaps& ty, int oy, const AxisTaps& tz, int oz) {
            const long long idx = ((long long)(D == 3 ? (oz ? tz.c1 : tz.c0) : 0) * g.ny + (oy ? ty.c1 : ty.c0)) * g.nx +
                                  (ox ? tx.c1 : tx.c0);
            const bool valid = (ox ? tx.v1 : tx.v0) && (oy ? ty.v1 : ty.v0) && (D == 2 || (oz ? tz.v1 : tz.v0));
            const float v = f[idx * 4];
            return valid ? v : 1.0f;
        };
        if (D == 2) {
            const float i0 = rd(ax, 0, ay, 0, az, 0) * ay.i + rd(ax, 0, ay, 1, az, 0) * ay.r;
            const float i1 = rd(ax, 1, ay, 0, az, 0) * ay.i + rd(ax, 1, ay, 1, az, 0) * ay.r;
            return i0 * ax.i + i1 * ax.r;
        }
        float c[2][2];
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy) c[ox][oy] = rd(ax, ox, ay, oy, az, 0) * az.i + rd(ax, ox, ay, oy, az, 1) * az.r;
        const float i0 = c[0][0] * ay.i + c[0][1] * ay.r;
        const float i1 = c[1][0] * ay.i + c[1][1] * ay.r;
        return i0 * ax.i + i1 * ax.r;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<vf4*>(s), 0, -1, 0x00020000);
    const unsigned sy = (unsigned)g.nx * 16u, sz = (unsigned)(g.nx * g.ny) * 16u;
    const unsigned corner = (unsigned)(((D == 3 ? az.c0 : 0) * g.ny + ay.c0) * g.nx + ax.c0) * 16u;
    auto tap = [&](int dx, int dy, int dz) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             rsrc, (int)(corner + (unsigned)dx * 16u),
                                             (int)((unsigned)dy * sy + (unsigned)dz * sz), 0));
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

// The re-warp gather out of the neighbourhood that is already in registers (3-D, wave-uniform): while a warp update
// stays inside (-1, 1) per axis its 2^3-voxel cell is {voxel, neighbour at sx} x {voxel, neighbour at sy} x {voxel,
// neighbour at sz} with s = -1 / +1 by the side the position falls on, i.e. it lies inside the voxel's own 3^3
// neighbourhood: seven of the eight taps are among the 19 loaded ones (at most two non-zero offsets) and only the far
// corner (sx, sy, sz) has to be fetched -- one dword load instead of eight 16-byte-strided ones.
// Written per axis in terms of NEAR (the voxel's own coordinate) and FAR (the neighbour): sample_linear's lerp is
// lower * (1 - r) + upper * r; with the cell below the voxel the near tap is the upper one, otherwise the lower one,
// so lower * i + upper * r == near * wn + far * wf with (wn, wf) = below ? (r, i) : (i, r) -- the same two products,
// and a float add commutes: bit-identical to state_gather, lerp order z, y, x.  Twelve selects pick the seven taps
// (1 + 1 + 1 for the axis neighbours, 3 for each of the three diagonals) instead of the 38 of a 27 -> 8 cell select.
// The far corner enters last on every level, so everything but three multiply-adds is computed BEFORE its load
// returns:  value = R + (Q + (P + corner * wfz) * wfy) * wfx.
struct Rewarp {
    float P, Q, R, wfz, wfy, wfx, corner;
    bool lerp;  // false: R already is the value
    __device__ inline float value() const {
        if (!lerp) return R;
        const float c11 = P + corner * wfz;
        const float iy1 = Q + c11 * wfy;
        return R + iy1 * wfx;
    }
};

// One axis of the cell.  With p = fl(coordinate + displacement) and d = p - coordinate (exact for coordinates >= 2:
// p lies within a factor 2 of the coordinate, Sterbenz), sample_linear's floor / ratio / 1 - ratio reduce to
//   d >= 0: floor = coordinate,     ratio = d (exact),      1 - ratio = fl(1 - d)
//   d <  0: floor = coordinate - 1, ratio = 1 + d (exact),  1 - ratio = -d (exact, so no rounding happens)
// i.e. far weight = |d| and near weight = fl(1 - |d|) on either side: three instructions per axis instead of a floor,
// two subtractions and two selects.  (At coordinate 1 a negative d is not exact; the caller votes those waves out.)
// near: the cell lies inside the voxel's own neighbourhood, floor in {coordinate - 1, coordinate}.
struct NearFar {
    bool below, near;
    float wn, wf;
    __device__ inline NearFar(float coordinate, float displacement) {
        const float p = coordinate + displacement;
        const float d = p - coordinate;
        below = d < 0.0f;
        near = d >= -1.0f && d < 1.0f;
        wf = fabsf(d);
        wn = 1.0f - wf;
    }
};

// Requires every active lane's 3^3 neighbourhood inside the array (the callers' `interior` vote or an INTERIOR list).
// Returns false (for the whole wave) when some lane's cell leaves the neighbourhood or a lane stands at coordinate 1;
// FAST32: 32-bit buffer offsets span the state (Grid::fast_ok).
template <int D, bool FAST32>
__device__ inline bool rewarp_from_taps(const TapsBase<D>& n, const vf4* __restrict__ s, const Grid& g, int i, int x,
                                        int y, int z, const float (&wv)[3], Rewarp& rw) {
    if (D != 3) return false;
    const NearFar ax((float)x, wv[0]), ay((float)(y + g.y_global_offset), wv[1]), az((float)(z + g.z_global_offset), wv[2]);
    const unsigned lowest = min(min((unsigned)x, (unsigned)y), (unsigned)z);
    if (!__all(ax.near && ay.near && az.near && lowest >= 2u)) return false;
    const int slice = g.nx * g.ny;
    const int ci = i + (ax.below ? -1 : 1) + (ay.below ? -g.nx : g.nx) + (az.below ? -slice : slice);
    if (FAST32) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<vf4*>(s), 0, -1, 0x00020000);
        rw.corner = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)((unsigned)ci * 16u), 0, 0));
    } else {
        rw.corner = reinterpret_cast<const float*>(s)[(long long)ci * 4];
    }
    auto L = [&](int dz, int dy, int dx) { return n.t[dz + 1][dy + 1][dx + 1].x; };
    auto pick2 = [&](bool b, float m, float p) { return b ? m : p; };
    const float t000 = L(0, 0, 0);
    const float t100 = pick2(az.below, L(-1, 0, 0), L(1, 0, 0));   // far z
    const float t010 = pick2(ay.below, L(0, -1, 0), L(0, 1, 0));   // far y
    const float t001 = pick2(ax.below, L(0, 0, -1), L(0, 0, 1));   // far x
    const float t110 = pick2(az.below, pick2(ay.below, L(-1, -1, 0), L(-1, 1, 0)), pick2(ay.below, L(1, -1, 0), L(1, 1, 0)));
    const float t101 = pick2(az.below, pick2(ax.below, L(-1, 0, -1), L(-1, 0, 1)), pick2(ax.below, L(1, 0, -1), L(1, 0, 1)));
    const float t011 = pick2(ay.below, pick2(ax.below, L(0, -1, -1), L(0, -1, 1)), pick2(ax.below, L(0, 1, -1), L(0, 1, 1)));
    // z lerp of the four (x, y) columns, y lerp of the two x columns, x lerp; [far y][far x]
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

// a voxel whose update is computed but whose re-warped value may still wait for the far corner's load
struct Deferred {
    Rewarp rw;
    float wv[3];
    int i;  // voxel index; < 0: nothing to finish
};

}  // namespace slav
}  // namespace lsf
