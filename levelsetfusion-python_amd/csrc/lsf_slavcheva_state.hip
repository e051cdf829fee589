// The fused per-voxel warp-update kernel of the Slavcheva-style (KillingFusion) optimizer on the STATE layout,
// D = 2, 3.  Reference: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330 with
// data_term.py, smoothing_term.py, level_set_term.py and field_warping.warp_field_advanced (:112-151).
//
// State layout: float4 [z][y][x] = (live, u, v, w) -- the two fields an iteration reads through the SAME 3^D
// neighbourhood and ping-pongs together.  One 16-byte load per neighbour instead of four dword loads from four planes:
// a wave of band-list voxels (2-3 short x-runs) touches partially used 64-byte sectors at every run end, and the
// vector L1 coalesces per 16 lanes -- 16 lanes x 16 B fill four sectors where 16 lanes x 4 B fill a quarter of one to
// two.  Measured on the 256^3 sphere pair: ~13 L1 accesses per dword wave-load on the planar layout (76 loads per
// voxel), DESIGN.md section 7.  One iteration reads state (16 B) + canonical (4 B) and writes state' (16 B):
// 36 B / voxel-update of compulsory HBM traffic in 3-D (SURVEY 8d books 52 B for the two-pass formulation).
// Arithmetic and operation order are those of the planar kernels (lsf_slavcheva_terms.h): results are bit-identical.
#include "lsf_slavcheva_terms.h"

using namespace lsf;
using namespace lsf::slav;

namespace {

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
    const AxisTaps ax = axis_taps(px, g.nx, 0), ay = axis_taps(py, g.ny, 0);
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
    const NearFar ax((float)x, wv[0]), ay((float)y, wv[1]), az((float)(z + g.z_global_offset), wv[2]);
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

#ifdef LSF_STATE_TRACE  // measurement builds only (tools/state_trace.py): shader-clock stamps of the INTERIOR list walk
__device__ unsigned long long* g_state_trace = nullptr;  // [block < kTraceBlocks][wave 16][unit kTraceUnits][stamp 8]
__device__ unsigned long long* g_state_waves = nullptr;  // [block][wave 16][8]: 100 MHz stamps entry / loop / exit, clocks
constexpr unsigned kTraceBlocks = 64, kTraceUnits = 16, kTraceStamps = 8;
#define LSF_TRACE(slot)                                                                                               \
    do {                                                                                                              \
        if (trace_row && (threadIdx.x & 63) == 0) trace_row[slot] = __builtin_readcyclecounter();                     \
    } while (0)
#else
#define LSF_TRACE(slot) do {} while (0)
#endif

// WALK: the dense tile walk over every voxel, a band list (ALL or BOUNDARY subset), an INTERIOR band list
constexpr int kWalkDense = 0, kWalkList = 1, kWalkListInterior = 2;

// a voxel whose update is computed but whose re-warped value may still wait for the far corner's load
struct Deferred {
    Rewarp rw;
    float wv[3];
    int i;  // voxel index; < 0: nothing to finish
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, int WALK>
__global__ __launch_bounds__(WALK == kWalkListInterior ? kCuBlock : kBlock)
// <= 128 VGPRs for the dense walk (at 132 = 3 waves per SIMD it streamed 20 % slower) and the CU-sized workgroups; the
// general list walk (voxels on a face, arrays of 4 GiB and more) may take 3 waves' worth instead of spilling -- a list
// walk runs as fast with 3 waves per SIMD as with 4 (measured with 768-thread workgroups)
__attribute__((amdgpu_waves_per_eu(WALK == kWalkList ? 3 : 4, 4)))
void slavcheva_state_kernel(const vf4* __restrict__ state_in,
                                                                 const float* __restrict__ canonical,
                                                                 vf4* __restrict__ state_out, Grid g, Params p,
                                                                 lsf_gate gate, lsf_iteration_record* record,
                                                                 const int* __restrict__ band_list,
                                                                 unsigned band_count) {
#ifdef LSF_STATE_TRACE
    unsigned long long* wave_row = (g_state_waves && WALK == kWalkListInterior)
                                       ? g_state_waves + ((unsigned long long)blockIdx.x * kMaxBlockWaves + threadIdx.x / 64) * 8 : nullptr;
    if (wave_row && (threadIdx.x & 63) == 0) wave_row[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    double en[3] = {0.0, 0.0, 0.0};
    // second half of a voxel: the re-warped live value, the snap of field_warping.py:138-141, the store
    auto finish = [&](const Deferred& d) {
        if (d.i < 0) return;
        float v = d.rw.value();
        float wv[3] = {d.wv[0], d.wv[1], d.wv[2]};
        if (1.0f - fabsf(v) < 1e-6f) {  // field_warping.py:138-141
            v = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v);
            wv[0] = wv[1] = wv[2] = 0.0f;
        }
        vf4 o;
        o.x = v; o.y = wv[0]; o.z = wv[1]; o.w = wv[2];
        // the dense walk streams the whole state out (non-temporal: 0.164 against 0.173 ms at 256^3); a band list's
        // output is what the next launch reads first, and it is still in L2 / MALL then (0.0336 against 0.0384 ms)
        if (WALK == kWalkDense) __builtin_nontemporal_store(o, &state_out[d.i]);
        // ... unless the two states' listed voxels cannot stay in the 256 MB Infinity Cache anyway (512^3: 224 MB next to
        // canonical and lists): streaming stores are 2 % faster then (146.4 -> 143.7 us), 19 % slower at 256^3
        else if (g.list_store_nt) __builtin_nontemporal_store(o, &state_out[d.i]);
        else state_out[d.i] = o;
    };
    // first half: gradient, warp = -g * rate, its length for the arg-max (a18), the re-warp's taps (a3)
    auto voxel = [&](int x, int y, int z, int i, const vf4& sc, float cn) {
        Deferred d;
        d.i = i;
        const float l = sc.x;
        float gv[3] = {0.0f, 0.0f, 0.0f};
        // outside the narrow-band union (tsdf_set_routines.py:19-52; the `continue` of slavcheva_optimizer2d.py:251-252)
        const bool in_band = !(fabsf(l) == 1.0f && fabsf(cn) == 1.0f);
        d.rw.lerp = false;
        d.rw.R = l;  // zero displacement: every lerp is a*1 + b*0 = a exactly, the gather returns live[p] bit for bit
        d.rw.P = d.rw.Q = d.rw.wfz = d.rw.wfy = d.rw.wfx = d.rw.corner = 0.0f;
        bool from_taps = false;
        if (in_band) {
            double e[3] = {0.0, 0.0, 0.0};
            const bool interior = x > 0 && x < g.nx - 1 && y > 0 && y < g.ny - 1 && (D == 2 || (z > 0 && z < g.nz - 1));
            if (g.wide_ok && __all(interior) && wave_span_ok(g, i)) {
                NbhStateFast<D> n;
                n.load(state_in, g, (unsigned)i, sc, !g.fast_ok);
                // the scalar form of the terms: the packed one (INTERIOR lists) costs these walks registers they need
                band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, e);
                float w_now[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < D; ++c) w_now[c] = (-gv[c]) * p.rate;
                if (g.fast_ok) from_taps = rewarp_from_taps<D, true>(n, state_in, g, i, x, y, z, w_now, d.rw);
                else from_taps = rewarp_from_taps<D, false>(n, state_in, g, i, x, y, z, w_now, d.rw);
            } else {
                const NbhState<D> n(state_in, g, x, y, z, sc);
                band_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, e);
            }
            if (z >= g.e_begin && z < g.e_end) {  // a slab's recomputed halo slices belong to the neighbour's sums
                en[0] += e[0];
                en[1] += e[1];
                en[2] += e[2];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) d.wv[c] = c < D ? (-gv[c]) * p.rate : 0.0f;
        const float len = vec_length<D>(d.wv);
        const bool moved = !(d.wv[0] == 0.0f && d.wv[1] == 0.0f && d.wv[2] == 0.0f);
        if (!moved) {
            d.rw.lerp = false;
            d.rw.R = l;
        } else if (!from_taps) {
            const float px = (float)x + d.wv[0], py = (float)y + d.wv[1];
            const float pz = D == 3 ? (float)(z + g.z_global_offset) + d.wv[2] : 0.0f;
            d.rw.lerp = false;
            d.rw.R = state_gather<D>(state_in, g, px, py, pz);
        }
        const unsigned long long q = pack_max(len, linear_index(g, x, y, z));
        best = q > best ? q : best;
        return d;
    };
    // The same for a unit of an INTERIOR band list, as ONE straight-line block for the whole wave: every lane's 3^D
    // neighbourhood lies inside the array (lanes past the end of the list stand on the list's last voxel), so the 18
    // neighbourhood loads are issued unconditionally as soon as the voxel index is known, `previous` -- the unit before,
    // whose far-corner load was issued one unit ago -- is finished while they are in flight, and only then does the
    // arithmetic wait for them: ONE memory round trip per unit on the critical path instead of two (neighbourhood ->
    // arithmetic -> far corner -> store), which four waves per SIMD could not cover.  No exec-mask regions around the
    // loads either: a value loaded inside a divergent region leaves it through a copy, i.e. a wait inside the region.
#ifdef LSF_STATE_TRACE
    unsigned long long* trace_row = nullptr;
    unsigned trace_unit = 0u;
#endif
    auto interior_voxel = [&](unsigned i, const vf4& sc, float cn, bool listed, const Deferred& previous) {
        NbhStateFast<D> n;
        LSF_TRACE(0);
        n.load(state_in, g, i, sc, false);
        __builtin_amdgcn_sched_barrier(0);
        LSF_TRACE(1);
        finish(previous);
        __builtin_amdgcn_sched_barrier(0);
        LSF_TRACE(2);
#ifdef LSF_STATE_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSF_TRACE(3);
#endif
        int x, y, z;
        decode_voxel(g, i, x, y, z);
        Deferred d;
        d.i = listed ? (int)i : -1;
        const float l = sc.x;
        const bool in_band = listed && !(fabsf(l) == 1.0f && fabsf(cn) == 1.0f);
        float gv[3] = {0.0f, 0.0f, 0.0f};
        double e[3] = {0.0, 0.0, 0.0};
        fast_voxel_gradient<D, SMOOTH, LEVELSET, DATA, ENERGY>(n, p, l, cn, gv, e);
        const bool counted = in_band && z >= g.e_begin && z < g.e_end;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d.wv[c] = (c < D && in_band) ? (-gv[c]) * p.rate : 0.0f;
            en[c] += counted ? e[c] : 0.0;
        }
        const float len = vec_length<D>(d.wv);
        const bool moved = !(d.wv[0] == 0.0f && d.wv[1] == 0.0f && d.wv[2] == 0.0f);
        if (!rewarp_from_taps<D, true>(n, state_in, g, (int)i, x, y, z, d.wv, d.rw)) {
            d.rw.lerp = false;
            d.rw.R = state_gather<D>(state_in, g, (float)x + d.wv[0], (float)y + d.wv[1],
                                     D == 3 ? (float)(z + g.z_global_offset) + d.wv[2] : 0.0f);
        }
        if (!moved) {  // zero displacement: the gather returns live[p] bit for bit (every lerp is a*1 + b*0)
            d.rw.lerp = false;
            d.rw.R = l;
        }
        const unsigned long long q = listed ? pack_max(len, i + g.index_offset) : 0ull;  // = linear_index(g, x, y, z)
        best = q > best ? q : best;
        LSF_TRACE(4);
        return d;
    };
    Deferred nothing;
    nothing.i = -1;
    // Band list (lsf_band_list_fill): only voxels that can be in the narrow-band union are visited; every other voxel
    // holds (live, 0) in BOTH ping-pong states and never changes.  Their arg-max candidates are all (length 0, own
    // index); the smallest index of the launch's z-range stands for them.
    if (WALK == kWalkDense) {
        for_each_voxel(g, [&](int x, int y, int z) {
            const int i = vidx(g, x, y, z);
            finish(voxel(x, y, z, i, state_in[i], canonical[i]));
        });
    } else {
        // Software-pipelined list walk: a unit's critical path would be list entry -> state / canonical -> neighbourhood
        // -> arithmetic -> far corner of the re-warp cell -> store, five dependent memory round trips with only 4 waves
        // per SIMD to hide them.  The list entry is fetched two units ahead, the voxel's own state one unit ahead, and
        // (INTERIOR lists) a unit's far corner is consumed during the NEXT unit, so that a unit starts with its
        // neighbourhood loads and waits for memory once.  Lanes past the end of the list read its LAST entry (a valid,
        // listed voxel: their loads stay inside the array) and are not processed.
        const WaveWalk w = wave_list_walk(band_count, g.list_group);
        const unsigned waves = blockDim.x / kWave, wave = threadIdx.x / kWave;
        auto entry = [&](unsigned unit, bool& listed) {
            const unsigned k = unit * kWave + (threadIdx.x & (kWave - 1));
            listed = unit < w.x_end && k < band_count;
            return (unsigned)band_list[k < band_count ? k : band_count - 1u];
        };
        // Sequence numbers: the first two of a wave are fixed (wave, wave + waves); INTERIOR lists deal the rest out of
        // an LDS counter as waves come free -- the SIMD issues oldest-wave-first, so the first waves of a workgroup run
        // ahead of the last ones (traced: 21 us against 27 us for the same six units) and would leave their CU idle early.
        __shared__ unsigned s_next_unit;
        if (WALK == kWalkListInterior) {
            if (threadIdx.x == 0) s_next_unit = 2u * waves;
            __syncthreads();
        }
        unsigned n_last = wave + waves;
        auto grab = [&]() {
            if (WALK != kWalkListInterior) return w.unit(n_last += waves);
            unsigned v = 0u;
            if ((threadIdx.x & (kWave - 1)) == 0) v = atomicAdd(&s_next_unit, 1u);
            return w.unit((unsigned)__builtin_amdgcn_readfirstlane((int)v));
        };
        unsigned u = band_count ? w.unit(wave) : w.x_end, u1 = w.unit(wave + waves), u2 = 0u;  // nothing is read from an empty list
        bool in0 = false, in1 = false, in2 = false;
        unsigned i0 = 0u, i1 = 0u;
        vf4 s0 = {0.0f, 0.0f, 0.0f, 0.0f};
        float c0 = 0.0f;
        if (u < w.x_end) {
            i0 = entry(u, in0);
            i1 = entry(u1, in1);
            s0 = state_in[i0];
            c0 = canonical[i0];
        }
        Deferred pending = nothing;
#ifdef LSF_STATE_TRACE
        if (wave_row && (threadIdx.x & 63) == 0) {
            wave_row[1] = __builtin_amdgcn_s_memrealtime();
            wave_row[4] = __builtin_readcyclecounter();
        }
#endif
        while (u < w.x_end) {
            u2 = grab();
            const vf4 s1 = state_in[i1];
            const float c1 = canonical[i1];
            const unsigned i2 = entry(u2, in2);
            if (WALK == kWalkListInterior) {
#ifdef LSF_STATE_TRACE
                const unsigned nth = trace_unit++;
                trace_row = (g_state_trace && blockIdx.x < kTraceBlocks && nth < kTraceUnits)
                                ? g_state_trace + ((((unsigned long long)blockIdx.x * kMaxBlockWaves + threadIdx.x / 64) * kTraceUnits + nth) * kTraceStamps)
                                : nullptr;
                if (trace_row && (threadIdx.x & 63) == 0) {
                    trace_row[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
                    trace_row[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
                }
#endif
                pending = interior_voxel(i0, s0, c0, in0, pending);
                LSF_TRACE(5);
#ifdef LSF_STATE_NO_DEFER  // measurement only (tools/build_variant.sh): the far corner is waited for at once
                finish(pending);
                pending = nothing;
#endif
            } else if (in0) {
                const unsigned zy = fast_div(i0, g.div_nx);
                const int x = (int)(i0 - zy * (unsigned)g.nx);
                const int z = (int)fast_div(zy, g.div_ny);
                const int y = (int)zy - z * g.ny;
                finish(voxel(x, y, z, (int)i0, s0, c0));
            }
            u = u1;
            u1 = u2;
            i0 = i1; in0 = in1; s0 = s1; c0 = c1;
            i1 = i2; in1 = in2;
        }
        finish(pending);
#ifdef LSF_STATE_TRACE
        if (wave_row && (threadIdx.x & 63) == 0) {
            wave_row[2] = __builtin_amdgcn_s_memrealtime();
            wave_row[5] = __builtin_readcyclecounter();
            wave_row[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            wave_row[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        }
#endif
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
            best = q > best ? q : best;
        }
    }
    double* dst[3] = {ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->data_energy : nullptr,
                      ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->smoothing_energy : nullptr,
                      ENERGY != LSF_ENERGY_NONE ? &record_slot(record)->level_set_energy : nullptr};
    block_reduce_commit<3>(best, en, record_max(record), dst);
#ifdef LSF_STATE_TRACE
    if (wave_row && (threadIdx.x & 63) == 0) wave_row[3] = __builtin_amdgcn_s_memrealtime();
#endif
}

// (live, warp planar or 0) -> state, optionally into two buffers (both ping-pong states start equal)
__global__ __launch_bounds__(kBlock) void state_pack_kernel(const float* __restrict__ live,
                                                            const float* __restrict__ warp, vf4* __restrict__ a,
                                                            vf4* __restrict__ b, long long first, long long n,
                                                            long long plane, int dims) {
    for (long long k = blockIdx.x * (long long)kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock) {
        const long long i = first + k;
        vf4 o;
        o.x = live[i];
        o.y = warp ? warp[i] : 0.0f;
        o.z = warp ? warp[plane + i] : 0.0f;
        o.w = (warp && dims == 3) ? warp[2 * plane + i] : 0.0f;
        a[i] = o;
        if (b) b[i] = o;
    }
}

// state -> live, warp planar [c][z][y][x] and / or interleaved [z][y][x][c] (any of them may be NULL)
__global__ __launch_bounds__(kBlock) void state_unpack_kernel(const vf4* __restrict__ s, float* __restrict__ live,
                                                              float* __restrict__ planar,
                                                              float* __restrict__ interleaved, long long first,
                                                              long long n, long long plane, int dims) {
    for (long long k = blockIdx.x * (long long)kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock) {
        const long long i = first + k;
        const vf4 v = s[i];
        if (live) live[i] = v.x;
        if (planar) {
            planar[i] = v.y;
            planar[plane + i] = v.z;
            if (dims == 3) planar[2 * plane + i] = v.w;
        }
        if (interleaved) {
            interleaved[i * dims] = v.y;
            interleaved[i * dims + 1] = v.z;
            if (dims == 3) interleaved[i * dims + 2] = v.w;
        }
    }
}

// ---- end of an optimize() call: state -> caller's fields AND the convergence statistics (a20) in ONE pass ------------
// Per-block partial results go to scratch rows (no same-address atomics: ~2000 of them serialise at ~12 ns each and
// made the two separate statistics kernels take 0.35 ms at 256^3 for 0.13 GB of reads); a one-block kernel combines.
constexpr int kFinalizeWords = 12;  // per block: packed warp max, packed diff max, diff min, 4 warp sums, 2 diff sums

struct FinalizeAccumulator {
    unsigned long long best_w = 0ull, best_d = 0ull;
    double sums[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // band count, above lo, sum len, sum len^2, sum d, sum d^2
    double mn = __longlong_as_double(0x7ff0000000000000ll);

    __device__ inline void add(const vf4& v, float cn, unsigned lin, int dims, float lo) {
        if (!(fabsf(v.x) == 1.0f && fabsf(cn) == 1.0f)) {
            const float wv[3] = {v.y, v.z, v.w};
            const float len = dims == 3 ? vec_length<3>(wv) : vec_length<2>(wv);
            const unsigned long long p = pack_max(len, lin);
            best_w = p > best_w ? p : best_w;
            sums[0] += 1.0;
            sums[1] += len > lo ? 1.0 : 0.0;
            sums[2] += (double)len;
            sums[3] += (double)len * (double)len;
        }
        const double d = fabs((double)cn - (double)v.x);
        const unsigned long long q = pack_max((float)d, lin);
        best_d = q > best_d ? q : best_d;
        sums[4] += d;
        sums[5] += d * d;
        mn = fmin(mn, d);
    }

    // block reduction -> one scratch row (no same-address atomics)
    __device__ inline void store_row(double* __restrict__ row) {
        __shared__ unsigned long long s_w[kBlock / kWave], s_d[kBlock / kWave];
        __shared__ double s_mn[kBlock / kWave], s_sum[6][kBlock / kWave];
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        best_w = wave_max_u64(best_w);
        best_d = wave_max_u64(best_d);
        for (int dl = kWave / 2; dl > 0; dl >>= 1) mn = fmin(mn, shfl_down_f64(mn, dl));
#pragma unroll
        for (int j = 0; j < 6; ++j) sums[j] = wave_sum_f64(sums[j]);
        if (lane == 0) {
            s_w[wave] = best_w; s_d[wave] = best_d; s_mn[wave] = mn;
#pragma unroll
            for (int j = 0; j < 6; ++j) s_sum[j][wave] = sums[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long w = 0ull, d = 0ull;
            double m = s_mn[0];
            for (int k = 0; k < kBlock / kWave; ++k) {
                w = s_w[k] > w ? s_w[k] : w;
                d = s_d[k] > d ? s_d[k] : d;
                m = fmin(m, s_mn[k]);
            }
            row[0] = __longlong_as_double((long long)w);
            row[1] = __longlong_as_double((long long)d);
            row[2] = m;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double t = 0.0;
                for (int k = 0; k < kBlock / kWave; ++k) t += s_sum[j][k];
                row[3 + j] = t;
            }
        }
    }
};

// s == nullptr: the final fields are planar (the SobolevFusion path): src_live and src_warp [c][z][y][x]
__global__ __launch_bounds__(kBlock) void state_finalize_kernel(const vf4* __restrict__ s,
                                                                const float* __restrict__ src_live,
                                                                const float* __restrict__ src_warp,
                                                                const float* __restrict__ canonical,
                                                                float* __restrict__ live, float* __restrict__ planar,
                                                                float* __restrict__ interleaved, long long first,
                                                                long long n, long long plane, int dims,
                                                                long long index_offset, float lo,
                                                                double* __restrict__ scratch) {
    FinalizeAccumulator acc;
    for (long long k = blockIdx.x * (long long)kBlock + threadIdx.x; k < n; k += (long long)gridDim.x * kBlock) {
        const long long i = first + k;
        vf4 v;
        if (s) {
            v = s[i];
        } else {
            v.x = src_live[i];
            v.y = src_warp[i];
            v.z = src_warp[plane + i];
            v.w = dims == 3 ? src_warp[2 * plane + i] : 0.0f;
        }
        if (live) live[i] = v.x;
        if (planar) {
            planar[i] = v.y;
            planar[plane + i] = v.z;
            if (dims == 3) planar[2 * plane + i] = v.w;
        }
        if (interleaved) {
            interleaved[i * dims] = v.y;
            interleaved[i * dims + 1] = v.z;
            if (dims == 3) interleaved[i * dims + 2] = v.w;
        }
        if (scratch) acc.add(v, canonical[i], (unsigned)(i + index_offset), dims, lo);
    }
    if (scratch) acc.store_row(scratch + (long long)blockIdx.x * kFinalizeWords);
}

// the same for the voxels of a band list only: everything else still holds (live, 0) -- the caller's live field is
// already right there, its warp is zero-filled, and what those voxels contribute to the statistics was counted by
// lsf_state_prepare (|canonical - live| is 0 or exactly 2 outside the band, for ever)
__global__ __launch_bounds__(kBlock) void state_finalize_list_kernel(const vf4* __restrict__ s,
                                                                     const float* __restrict__ canonical,
                                                                     float* __restrict__ live,
                                                                     float* __restrict__ interleaved,
                                                                     const int* __restrict__ list, unsigned count,
                                                                     int dims, long long index_offset, float lo,
                                                                     double* __restrict__ scratch) {
    FinalizeAccumulator acc;
    for (unsigned k = blockIdx.x * kBlock + threadIdx.x; k < count; k += gridDim.x * kBlock) {
        const long long i = list[k];
        const vf4 v = s[i];
        if (live) live[i] = v.x;
        if (interleaved) {
            interleaved[i * dims] = v.y;
            interleaved[i * dims + 1] = v.z;
            if (dims == 3) interleaved[i * dims + 2] = v.w;
        }
        if (scratch) acc.add(v, canonical[i], (unsigned)(i + index_offset), dims, lo);
    }
    if (scratch) acc.store_row(scratch + (long long)blockIdx.x * kFinalizeWords);
}

// one block: rows -> out16 = the 8-double layouts of lsf_warp_statistics and lsf_tsdf_difference_statistics
// unlisted: the voxels no row covers (list variant): `opposite` of them have |canonical - live| = 2 (the first one at
// linear index `first_opposite`), the rest 0; first_voxel = linear index of the array's first voxel
__global__ __launch_bounds__(kBlock) void state_finalize_combine_kernel(const double* __restrict__ scratch,
                                                                        unsigned rows, double voxels,
                                                                        double* __restrict__ out16, double unlisted,
                                                                        double opposite, long long first_opposite,
                                                                        long long first_voxel) {
    unsigned long long best_w = 0ull, best_d = 0ull;
    double sums[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double mn = __longlong_as_double(0x7ff0000000000000ll);
    for (unsigned r = threadIdx.x; r < rows; r += kBlock) {
        const double* row = scratch + (long long)r * kFinalizeWords;
        const unsigned long long w = (unsigned long long)__double_as_longlong(row[0]);
        const unsigned long long d = (unsigned long long)__double_as_longlong(row[1]);
        best_w = w > best_w ? w : best_w;
        best_d = d > best_d ? d : best_d;
        mn = fmin(mn, row[2]);
#pragma unroll
        for (int j = 0; j < 6; ++j) sums[j] += row[3 + j];
    }
    __shared__ unsigned long long s_w[kBlock / kWave], s_d[kBlock / kWave];
    __shared__ double s_mn[kBlock / kWave], s_sum[6][kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    best_w = wave_max_u64(best_w);
    best_d = wave_max_u64(best_d);
    for (int dl = kWave / 2; dl > 0; dl >>= 1) mn = fmin(mn, shfl_down_f64(mn, dl));
#pragma unroll
    for (int j = 0; j < 6; ++j) sums[j] = wave_sum_f64(sums[j]);
    if (lane == 0) {
        s_w[wave] = best_w; s_d[wave] = best_d; s_mn[wave] = mn;
#pragma unroll
        for (int j = 0; j < 6; ++j) s_sum[j][wave] = sums[j];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned long long w = 0ull, d = 0ull;
    double m = s_mn[0], t[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < kBlock / kWave; ++k) {
        w = s_w[k] > w ? s_w[k] : w;
        d = s_d[k] > d ? s_d[k] : d;
        m = fmin(m, s_mn[k]);
#pragma unroll
        for (int j = 0; j < 6; ++j) t[j] += s_sum[j][k];
    }
    if (unlisted > opposite) m = fmin(m, 0.0);
    if (opposite > 0.0) {
        m = fmin(m, 2.0);
        const unsigned long long q = pack_max(2.0f, (unsigned)first_opposite);
        d = q > d ? q : d;
        t[4] += 2.0 * opposite;
        t[5] += 4.0 * opposite;
    }
    // the maximum is 0: every voxel ties and the first one of the array wins, listed or not
    if (unlisted > 0.0 && (d >> 32) == 0ull) d = pack_max(0.0f, (unsigned)first_voxel);
    // warp: [count_band, count_above_lo, max_len, sum_len, sum_len^2, argmax, 0, 0]
    out16[0] = t[0]; out16[1] = t[1];
    out16[2] = w ? (double)unpack_max_value(w) : 0.0;
    out16[3] = t[2]; out16[4] = t[3];
    out16[5] = w ? (double)(~(unsigned)w) : -1.0;
    out16[6] = 0.0; out16[7] = 0.0;
    // tsdf: [count, min, max, sum, sum^2, argmax, 0, 0] of |canonical - live|
    out16[8] = voxels; out16[9] = m;
    out16[10] = d ? (double)unpack_max_value(d) : 0.0;
    out16[11] = t[4]; out16[12] = t[5];
    out16[13] = d ? (double)(~(unsigned)d) : -1.0;
    out16[14] = 0.0; out16[15] = 0.0;
}

struct LaunchArgs {
    unsigned blocks, threads;
    hipStream_t s;
    const vf4* state_in;
    const float* canonical;
    vf4* state_out;
    Grid g;
    Params p;
    lsf_gate gate;
    lsf_iteration_record* record;
    const int* band_list;
    unsigned band_count;
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, int WALK>
void launch_one(const LaunchArgs& a) {
    hipLaunchKernelGGL((slavcheva_state_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY, WALK>), dim3(a.blocks),
                       dim3(a.threads), 0, a.s, a.state_in, a.canonical, a.state_out, a.g, a.p, a.gate,
                       a.record, a.band_list, a.band_count);
}

template <int D, int SMOOTH, bool LEVELSET, int DATA, int WALK>
void pick_energy(int energy, const LaunchArgs& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT, WALK>(a); break;
        case LSF_ENERGY_VECTORIZED: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED, WALK>(a); break;
        default: launch_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE, WALK>(a); break;
    }
}

template <int D, int WALK>
void pick_terms(const lsf_slavcheva_params* q, const LaunchArgs& a) {
    const bool killing = q->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = q->level_set_enabled != 0;
    const bool fdm = q->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = q->energy_mode;
#define LSF_PICK(S, L, DM) pick_energy<D, S, L, DM, WALK>(e, a)
    if (killing) {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_BASIC); }
    } else {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_BASIC); }
    }
#undef LSF_PICK
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

// CUs of the current device (256 on an MI355X in SPX mode; a partition has fewer), asked once per device
inline unsigned compute_units() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (const char* e = getenv("LSF_LIST_BLOCKS")) {  // measurement knob
            const int v = atoi(e);
            if (v > 0) n = v;
        }
        cached[dev] = n;
    }
    return (unsigned)cached[dev];
}

inline bool range_of(const lsf_grid* grid, long long& first, long long& n) {
    const long long slice = (long long)grid->ny * grid->nx;
    first = slice * grid->z_begin;
    n = slice * (grid->z_end - grid->z_begin);
    return n > 0;
}

inline unsigned stream_blocks(long long n, long long cap = 8192) {
    const long long want = (n + kBlock - 1) / kBlock;
    return (unsigned)(want < cap ? want : cap);
}

// finalize: 8 blocks per CU keep enough loads in flight; fewer rows for the one-block combine kernel
inline unsigned finalize_blocks(long long n) { return stream_blocks(n, 2048); }

}  // namespace

#ifdef LSF_STATE_TRACE
extern "C" int lsf_debug_set_state_trace(void* units, void* waves) {
    if (int e = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_state_trace), &units, sizeof(units))) return e;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_state_waves), &waves, sizeof(waves));
}
#endif

extern "C" int lsf_state_pack(const float* live, const float* warp_planar, float* state_a, float* state_b,
                              const lsf_grid* grid, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !state_a) return LSF_ERR_BAD_ARGUMENT;
    long long first, n;
    if (!range_of(grid, first, n)) return 0;
    hipLaunchKernelGGL(state_pack_kernel, dim3(stream_blocks(n)), dim3(kBlock), 0, as_stream(stream), live, warp_planar,
                       reinterpret_cast<vf4*>(state_a), reinterpret_cast<vf4*>(state_b), first, n,
                       (long long)grid->nz * grid->ny * grid->nx, grid->dims);
    return launch_status();
}

extern "C" int lsf_state_unpack(const float* state, float* live_out, float* warp_planar_out,
                                float* warp_interleaved_out, const lsf_grid* grid, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state) return LSF_ERR_BAD_ARGUMENT;
    long long first, n;
    if (!range_of(grid, first, n)) return 0;
    hipLaunchKernelGGL(state_unpack_kernel, dim3(stream_blocks(n)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const vf4*>(state), live_out, warp_planar_out, warp_interleaved_out, first, n,
                       (long long)grid->nz * grid->ny * grid->nx, grid->dims);
    return launch_status();
}

extern "C" int64_t lsf_state_finalize_scratch_elements(const lsf_grid* grid) {
    if (check_grid(grid)) return 0;
    long long first, n;
    range_of(grid, first, n);
    return (int64_t)finalize_blocks(n > 0 ? n : 1) * kFinalizeWords;
}

extern "C" int lsf_state_finalize(const float* state, const float* canonical, float* live_out,
                                  float* warp_planar_out, float* warp_interleaved_out, const lsf_grid* grid,
                                  float lower_threshold, double* statistics16, double* scratch, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state || (statistics16 && (!canonical || !scratch))) return LSF_ERR_BAD_ARGUMENT;
    long long first, n;
    if (!range_of(grid, first, n)) return statistics16 ? LSF_ERR_BAD_ARGUMENT : 0;
    const unsigned blocks = finalize_blocks(n);
    const long long slice = (long long)grid->ny * grid->nx;
    hipLaunchKernelGGL(state_finalize_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const vf4*>(state), (const float*)nullptr, (const float*)nullptr, canonical,
                       live_out, warp_planar_out, warp_interleaved_out, first, n, (long long)grid->nz * slice, grid->dims,
                       slice * grid->z_global_offset, lower_threshold, statistics16 ? scratch : (double*)nullptr);
    if (statistics16)
        hipLaunchKernelGGL(state_finalize_combine_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), scratch, blocks,
                           (double)n, statistics16, 0.0, 0.0, 0ll, 0ll);
    return launch_status();
}

extern "C" int lsf_planar_finalize(const float* live, const float* warp_planar, const float* canonical, float* live_out,
                                   float* warp_interleaved_out, const lsf_grid* grid, float lower_threshold,
                                   double* statistics16, double* scratch, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !warp_planar || live_out == live || (statistics16 && (!canonical || !scratch)))
        return LSF_ERR_BAD_ARGUMENT;
    long long first, n;
    if (!range_of(grid, first, n)) return statistics16 ? LSF_ERR_BAD_ARGUMENT : 0;
    const unsigned blocks = finalize_blocks(n);
    const long long slice = (long long)grid->ny * grid->nx;
    hipLaunchKernelGGL(state_finalize_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), (const vf4*)nullptr, live,
                       warp_planar, canonical, live_out, (float*)nullptr, warp_interleaved_out, first, n,
                       (long long)grid->nz * slice, grid->dims, slice * grid->z_global_offset, lower_threshold,
                       statistics16 ? scratch : (double*)nullptr);
    if (statistics16)
        hipLaunchKernelGGL(state_finalize_combine_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), scratch, blocks,
                           (double)n, statistics16, 0.0, 0.0, 0ll, 0ll);
    return launch_status();
}

extern "C" int lsf_state_finalize_listed(const float* state, const float* canonical, float* live_out,
                                         float* warp_interleaved_out, const lsf_grid* grid,
                                         const int32_t* const* band_lists, const int64_t* band_counts, int32_t n_lists,
                                         int64_t opposite_count, int64_t first_opposite, float lower_threshold,
                                         double* statistics16, double* scratch, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state || n_lists < 0 || n_lists > 2 || (n_lists && (!band_lists || !band_counts)) ||
        (statistics16 && (!canonical || !scratch)) || grid->z_begin != 0 || grid->z_end != grid->nz)
        return LSF_ERR_BAD_ARGUMENT;
    long long first, n;
    if (!range_of(grid, first, n)) return statistics16 ? LSF_ERR_BAD_ARGUMENT : 0;
    unsigned rows = 0;
    long long listed = 0;
    for (int32_t k = 0; k < n_lists; ++k)  // everything is checked before anything is launched
        if (band_counts[k] < 0 || band_counts[k] > 0x7fffffffll || (band_counts[k] && !band_lists[k])) return LSF_ERR_BAD_ARGUMENT;
    for (int32_t k = 0; k < n_lists; ++k) {
        if (band_counts[k] == 0) continue;
        const unsigned blocks = finalize_blocks(band_counts[k]);
        hipLaunchKernelGGL(state_finalize_list_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream),
                           reinterpret_cast<const vf4*>(state), canonical, live_out, warp_interleaved_out, band_lists[k],
                           (unsigned)band_counts[k], grid->dims, (long long)grid->ny * grid->nx * grid->z_global_offset,
                           lower_threshold, statistics16 ? scratch + (long long)rows * kFinalizeWords : (double*)nullptr);
        rows += blocks;
        listed += band_counts[k];
    }
    if (statistics16) {
        const long long first_voxel = (long long)grid->ny * grid->nx * grid->z_global_offset;
        hipLaunchKernelGGL(state_finalize_combine_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), scratch, rows,
                           (double)n, statistics16, (double)(n - listed), (double)opposite_count,
                           (long long)first_opposite + first_voxel, first_voxel);
    }
    return launch_status();
}

extern "C" int lsf_slavcheva_state_iteration(const float* state_in, const float* canonical, float* state_out,
                                             const lsf_grid* grid, const lsf_slavcheva_params* params,
                                             const lsf_gate* gate, lsf_iteration_record* record,
                                             const int32_t* band_list, int64_t band_count, int32_t band_subset,
                                             void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state_in || !canonical || !state_out || state_out == state_in || !params || !record)
        return LSF_ERR_BAD_ARGUMENT;
    const int tile_y = 4;
    Grid g = make_grid(grid, tile_y);
    g.fast_ok = g.plane * 16 < 0xffffffffll;  // 32-bit byte offsets into the whole float4 state
    // the buffer-load neighbourhood needs its scalar tap offsets (two slices + two rows) to fit 32 bits; arrays of
    // 4 GiB and more then qualify wave by wave (wave_span_ok)
    g.wide_ok = 16ll * (2ll * grid->nx * grid->ny + 2ll * grid->nx + 3) < 0x7fffffffll;
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    const bool listed = band_list != nullptr;
    if (listed && (band_count < 0 || band_count > 0x7fffffffll ||
                   !(band_subset == LSF_BAND_ALL || band_subset == LSF_BAND_INTERIOR || band_subset == LSF_BAND_BOUNDARY)))
        return LSF_ERR_BAD_ARGUMENT;
    // the INTERIOR kernel has no other neighbourhood path: on arrays of 4 GiB and more an INTERIOR list runs through the
    // general list kernel, which takes the same fast path wave by wave
    const bool all_interior = listed && band_subset == LSF_BAND_INTERIOR && g.fast_ok;
    // INTERIOR lists: one CU-sized workgroup per CU (wave_list_walk); the general list kernel (voxels on a face of the
    // array, arrays of 4 GiB and more) needs a few more registers than a CU-sized workgroup may have and keeps 256-thread
    // workgroups, every one the same number of 64-entry wave-units per wave
    const unsigned blocks = all_interior ? cu_list_blocks((unsigned)band_count, compute_units())
                            : listed     ? band_list_blocks((unsigned)band_count, 128u)
                                         : launch_blocks(t.total, blocks_per_xcd());
    static const unsigned list_threads = [] {  // measurement knob: waves per CU of the INTERIOR list walk
        const char* e = getenv("LSF_LIST_THREADS");
        const int v = e ? atoi(e) : 0;
        return (v >= 64 && v <= kCuBlock && v % 64 == 0) ? (unsigned)v : (unsigned)kCuBlock;
    }();
    static const unsigned list_group = [] {  // measurement knob: wave-units per group of the list walk
        const char* e = getenv("LSF_LIST_GROUP");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? (unsigned)v : 0u;
    }();
    if (list_group) g.list_group = list_group;
    g.list_store_nt = listed && band_count * 32ll > 200ll * 1000 * 1000;
    LaunchArgs a{blocks, all_interior ? list_threads : (unsigned)(kTileX * tile_y), as_stream(stream), reinterpret_cast<const vf4*>(state_in), canonical,
                 reinterpret_cast<vf4*>(state_out), g, make_params(params), gate_or_open(gate), record, band_list,
                 (unsigned)band_count};
    if (grid->dims == 2) {
        if (all_interior) pick_terms<2, kWalkListInterior>(params, a);
        else if (listed) pick_terms<2, kWalkList>(params, a);
        else pick_terms<2, kWalkDense>(params, a);
    } else {
        if (all_interior) pick_terms<3, kWalkListInterior>(params, a);
        else if (listed) pick_terms<3, kWalkList>(params, a);
        else pick_terms<3, kWalkDense>(params, a);
    }
    return launch_status();
}
