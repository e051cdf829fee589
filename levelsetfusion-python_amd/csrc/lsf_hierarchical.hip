// Hierarchical optimizer iteration kernels.
// One launch = one pass of the loop body of nonrigid_opt/hierarchical/hierarchical_optimizer2d.py:184-225
// over one pyramid level, dimension-generic (D = 2, 3).
//
// Data layout: the live field and its D np.gradient components are STATIC per optimize()
// call and are only ever read through the data-dependent D-linear gather, so they are packed as one float4 per
// voxel (live, gx, gy, gz): one 16-byte load per tap instead of four scattered dwords.  warp / gradient are
// planar ([c][z][y][x]) so that the pointwise and 7-point-stencil accesses are unit-stride dword streams.
#include "lsf_device.h"

using namespace lsf;

namespace {

struct Packed {
    float l, gx, gy, gz;
};

__device__ inline Packed select_packed(const float4& v, bool valid) {
    // OOB taps: live reads 1 (field_warping.py:67-85), gradients read 0 (:88-109 with replacement 0)
    Packed p;
    p.l = valid ? v.x : 1.0f;
    p.gx = valid ? v.y : 0.0f;
    p.gy = valid ? v.z : 0.0f;
    p.gz = valid ? v.w : 0.0f;
    return p;
}

__device__ inline Packed lerp(const Packed& a, const Packed& b, float inv, float ratio) {
    Packed r;
    r.l = a.l * inv + b.l * ratio;
    r.gx = a.gx * inv + b.gx * ratio;
    r.gy = a.gy * inv + b.gy * ratio;
    r.gz = a.gz * inv + b.gz * ratio;
    return r;
}

// D-linear gather of the packed field: lerp along z, then y, then x (oracle.sample_linear); taps are read from
// clamped coordinates and replaced by select when out of bounds (no branches around the 16-byte loads)
template <int D>
__device__ inline Packed gather_packed(const float4* __restrict__ s, const Grid& g, float px, float py, float pz) {
    const AxisTaps ax = axis_taps(px, g.nx, 0), ay = axis_taps(py, g.ny, 0);
    const int row0 = ay.c0 * g.nx, row1 = ay.c1 * g.nx;
    Packed c[2][2];
    if (D == 2) {
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy) {
                const int xy = (oy ? row1 : row0) + (ox ? ax.c1 : ax.c0);
                c[ox][oy] = select_packed(s[xy], (ox ? ax.v1 : ax.v0) && (oy ? ay.v1 : ay.v0));
            }
    } else {
        // the gather operand may hold more slices than the launch's grid (a z-slab run whose warps outgrow the halo
        // reads a replicated copy of the whole level: lsf_hier_params::packed_nz)
        const AxisTaps az = axis_taps(pz, g.gather_nz, g.gather_z_offset);
        const int slice = g.nx * g.ny;
        const int s0 = az.c0 * slice, s1 = az.c1 * slice;
#pragma unroll
        for (int ox = 0; ox < 2; ++ox)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy) {
                const int xy = (oy ? row1 : row0) + (ox ? ax.c1 : ax.c0);
                const bool vxy = (ox ? ax.v1 : ax.v0) && (oy ? ay.v1 : ay.v0);
                const Packed a = select_packed(s[s0 + xy], vxy && az.v0);
                const Packed b = select_packed(s[s1 + xy], vxy && az.v1);
                c[ox][oy] = lerp(a, b, az.i, az.r);
            }
    }
    const Packed i0 = lerp(c[0][0], c[0][1], ay.i, ay.r);
    const Packed i1 = lerp(c[1][0], c[1][1], ay.i, ay.r);
    return lerp(i0, i1, ax.i, ax.r);
}

// scipy.ndimage.laplace(mode='nearest') of one plane at (x,y,z): per-axis second differences evaluated in
// double and stored float32, summed in float32, slowest axis first (oracle.laplace_replicate)
// ENERGY: also adds this plane's share of the reference's "tikhonov energy" printout to *energy -- the sum over the
// axes of np.gradient(previous gradient)^2 (hierarchical_optimizer2d.py:204-210; central differences inside, one-sided
// at the border: the clamped neighbour is the centre there)
template <int D, bool ENERGY>
__device__ inline float laplace_replicate(const float* __restrict__ a, const Grid& g, int x, int y, int z,
                                          double* energy) {
    // neighbours are read from CLAMPED offsets (a missing neighbour IS the centre in mode='nearest'): unconditional
    // loads the compiler can issue together, instead of one exec-masked load + wait per neighbour
    const int i = vidx(g, x, y, z);
    const float a0 = a[i];
    const int sy = g.nx, sz = g.nx * g.ny;
    float out;
    const float ym = a[i - (y > 0 ? sy : 0)], yp = a[i + (y < g.ny - 1 ? sy : 0)];
    const float xm = a[i - (x > 0 ? 1 : 0)], xp = a[i + (x < g.nx - 1 ? 1 : 0)];
    float d2y = second_difference_f64(ym, a0, yp);
    float d2x = second_difference_f64(xm, a0, xp);
    if (ENERGY) {
        const float gy = (y > 0 && y < g.ny - 1) ? (yp - ym) * 0.5f : (g.ny > 1 ? yp - ym : 0.0f);
        const float gx = (x > 0 && x < g.nx - 1) ? (xp - xm) * 0.5f : (g.nx > 1 ? xp - xm : 0.0f);
        *energy += (double)(gy * gy) + (double)(gx * gx);
    }
    if (D == 3) {
        const float zm = a[i - (z > 0 ? sz : 0)], zp = a[i + (z < g.nz - 1 ? sz : 0)];
        float d2z = second_difference_f64(zm, a0, zp);
        if (ENERGY) {
            const float gz = (z > 0 && z < g.nz - 1) ? (zp - zm) * 0.5f : (g.nz > 1 ? zp - zm : 0.0f);
            *energy += (double)(gz * gz);
        }
        out = d2z + d2y;
    } else {
        out = d2y;
    }
    return out + d2x;
}

// PREVMAX: also the maximum of |g_prev| into gate.prev_record (lsf_hier_params::previous_max)
template <int D, bool TIK, bool UPDATE, bool ENERGY, bool PREVMAX = false>
__global__ __launch_bounds__(kBlock) void hier_iteration_kernel(const float4* __restrict__ packed,
                                                                const float* __restrict__ canonical,
                                                                float* __restrict__ warp,
                                                                const float* __restrict__ g_prev,
                                                                float* __restrict__ g_out, Grid g, float amp,
                                                                float strength, float rate, lsf_gate gate,
                                                                lsf_iteration_record* record) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    double sums[2] = {0.0, 0.0};  // data energy sum(diff^2); tikhonov energy sum |np.gradient(previous gradient)|^2
    for_each_voxel(g, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        // streamed once per iteration (warp, canonical, the gradient written): non-temporal, so that they do not push
        // the REUSED lines -- the packed field's z + 1 slice and the previous gradient's z -/+ 1 slices, needed again
        // one slice later -- out of the 4 MB L2 (a 256^2 slice of everything is 4.3 MB)
        float w[3];
        w[0] = __builtin_nontemporal_load(warp + i);
        w[1] = __builtin_nontemporal_load(warp + g.plane + i);
        w[2] = D == 3 ? __builtin_nontemporal_load(warp + 2 * g.plane + i) : 0.0f;
        const float px = (float)x + w[0], py = (float)y + w[1];
        const float pz = D == 3 ? (float)(z + g.z_global_offset) + w[2] : 0.0f;
        const Packed s = gather_packed<D>(packed, g, px, py, pz);
        const float diff = s.l - __builtin_nontemporal_load(canonical + i);
        const float live_grad[3] = {s.gx, s.gy, s.gz};
        float gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < D; ++c) {
            float gd = diff * live_grad[c];
            if (TIK) {
                float lap = laplace_replicate<D, ENERGY>(g_prev + c * g.plane, g, x, y, z, &sums[1]);
                gv[c] = amp * gd - strength * lap;
            } else {
                gv[c] = amp * gd;
            }
            if (g_out) __builtin_nontemporal_store(gv[c], g_out + c * g.plane + i);
        }
        if (UPDATE) {
#pragma unroll
            for (int c = 0; c < D; ++c) __builtin_nontemporal_store(w[c] - rate * gv[c], warp + c * g.plane + i);
            unsigned long long p = pack_max(vec_length<D>(gv), linear_index(g, x, y, z));
            best = p > best ? p : best;
        }
        if (PREVMAX) {  // the centre values the Laplacian has loaded already
            float gp[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < D; ++c) gp[c] = g_prev[c * g.plane + i];
            unsigned long long p = pack_max(vec_length<D>(gp), linear_index(g, x, y, z));
            best = p > best ? p : best;
        }
        if (ENERGY) sums[0] += (double)diff * (double)diff;
    });
    if (UPDATE || ENERGY || PREVMAX) {
        double* dst[2] = {ENERGY ? &record_slot(record)->data_energy : nullptr,
                          ENERGY && TIK ? &record_slot(record)->smoothing_energy : nullptr};
        unsigned long long* max_dst = UPDATE ? record_max(record) : nullptr;
        if (PREVMAX) max_dst = record_max(const_cast<lsf_iteration_record*>(gate.prev_record));
        block_reduce_commit<2>(best, sums, max_dst, dst);
    }
}

// MOVE = false: only the record's maximum (the filter kernel has already moved the warp: lsf_convolve_xyz)
template <int D, bool MOVE>
__global__ __launch_bounds__(kBlock) void hier_update_kernel(const float* __restrict__ gfield,
                                                             float* __restrict__ warp, Grid g, float rate,
                                                             lsf_gate gate, lsf_iteration_record* record) {
    if (gate_closed(gate)) return;
    unsigned long long best = 0ull;
    for_each_voxel(g, [&](int x, int y, int z) {
        const int i = vidx(g, x, y, z);
        float gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < D; ++c) {
            gv[c] = gfield[c * g.plane + i];
            if (MOVE) warp[c * g.plane + i] = warp[c * g.plane + i] - rate * gv[c];
        }
        unsigned long long p = pack_max(vec_length<D>(gv), linear_index(g, x, y, z));
        best = p > best ? p : best;
    });
    const double sums[1] = {0.0};
    double* dst[1] = {nullptr};
    block_reduce_commit<0>(best, sums, record_max(record), dst);
}

template <int D, bool TIK, bool UPDATE>
void launch_hier(bool energy, unsigned blocks, hipStream_t s, const float4* packed, const float* canonical,
                 float* warp, const float* g_prev, float* g_out, const Grid& g, const lsf_hier_params* p,
                 const lsf_gate& gate, lsf_iteration_record* record) {
    if (energy)
        hipLaunchKernelGGL((hier_iteration_kernel<D, TIK, UPDATE, true>), dim3(blocks), dim3(kBlock), 0, s, packed,
                           canonical, warp, g_prev, g_out, g, p->data_term_amplifier, p->tikhonov_strength, p->rate,
                           gate, record);
    else
        hipLaunchKernelGGL((hier_iteration_kernel<D, TIK, UPDATE, false>), dim3(blocks), dim3(kBlock), 0, s, packed,
                           canonical, warp, g_prev, g_out, g, p->data_term_amplifier, p->tikhonov_strength, p->rate,
                           gate, record);
}

// persistent grid of the 3-D iteration kernel: 127 blocks per XCD = 4 per CU, all resident from the first tile to the
// last (85 VGPRs would allow 5).  Measured at 256^3 (tools/hier_kernel_time.py, LSF_BLOCKS_PER_XCD): the time is not
// monotonic in the grid size -- 0.255 / 0.275 / 0.301 / 0.269 / 0.306 ms for 127 / 128 / 160 / 251 / 320 blocks per XCD
// without the in-kernel update: what matters is how the tiles in flight line up with the z +/- 1 slices still in L2,
// and powers of two alias.  End to end 127 against 251: +2..3 % (Tikhonov), +2 % / +6 % (with the kernel, 256^3 / 512^3).
constexpr unsigned kHierBlocksPerXcd3d = 127;

template <int D>
void dispatch_hier(unsigned blocks, hipStream_t s, const float4* packed, const float* canonical, float* warp,
                   const float* g_prev, float* g_out, const Grid& g, const lsf_hier_params* p, const lsf_gate& gate,
                   lsf_iteration_record* record) {
    const bool e = p->compute_energy != 0;
    if (D == 3 && p->previous_max && p->tikhonov_enabled && !p->apply_update && gate.prev_record) {
        if (e)
            hipLaunchKernelGGL((hier_iteration_kernel<3, true, false, true, true>), dim3(blocks), dim3(kBlock), 0, s,
                               packed, canonical, warp, g_prev, g_out, g, p->data_term_amplifier, p->tikhonov_strength,
                               p->rate, gate, record);
        else
            hipLaunchKernelGGL((hier_iteration_kernel<3, true, false, false, true>), dim3(blocks), dim3(kBlock), 0, s,
                               packed, canonical, warp, g_prev, g_out, g, p->data_term_amplifier, p->tikhonov_strength,
                               p->rate, gate, record);
        return;
    }
    if (p->tikhonov_enabled) {
        if (p->apply_update) launch_hier<D, true, true>(e, blocks, s, packed, canonical, warp, g_prev, g_out, g, p, gate, record);
        else launch_hier<D, true, false>(e, blocks, s, packed, canonical, warp, g_prev, g_out, g, p, gate, record);
    } else {
        if (p->apply_update) launch_hier<D, false, true>(e, blocks, s, packed, canonical, warp, g_prev, g_out, g, p, gate, record);
        else launch_hier<D, false, false>(e, blocks, s, packed, canonical, warp, g_prev, g_out, g, p, gate, record);
    }
}

}  // namespace

extern "C" int lsf_hier_iteration(const float* packed_live4, const float* canonical, float* warp_planar,
                                  const float* g_prev_planar, float* g_out_planar, const lsf_grid* grid,
                                  const lsf_hier_params* params, const lsf_gate* gate, lsf_iteration_record* record,
                                  void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!packed_live4 || !canonical || !warp_planar || !params || !record) return LSF_ERR_BAD_ARGUMENT;
    if (params->tikhonov_enabled && (!g_prev_planar || g_prev_planar == g_out_planar)) return LSF_ERR_BAD_ARGUMENT;
    if (!params->apply_update && !g_out_planar) return LSF_ERR_BAD_ARGUMENT;
    if (params->packed_nz < 0 || (params->packed_nz > 0 && grid->dims != 3) ||
        (long long)params->packed_nz * grid->ny * grid->nx > 0x7fffffffll)
        return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    if (params->packed_nz > 0) {
        g.gather_nz = params->packed_nz;
        g.gather_z_offset = params->packed_z_global_offset;
    }
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    const float4* packed = reinterpret_cast<const float4*>(packed_live4);
    lsf_gate gt = gate_or_open(gate);
    if (grid->dims == 2)
        dispatch_hier<2>(launch_blocks(t.total), as_stream(stream), packed, canonical, warp_planar, g_prev_planar, g_out_planar, g,
                         params, gt, record);
    else
        dispatch_hier<3>(launch_blocks(t.total, blocks_per_xcd(kHierBlocksPerXcd3d)), as_stream(stream), packed, canonical, warp_planar, g_prev_planar, g_out_planar, g,
                         params, gt, record);
    return launch_status();
}

extern "C" int lsf_hier_update(const float* g_planar, float* warp_planar, const lsf_grid* grid, float rate,
                               const lsf_gate* gate, lsf_iteration_record* record, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!g_planar || !record) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    lsf_gate gt = gate_or_open(gate);
    const dim3 blocks(launch_blocks(t.total));
    hipStream_t s = as_stream(stream);
    if (grid->dims == 2 && warp_planar)
        hipLaunchKernelGGL((hier_update_kernel<2, true>), blocks, dim3(kBlock), 0, s, g_planar, warp_planar, g, rate, gt, record);
    else if (grid->dims == 2)
        hipLaunchKernelGGL((hier_update_kernel<2, false>), blocks, dim3(kBlock), 0, s, g_planar, warp_planar, g, rate, gt, record);
    else if (warp_planar)
        hipLaunchKernelGGL((hier_update_kernel<3, true>), blocks, dim3(kBlock), 0, s, g_planar, warp_planar, g, rate, gt, record);
    else
        hipLaunchKernelGGL((hier_update_kernel<3, false>), blocks, dim3(kBlock), 0, s, g_planar, warp_planar, g, rate, gt, record);
    return launch_status();
}
