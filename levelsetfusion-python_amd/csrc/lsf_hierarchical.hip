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

// ---- 2-D levels, K iterations per launch: temporal blocking through LDS (round 6) ------------------------------------------
// A 512^2 level is 1 MiB per field and 1024 voxels per CU: one iteration per launch is a ~4.5 us kernel plus a ~1.5 us launch
// boundary whatever the level's size (BASELINE config 2 ran at 7.5 us per iteration from a HIP graph).  An iteration at voxel
// p reads the previous gradient at p and its 4 neighbours, the warp and the canonical value at p, and the STATIC packed live
// field at the data-dependent position p + warp(p) -- so K iterations of a tile T need nothing that K - 1 more rows and
// columns of the dynamic fields around T do not determine.  A workgroup (a tile T of 32 x 32 voxels, or 16 x 16 on small
// levels) loads the warp on T (+) (K - 1) and the previous
// gradient on T (+) K into LDS, runs iteration j on T (+) (K - 1 - j) -- recomputing, bit for bit, what the neighbouring
// workgroups compute for those voxels: the same arithmetic on the same inputs --, and stores its own tile's warp and last
// gradient: no grid-wide barrier (4-5 us on this chip, more than a launch boundary), no hand-off.  Array edges clip the
// regions; the Laplacian's edge rule (a missing neighbour is the centre) then applies exactly where the level ends.  Every
// iteration's maximum is taken over the tile's OWN voxels.  Without the energy printouts.  The stop test
// (hierarchical_optimizer2d.py:169-171: a maximum below the threshold ends the level) is looked at launch by launch: a launch
// whose predecessor holds an iteration that met it -- or that did not run itself -- is a no-op, so the launch in which the
// level converged is the last one to write, its INPUT buffers stay as they were, and the caller repeats it with the number
// of iterations the reference would have run (lsf_hier_level_run_2d).
constexpr int kBlkMaxK = 8, kBlkThreads = 1024;  // (tiles of 32 x 32 or 16 x 16 voxels, see lsf_hier_level_run_2d)

template <int T>
__global__ __launch_bounds__(kBlkThreads) void hier2d_blocked_kernel(const float4* __restrict__ packed,
                                                                     const float* __restrict__ canonical,
                                                                     const float* __restrict__ warp_in,
                                                                     float* __restrict__ warp_out,
                                                                     const float* __restrict__ g_in,
                                                                     float* __restrict__ g_out, Grid g, float amp,
                                                                     float strength, float rate,
                                                                     lsf_iteration_record* records, int k,
                                                                     const lsf_iteration_record* previous,
                                                                     int previous_count, float threshold, int tik) {
    if (previous_count > 0) {
        // the predecessor's records, one (record, slot) word per lane -- one round trip instead of a chain of them; every wave
        // of every workgroup reads the same words and takes the same way
        static_assert(LSF_RECORD_SLOTS * kBlkMaxK == kWave, "one lane per slot of a launch's records");
        const int l = threadIdx.x & (kWave - 1), r = l / LSF_RECORD_SLOTS;
        unsigned long long p = r < previous_count ? previous[r].slot[l % LSF_RECORD_SLOTS].max_packed : ~0ull;
#pragma unroll
        for (int step = 1; step < LSF_RECORD_SLOTS; step <<= 1) {
            const unsigned long long q = __shfl_xor(p, step);
            p = q > p ? q : p;
        }
        const bool met = r < previous_count && (p == 0ull || unpack_max_value(p) < threshold);
        if (__any(met)) return;
    }
    // (row pitch W + 1: a ring's left / right columns are cells a row apart -- at a pitch of 48 or 32 words they would
    // fall into two LDS banks, or one)
    constexpr int H = kBlkMaxK, N = T + 2 * H, W = N + 1;
    constexpr int C = (N * N + kBlkThreads - 1) / kBlkThreads;  // cells per thread: 3 for 32 x 32 tiles, 1 for 16 x 16
    __shared__ float s_w[2][N * W];          // the warp's two components
    __shared__ float s_g[2][2][N * W];       // [buffer][component]: iteration j reads buffer j % 2, writes the other
    __shared__ unsigned long long s_best[kBlkMaxK][kBlkThreads / kWave];
    const int tiles_x = (g.nx + T - 1) / T;
    const int tx0 = (int)(blockIdx.x % tiles_x) * T, ty0 = (int)(blockIdx.x / tiles_x) * T;
    const int tx1 = min(tx0 + T, g.nx), ty1 = min(ty0 + T, g.ny);
    const long long plane = g.plane;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    // A thread owns the SAME cells in every iteration, so everything about a cell that does not change -- its voxel, its
    // canonical value, the Laplacian's clamped neighbour offsets, how many rings outside the tile it lies -- is worked out
    // once per launch.  The cells are numbered RING BY RING: the tile's own T x T voxels first (row-major), then the ring of
    // voxels one step outside it, the next ring, ...; cell number q belongs to thread q mod 1024.  Iteration j works on the
    // cells at most k - 1 - j rings out, i.e. on a PREFIX of that numbering: whole waves work or skip (numbered row by row
    // over the 48 x 48 image, with a wave's 64 cells spread over two rows, an iteration took 3.2 us instead of ~2).
    int cell[C], ring[C], vox[C], nb[C];
    float fx[C], fy[C], cn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int q = (int)threadIdx.x + c * kBlkThreads;
        int u, v, r = 0;  // tile coordinates (0 .. T - 1 inside the tile) and ring
        if (q < T * T) {
            u = q % T;
            v = q / T;
        } else {
            const int h = q - T * T;  // rings 1 .. r hold 4 r (T + r) cells
            r = 1;
            while (r < H && h >= 4 * r * (T + r)) ++r;
            const int p = h - 4 * (r - 1) * (T + r - 1), side = T + 2 * r;  // position on ring r: top, bottom, left, right
            if (p < side) { u = p - r; v = -r; }
            else if (p < 2 * side) { u = p - side - r; v = T - 1 + r; }
            else if (p < 3 * side - 2) { u = -r; v = p - 2 * side + 1 - r; }
            else { u = T - 1 + r; v = p - (3 * side - 2) + 1 - r; }
        }
        const int x = tx0 + u, y = ty0 + v;
        const bool in_array = q < N * N && x >= 0 && x < g.nx && y >= 0 && y < g.ny;
        cell[c] = (v + H) * W + (u + H);
        // rings outside the tile AS CLIPPED BY THE ARRAY (0 = one of the tile's own voxels): a tile at the array's edge is
        // smaller than T x T, and its missing rows / columns count as rings
        const int dx = max(max(tx0 - x, x - (tx1 - 1)), 0), dy = max(max(ty0 - y, y - (ty1 - 1)), 0);
        ring[c] = in_array ? max(dx, dy) : 0x7fff;
        vox[c] = in_array ? y * g.nx + x : 0;
        fx[c] = (float)x;
        fy[c] = (float)y;
        // the Laplacian's neighbours: a missing one is the centre (scipy's mode='nearest')
        nb[c] = (y > 0 ? 1 : 0) | (y < g.ny - 1 ? 2 : 0) | (x > 0 ? 4 : 0) | (x < g.nx - 1 ? 8 : 0);
        cn[c] = 0.0f;
    }
    // the previous gradient k rings out, the warp with it (needed k - 1 rings out only; one load pattern), the canonical
    // values: all of a thread's loads in flight together
    {
        float v[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (ring[c] <= (tik ? k : 0)) {  // (without the Tikhonov term a voxel depends on nothing around it)
                v[c][0] = tik ? g_in[vox[c]] : 0.0f;
                v[c][1] = tik ? g_in[plane + vox[c]] : 0.0f;
                v[c][2] = warp_in[vox[c]];
                v[c][3] = warp_in[plane + vox[c]];
                cn[c] = canonical[vox[c]];
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (ring[c] <= (tik ? k : 0)) {
                s_g[0][0][cell[c]] = v[c][0];
                s_g[0][1][cell[c]] = v[c][1];
                s_w[0][cell[c]] = v[c][2];
                s_w[1][cell[c]] = v[c][3];
            }
        }
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {  // (k is uniform over the grid)
        const int cur = j & 1, nxt = cur ^ 1, reach = tik ? k - 1 - j : 0;
        float w0[C], w1[C];
        Packed smp[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (ring[c] <= reach) {
                w0[c] = s_w[0][cell[c]];
                w1[c] = s_w[1][cell[c]];
                smp[c] = gather_packed<2>(packed, g, fx[c] + w0[c], fy[c] + w1[c], 0.0f);
            }
        }
        unsigned long long best = 0ull;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (ring[c] <= reach) {
                const int l = cell[c];
                const float diff = smp[c].l - cn[c];
                const float live_grad[2] = {smp[c].gx, smp[c].gy};
                const int up = (nb[c] & 1) ? W : 0, down = (nb[c] & 2) ? W : 0, left = (nb[c] & 4) ? 1 : 0,
                          right = (nb[c] & 8) ? 1 : 0;
                float gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const float* a = s_g[cur][d];
                    const float a0 = a[l];
                    const float d2y = second_difference_f64(a[l - up], a0, a[l + down]);
                    const float d2x = second_difference_f64(a[l - left], a0, a[l + right]);
                    const float lap = d2y + d2x;
                    const float gd = diff * live_grad[d];
                    gv[d] = tik ? amp * gd - strength * lap : amp * gd;
                }
                s_g[nxt][0][l] = gv[0];
                s_g[nxt][1][l] = gv[1];
                s_w[0][l] = w0[c] - rate * gv[0];
                s_w[1][l] = w1[c] - rate * gv[1];
                if (ring[c] == 0) {
                    const unsigned long long p = pack_max(vec_length<2>(gv), (unsigned)vox[c] + g.index_offset);
                    best = p > best ? p : best;
                }
            }
        }
        // this iteration's maximum over the tile's own voxels, wave by wave (the workgroup's reduction follows the loop)
        const unsigned long long m = wave_max_u64(best);
        if (lane == 0) s_best[j][wave] = m;
        __syncthreads();
    }
    // the tile's own voxels: the warp after k iterations and the gradient of the last one
    {
        const int fin = k & 1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (ring[c] == 0) {
                g_out[vox[c]] = s_g[fin][0][cell[c]];
                g_out[plane + vox[c]] = s_g[fin][1][cell[c]];
                warp_out[vox[c]] = s_w[0][cell[c]];
                warp_out[plane + vox[c]] = s_w[1][cell[c]];
            }
        }
    }
    // the k maxima: the waves' partial maxima are in LDS (the loop's barriers have published them)
    if (wave == 0) {
        for (int j = 0; j < k; ++j) {
            const unsigned long long mm = wave_max_u64(lane < kBlkThreads / kWave ? s_best[j][lane] : 0ull);
            if (lane == 0 && mm != 0ull) atomicMax(record_max(records + j), mm);
        }
    }
}

// ---- ... and with the gradient kernel (the reference's default constructor, hierarchical_optimizer2d.py:63-73,213-216) -------
// Per iteration and voxel the filter (y pass, then x pass: math_utils/convolution.py:77-83; zero padding, float64 sums in tap
// order, one float32 rounding per pass) widens what an iteration depends on to R + 1 = taps / 2 + 1 voxels: the raw gradient
// needs the previous FILTERED gradient one voxel around (the Laplacian), the y pass the raw gradient R rows up and down, the x
// pass the y pass's results R columns left and right.  So K iterations of a tile need the previous gradient K (R + 1) and the
// warp (K - 1)(R + 1) + R voxels around it: K = 2 for seven taps inside the same 48 x 48 image (8 rings).  Iteration j, with
// r = (K - 1 - j)(R + 1) rings still to be produced for later iterations:
//   G  raw gradient             on the cells at most r + R rings out                      (LDS: s_raw)
//   Y  y pass                   where |dx| <= r + R and |dy| <= r                         (LDS: s_t)
//   X  x pass, update, maximum  on the cells at most r rings out: the filtered gradient replaces the previous one, the warp
//      moves by it; the maximum is taken over the tile's own voxels
// One launch instead of four per iteration (lsf_hier_iteration, two filter passes, lsf_hier_update): the per-iteration path
// runs at 20 us per iteration from a HIP graph.  Cells outside the ARRAY are never written and hold zeros: the filter's zero
// padding.  Same arithmetic on the same inputs as the four kernels: the same bits.
template <int T, int NT, bool FMA>
__global__ __launch_bounds__(kBlkThreads) void hier2d_blocked_filter_kernel(const float4* __restrict__ packed,
                                                                            const float* __restrict__ canonical,
                                                                            const float* __restrict__ warp_in,
                                                                            float* __restrict__ warp_out,
                                                                            const float* __restrict__ g_in,
                                                                            float* __restrict__ g_out, Grid g, float amp,
                                                                            float strength, float rate, TapsN<NT> taps,
                                                                            lsf_iteration_record* records, int k,
                                                                            const lsf_iteration_record* previous,
                                                                            int previous_count, float threshold, int tik) {
    if (previous_count > 0) {  // (as hier2d_blocked_kernel: the predecessor's records, one word per lane)
        const int l = threadIdx.x & (kWave - 1), r = l / LSF_RECORD_SLOTS;
        unsigned long long p = r < previous_count ? previous[r].slot[l % LSF_RECORD_SLOTS].max_packed : ~0ull;
#pragma unroll
        for (int step = 1; step < LSF_RECORD_SLOTS; step <<= 1) {
            const unsigned long long q = __shfl_xor(p, step);
            p = q > p ? q : p;
        }
        const bool met = r < previous_count && (p == 0ull || unpack_max_value(p) < threshold);
        if (__any(met)) return;
    }
    constexpr int R = NT / 2;      // the filter's reach
    const int S = tik ? R + 1 : R;  // rings an iteration consumes (the Laplacian of the previous gradient: one more)
    constexpr int H = kBlkMaxK, N = T + 2 * H, W = N + 1;
    constexpr int C = (N * N + kBlkThreads - 1) / kBlkThreads;
    __shared__ float s_w[2][N * W];    // the warp's two components
    __shared__ float s_gp[2][N * W];   // the previous (filtered) gradient; the x pass writes the next one over it
    __shared__ float s_raw[2][N * W];  // the raw gradient
    __shared__ float s_t[2][N * W];    // the y pass's results
    __shared__ unsigned long long s_best[kBlkMaxK][kBlkThreads / kWave];
    const int tiles_x = (g.nx + T - 1) / T;
    const int tx0 = (int)(blockIdx.x % tiles_x) * T, ty0 = (int)(blockIdx.x / tiles_x) * T;
    const int tx1 = min(tx0 + T, g.nx), ty1 = min(ty0 + T, g.ny);
    const long long plane = g.plane;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int i = threadIdx.x; i < N * W; i += kBlkThreads) {  // zeros wherever nothing is ever computed: the zero padding
        s_raw[0][i] = 0.0f; s_raw[1][i] = 0.0f;
        s_t[0][i] = 0.0f; s_t[1][i] = 0.0f;
    }
    // the cells of a thread, numbered ring by ring (hier2d_blocked_kernel), with their distance from the tile along each axis
    int cell[C], rx[C], ry[C], vox[C], nb[C];
    float fx[C], fy[C], cn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int q = (int)threadIdx.x + c * kBlkThreads;
        int u, v, r = 0;
        if (q < T * T) {
            u = q % T;
            v = q / T;
        } else {
            const int h = q - T * T;
            r = 1;
            while (r < H && h >= 4 * r * (T + r)) ++r;
            const int p = h - 4 * (r - 1) * (T + r - 1), side = T + 2 * r;
            if (p < side) { u = p - r; v = -r; }
            else if (p < 2 * side) { u = p - side - r; v = T - 1 + r; }
            else if (p < 3 * side - 2) { u = -r; v = p - 2 * side + 1 - r; }
            else { u = T - 1 + r; v = p - (3 * side - 2) + 1 - r; }
        }
        const int x = tx0 + u, y = ty0 + v;
        const bool in_array = q < N * N && x >= 0 && x < g.nx && y >= 0 && y < g.ny;
        cell[c] = (v + H) * W + (u + H);
        rx[c] = in_array ? max(max(tx0 - x, x - (tx1 - 1)), 0) : 0x7fff;
        ry[c] = in_array ? max(max(ty0 - y, y - (ty1 - 1)), 0) : 0x7fff;
        vox[c] = in_array ? y * g.nx + x : 0;
        fx[c] = (float)x;
        fy[c] = (float)y;
        nb[c] = (y > 0 ? 1 : 0) | (y < g.ny - 1 ? 2 : 0) | (x > 0 ? 4 : 0) | (x < g.nx - 1 ? 8 : 0);
        cn[c] = 0.0f;
    }
    {
        float v[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (max(rx[c], ry[c]) <= k * S) {
                v[c][0] = tik ? g_in[vox[c]] : 0.0f;
                v[c][1] = tik ? g_in[plane + vox[c]] : 0.0f;
                v[c][2] = warp_in[vox[c]];
                v[c][3] = warp_in[plane + vox[c]];
                cn[c] = canonical[vox[c]];
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (max(rx[c], ry[c]) <= k * S) {
                s_gp[0][cell[c]] = v[c][0];
                s_gp[1][cell[c]] = v[c][1];
                s_w[0][cell[c]] = v[c][2];
                s_w[1][cell[c]] = v[c][3];
            }
        }
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        const int r = (k - 1 - j) * S;
        // G: the raw gradient
        {
            float w0[C], w1[C];
            Packed smp[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (max(rx[c], ry[c]) <= r + R) {
                    w0[c] = s_w[0][cell[c]];
                    w1[c] = s_w[1][cell[c]];
                    smp[c] = gather_packed<2>(packed, g, fx[c] + w0[c], fy[c] + w1[c], 0.0f);
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (max(rx[c], ry[c]) <= r + R) {
                    const int l = cell[c];
                    const float diff = smp[c].l - cn[c];
                    const float live_grad[2] = {smp[c].gx, smp[c].gy};
                    const int up = (nb[c] & 1) ? W : 0, down = (nb[c] & 2) ? W : 0, left = (nb[c] & 4) ? 1 : 0,
                              right = (nb[c] & 8) ? 1 : 0;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const float* a = s_gp[d];
                        const float a0 = a[l];
                        const float d2y = second_difference_f64(a[l - up], a0, a[l + down]);
                        const float d2x = second_difference_f64(a[l - left], a0, a[l + right]);
                        const float lap = d2y + d2x;
                        const float gd = diff * live_grad[d];
                        s_raw[d][l] = tik ? amp * gd - strength * lap : amp * gd;
                    }
                }
            }
        }
        __syncthreads();
        // Y: out[y] = sum_j k[j] * in[y + R - j]
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (rx[c] <= r + R && ry[c] <= r) {
                const int l = cell[c];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    double acc = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc = mac<FMA>(acc, taps.k[t], (double)s_raw[d][l + (R - t) * W]);
                    s_t[d][l] = (float)acc;
                }
            }
        }
        __syncthreads();
        // X, the update, the maximum
        unsigned long long best = 0ull;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (max(rx[c], ry[c]) <= r) {
                const int l = cell[c];
                float gv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    double acc = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc = mac<FMA>(acc, taps.k[t], (double)s_t[d][l + (R - t)]);
                    gv[d] = (float)acc;
                    s_gp[d][l] = gv[d];
                    s_w[d][l] = s_w[d][l] - rate * gv[d];
                }
                if (max(rx[c], ry[c]) == 0) {
                    const unsigned long long p = pack_max(vec_length<2>(gv), (unsigned)vox[c] + g.index_offset);
                    best = p > best ? p : best;
                }
            }
        }
        const unsigned long long m = wave_max_u64(best);
        if (lane == 0) s_best[j][wave] = m;
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (max(rx[c], ry[c]) == 0) {
            g_out[vox[c]] = s_gp[0][cell[c]];
            g_out[plane + vox[c]] = s_gp[1][cell[c]];
            warp_out[vox[c]] = s_w[0][cell[c]];
            warp_out[plane + vox[c]] = s_w[1][cell[c]];
        }
    }
    if (wave == 0) {
        for (int j = 0; j < k; ++j) {
            const unsigned long long mm = wave_max_u64(lane < kBlkThreads / kWave ? s_best[j][lane] : 0ull);
            if (lane == 0 && mm != 0ull) atomicMax(record_max(records + j), mm);
        }
    }
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

template <int NT>
static int launch_blocked_filter(bool small, bool fma, unsigned tiles, hipStream_t s, const float4* packed,
                                 const float* canonical, const float* w_in, float* w_out, const float* g_in, float* g_out,
                                 const Grid& g, const lsf_hier_params* params, const double* taps_host,
                                 lsf_iteration_record* records, int k, const lsf_iteration_record* previous,
                                 int previous_count, float threshold) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
#define LSF_LAUNCH_BLOCKED(TILE, FMA)                                                                                   \
    hipLaunchKernelGGL((hier2d_blocked_filter_kernel<TILE, NT, FMA>), dim3(tiles), dim3(kBlkThreads), 0, s, packed,       \
                       canonical, w_in, w_out, g_in, g_out, g, params->data_term_amplifier, params->tikhonov_strength,   \
                       params->rate, taps, records, k, previous, previous_count, threshold, params->tikhonov_enabled)
    if (small && fma) LSF_LAUNCH_BLOCKED(16, true);
    else if (small) LSF_LAUNCH_BLOCKED(16, false);
    else if (fma) LSF_LAUNCH_BLOCKED(32, true);
    else LSF_LAUNCH_BLOCKED(32, false);
#undef LSF_LAUNCH_BLOCKED
    return launch_status();
}

extern "C" int lsf_hier_level_run_2d(const float* packed_live4, const float* canonical, float* warp_a, float* warp_b,
                                     float* g_a, float* g_b, const lsf_grid* grid, const lsf_hier_params* params,
                                     const double* taps_host, int32_t n_taps, lsf_iteration_record* records,
                                     int32_t iterations, int32_t iterations_per_launch, float threshold, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!packed_live4 || !canonical || !warp_a || !warp_b || warp_a == warp_b || !g_a || !g_b || g_a == g_b || !params ||
        !records || iterations < 0 || iterations_per_launch < 1 || iterations_per_launch > kBlkMaxK || n_taps < 0 ||
        (n_taps > 0 && !taps_host))
        return LSF_ERR_BAD_ARGUMENT;
    // 2-D, Tikhonov term, no energy sums, the level's own packed field; the update applied in the launch -- by the iteration
    // itself without a gradient kernel (apply_update), behind the filter with one (apply_update off, as lsf_hier_iteration
    // is called then)
    const bool filtered = n_taps > 0;
    if (grid->dims != 2 || (params->apply_update != 0) == filtered || params->compute_energy || params->previous_max ||
        params->packed_nz != 0)
        return LSF_ERR_BAD_DIMS;
    if (filtered && n_taps != 3 && n_taps != 5 && n_taps != 7 && n_taps != 9) return LSF_ERR_BAD_DIMS;
    // the rings an iteration consumes (the Tikhonov term's Laplacian: 1; a filter: taps / 2) times the iterations of a launch:
    // inside the image
    const int tik = params->tikhonov_enabled ? 1 : 0;
    if (iterations_per_launch * ((filtered ? n_taps / 2 : 0) + tik) > kBlkMaxK) return LSF_ERR_BAD_ARGUMENT;
    const Grid g = make_grid(grid);
    // 32 x 32 tiles keep the recomputed rings cheapest (1.5 x the level's voxels over a launch of eight); levels too small
    // to give every CU such a tile take 16 x 16 tiles: more workgroups, each with a third of the cells
    auto tiles_of = [&](int t) { return (unsigned)((grid->nx + t - 1) / t) * (unsigned)((grid->ny + t - 1) / t); };
    const bool small = tiles_of(32) < 192u;
    const unsigned tiles = tiles_of(small ? 16 : 32);
    const float4* packed = reinterpret_cast<const float4*>(packed_live4);
    const bool fma = filtered && taps_are_float32(taps_host, n_taps);
    hipStream_t s = as_stream(stream);
    int launch = 0;
    for (int32_t i = 0; i < iterations; i += iterations_per_launch, ++launch) {
        const int k = iterations - i < iterations_per_launch ? iterations - i : iterations_per_launch;
        const bool even = (launch & 1) == 0;  // launch b reads (warp_a, g_a) when b is even and writes the other pair
        // a stop test that can fire: the launch looks at its predecessor's records first
        const lsf_iteration_record* previous = threshold > 0.0f && launch > 0 ? records + i - iterations_per_launch : nullptr;
        const int previous_count = previous ? iterations_per_launch : 0;
        const float* w_in = even ? warp_a : warp_b;
        float* w_out = even ? warp_b : warp_a;
        const float* g_in = even ? g_a : g_b;
        float* g_out = even ? g_b : g_a;
        if (filtered) {
            int e = 0;
            switch (n_taps) {
                case 3: e = launch_blocked_filter<3>(small, fma, tiles, s, packed, canonical, w_in, w_out, g_in, g_out, g, params, taps_host, records + i, k, previous, previous_count, threshold); break;
                case 5: e = launch_blocked_filter<5>(small, fma, tiles, s, packed, canonical, w_in, w_out, g_in, g_out, g, params, taps_host, records + i, k, previous, previous_count, threshold); break;
                case 7: e = launch_blocked_filter<7>(small, fma, tiles, s, packed, canonical, w_in, w_out, g_in, g_out, g, params, taps_host, records + i, k, previous, previous_count, threshold); break;
                default: e = launch_blocked_filter<9>(small, fma, tiles, s, packed, canonical, w_in, w_out, g_in, g_out, g, params, taps_host, records + i, k, previous, previous_count, threshold); break;
            }
            if (e) return e;
            continue;
        }
        if (small)
            hipLaunchKernelGGL(hier2d_blocked_kernel<16>, dim3(tiles), dim3(kBlkThreads), 0, s, packed, canonical, w_in, w_out,
                               g_in, g_out, g, params->data_term_amplifier, params->tikhonov_strength, params->rate,
                               records + i, k, previous, previous_count, threshold, params->tikhonov_enabled);
        else
            hipLaunchKernelGGL(hier2d_blocked_kernel<32>, dim3(tiles), dim3(kBlkThreads), 0, s, packed, canonical, w_in, w_out,
                               g_in, g_out, g, params->data_term_amplifier, params->tikhonov_strength, params->rate,
                               records + i, k, previous, previous_count, threshold, params->tikhonov_enabled);
        if (int e = launch_status()) return e;
    }
    return 0;
}
