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
#include "lsf_slavcheva_state_taps.h"

using namespace lsf;
using namespace lsf::slav;

namespace {

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

// WALK: the dense tile walk over every voxel, a band list (ALL or BOUNDARY subset), an INTERIOR band list -- and the two
// list walks once more for slabs cut along y (lsf_grid::y_global_offset / ny_global / energy_y_*).  Whole volumes and
// z-slabs run instantiations that know those fields to be trivial: carried as live scalars they cost the INTERIOR walk
// 19 VALU instructions per 64 voxels, six more scalar spills into vector lanes and a 16-byte scratch frame (ISA of round
// 3's kernel against round 4's with the y fields added).
constexpr int kWalkDense = 0, kWalkList = 1, kWalkListInterior = 2, kWalkListYCut = 3, kWalkListInteriorYCut = 4;
constexpr int walk_of(int walk_code) {
    return walk_code == kWalkListYCut ? kWalkList : (walk_code == kWalkListInteriorYCut ? kWalkListInterior : walk_code);
}

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY, int WALK_CODE>
__global__ __launch_bounds__(walk_of(WALK_CODE) == kWalkListInterior ? kCuBlock : kBlock)
// <= 128 VGPRs for the dense walk (at 132 = 3 waves per SIMD it streamed 20 % slower) and the CU-sized workgroups; the
// general list walk (voxels on a face, arrays of 4 GiB and more) may take 3 waves' worth instead of spilling -- a list
// walk runs as fast with 3 waves per SIMD as with 4 (measured with 768-thread workgroups)
__attribute__((amdgpu_waves_per_eu(walk_of(WALK_CODE) == kWalkList ? 3 : 4, 4)))
void slavcheva_state_kernel(const vf4* __restrict__ state_in,
                                                                 const float* __restrict__ canonical,
                                                                 vf4* __restrict__ state_out, Grid g, Params p,
                                                                 lsf_gate gate, lsf_iteration_record* record,
                                                                 const int* __restrict__ band_list,
                                                                 unsigned band_count) {
    constexpr int WALK = walk_of(WALK_CODE);
    if (WALK_CODE == WALK) {  // not cut along y: every row is a row of the volume, every row's energies count
        g.y_global_offset = 0;
        g.ny_global = g.ny;
        g.ey_begin = -0x7fffffff - 1;
        g.ey_end = 0x7fffffff;
        g.y_cut = 0;
    }
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
            if (z >= g.e_begin && z < g.e_end && y >= g.ey_begin && y < g.ey_end) {  // a slab's recomputed halo slices / rows belong to the neighbour's sums
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
            const float px = (float)x + d.wv[0], py = (float)(y + g.y_global_offset) + d.wv[1];
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
        const bool counted = in_band && z >= g.e_begin && z < g.e_end && y >= g.ey_begin && y < g.ey_end;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            d.wv[c] = (c < D && in_band) ? (-gv[c]) * p.rate : 0.0f;
            en[c] += counted ? e[c] : 0.0;
        }
        const float len = vec_length<D>(d.wv);
        const bool moved = !(d.wv[0] == 0.0f && d.wv[1] == 0.0f && d.wv[2] == 0.0f);
        if (!rewarp_from_taps<D, true>(n, state_in, g, (int)i, x, y, z, d.wv, d.rw)) {
            d.rw.lerp = false;
            d.rw.R = state_gather<D>(state_in, g, (float)x + d.wv[0], (float)(y + g.y_global_offset) + d.wv[1],
                                     D == 3 ? (float)(z + g.z_global_offset) + d.wv[2] : 0.0f);
        }
        if (!moved) {  // zero displacement: the gather returns live[p] bit for bit (every lerp is a*1 + b*0)
            d.rw.lerp = false;
            d.rw.R = l;
        }
        // i + index_offset = linear_index(g, x, y, z) unless the slab is cut along y (wave-uniform)
        const unsigned long long q = listed ? pack_max(len, g.y_cut ? linear_index(g, x, y, z) : i + g.index_offset) : 0ull;
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
                                                                     double* __restrict__ scratch,
                                                                     const int* __restrict__ skip_flag,
                                                                     const lsf_iteration_record* __restrict__ guard_records,
                                                                     unsigned guard_count, float guard_limit) {
    // the chain launch in front of this pass found an update its dependency windows do not cover: the state is not
    // the reference's, the caller's fields stay as they are (lsf_slavcheva_state_chain)
    if (skip_flag && *skip_flag != 0) return;
    // states initialised near the band only (lsf_state_pack_needed): an executed iteration whose maximum update is not
    // below the reach may have gathered uninitialised words -- every block looks at the records itself (a few hundred
    // loads, all blocks at once) instead of a one-block kernel in front of the pass
    if (guard_records) {
        bool bad = false;
        for (unsigned k = threadIdx.x; k < guard_count * LSF_RECORD_SLOTS; k += kBlock) {
            const unsigned long long p = guard_records[k / LSF_RECORD_SLOTS].slot[k % LSF_RECORD_SLOTS].max_packed;
            if (p != 0ull) bad |= !(unpack_max_value(p) < guard_limit);
        }
        if (__syncthreads_or(bad)) return;
    }
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
    if constexpr (D == 3 && (WALK == kWalkList || WALK == kWalkListInterior)) {
        if (a.g.y_cut) {
            hipLaunchKernelGGL((slavcheva_state_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY,
                                                       WALK == kWalkList ? kWalkListYCut : kWalkListInteriorYCut>),
                               dim3(a.blocks), dim3(a.threads), 0, a.s, a.state_in, a.canonical, a.state_out, a.g, a.p, a.gate,
                               a.record, a.band_list, a.band_count);
            return;
        }
    }
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
    if (int e = check_grid(grid, true)) return e;
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
    if (int e = check_grid(grid, true)) return e;
    if (!state) return LSF_ERR_BAD_ARGUMENT;
    long long first, n;
    if (!range_of(grid, first, n)) return 0;
    hipLaunchKernelGGL(state_unpack_kernel, dim3(stream_blocks(n)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const vf4*>(state), live_out, warp_planar_out, warp_interleaved_out, first, n,
                       (long long)grid->nz * grid->ny * grid->nx, grid->dims);
    return launch_status();
}

extern "C" int64_t lsf_state_finalize_scratch_elements(const lsf_grid* grid) {
    if (check_grid(grid, true)) return 0;
    long long first, n;
    range_of(grid, first, n);
    return (int64_t)finalize_blocks(n > 0 ? n : 1) * kFinalizeWords;
}

extern "C" int lsf_state_finalize(const float* state, const float* canonical, float* live_out,
                                  float* warp_planar_out, float* warp_interleaved_out, const lsf_grid* grid,
                                  float lower_threshold, double* statistics16, double* scratch, void* stream) {
    if (int e = check_grid(grid, true)) return e;
    if (!state || (statistics16 && (!canonical || !scratch))) return LSF_ERR_BAD_ARGUMENT;
    if (statistics16 && grid_is_y_cut(grid)) return LSF_ERR_BAD_ARGUMENT;  // the statistics report z-cut indices only
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
                                         double* statistics16, double* scratch, const int32_t* skip_flag,
                                         const lsf_iteration_record* guard_records, int32_t guard_count, float guard_limit,
                                         void* stream) {
    if (int e = check_grid(grid, true)) return e;
    if (statistics16 && grid_is_y_cut(grid)) return LSF_ERR_BAD_ARGUMENT;  // the statistics report z-cut indices only
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
                           lower_threshold, statistics16 ? scratch + (long long)rows * kFinalizeWords : (double*)nullptr,
                           skip_flag, guard_records, (unsigned)(guard_records && guard_count > 0 ? guard_count : 0), guard_limit);
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
    if (int e = check_grid(grid, /*allow_y_cut=*/true)) return e;
    if (!state_in || !canonical || !state_out || state_out == state_in || !params || !record)
        return LSF_ERR_BAD_ARGUMENT;
    if (grid_is_y_cut(grid) && (!band_list || grid->dims != 3)) return LSF_ERR_BAD_ARGUMENT;  // y-cut slabs: 3-D band lists
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
    // waves per CU of the INTERIOR list walk (2, 3 or 4 per SIMD run equally fast: 33.0 / 31.6 / 31.5 us) and wave-units per
    // group (1 ... 16 within 3 %, 32 is 8 % slower): measured in round 2; variant builds
    // (tools/build_variant.sh NAME - -DLSF_LIST_THREADS=n -DLSF_LIST_GROUP=n) override them for measurements
#ifdef LSF_LIST_THREADS
    const unsigned list_threads = LSF_LIST_THREADS;
#else
    const unsigned list_threads = kCuBlock;
#endif
#ifdef LSF_LIST_GROUP
    g.list_group = LSF_LIST_GROUP;
#endif
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
