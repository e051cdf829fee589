// K fused warp-update iterations of an INTERIOR band list in ONE launch: the "chain" kernel.
// Reference loop: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330 executed :360-362 times with nothing in
// between but the stop test -- which cannot fire in a fixed-count run (min_iterations == max_iterations).
//
// Why: a launch of the per-iteration kernel (lsf_slavcheva_state.hip) costs ~9 us that do not shrink with the kernel --
// launch gap, prologue, the drain of the slowest CU, epilogue -- against ~22 us of throughput at 256^3.  Here one
// CU-sized workgroup per CU stays resident for all K iterations and nothing ever waits for the whole chip:
//   * the list is cut into contiguous CHUNKS (a chunk of a sorted list is a thin z-range);
//   * work item (chunk c, iteration i) needs the items (d, i - 1) of the chunks d whose voxels lie within the stencil's and
//     the re-warp gather's reach of chunk c -- a WINDOW of a few chunks either side, computed once per list
//     (chain_plan_kernel).  The same condition covers the write-after-read hazard of the two ping-pong states: (d, i - 1)
//     done means d has finished READING the buffer (c, i) writes;
//   * a workgroup publishes progress[c] = i + 1 after its stores have left (write-through, below) and polls the window's
//     progress words before it starts an item.  A CU that is ahead waits for its neighbours only; the chip never drains.
// Stages: with S > 1 the CUs are split into S groups and group s runs the iterations i = s (mod S) over ALL chunks, dealt
// round-robin inside the group: iteration i + 1 follows iteration i through the list at a distance of one window, so the
// state a stage reads was written moments ago by the stage before (Infinity Cache instead of HBM when the two states do
// not fit the cache next to each other: 512^3).  Every CU walks its items in increasing iteration order, so the
// dependency graph has no cycle through a waiting workgroup: no deadlock as long as all workgroups are resident (one
// 1024-thread workgroup per CU; checked by the host against the occupancy query).  Every wait is bounded
// (ChainPlan::timeout_ticks): on a timeout the launch poisons its last record and every workgroup leaves.
//
// Visibility inside a launch (MI355X_MICROARCH.md "inter-workgroup visibility", cdna_hip_programming.md Guideline 16,
// recipe R1): per-XCD L2s are not coherent and a CU's L1 is never refreshed by other CUs' stores.  Producer: every
// state store is a 16-byte WRITE-THROUGH store (sc1), every storing wave drains (s_waitcnt vmcnt(0)) before the
// workgroup barrier, then ONE lane stores the progress word (sc1).  Consumer: ONE wave polls the window's words with
// relaxed agent-scope loads (sc1), then ONE agent-scope acquire (buffer_inv sc1: this CU's L1), its wait, the workgroup
// barrier, then plain loads.  canonical and the list are never written during a launch.
//
// Results are those of K launches of the per-iteration kernel, bit for bit (same per-voxel code: lsf_slavcheva_state_taps.h),
// provided no update reaches beyond the windows: |warp update| < 2 voxels.  The kernel raises a flag otherwise
// (control word 1) which the finalize pass honours (it leaves the caller's fields alone) and the host sees in the records'
// maxima: the caller then repeats the call with per-iteration launches.
#include "lsf_slavcheva_state_taps.h"
#include "lsf_hip_chain.h"

using namespace lsf;
using namespace lsf::slav;

namespace {

typedef unsigned __attribute__((address_space(1))) gu32;  // agent-scope accesses go through global, never flat

constexpr unsigned kControlWords = 16;      // [0] abort (a wait timed out), [1] reach violated; then the progress words
constexpr unsigned kMaxChunks = 4096;       // chain_plan_kernel keeps two words per chunk in LDS
constexpr unsigned kStageChunkUnits = 96;   // target chunk size when chunks are dealt round-robin (S > 1)
constexpr float kReachLimit = 2.0f;         // windows cover stencil + gather for update lengths below this
constexpr int kReachSlices = 2;

struct ChainShape {
    unsigned workgroups, stages, members, chunks, units, progress_words, threads;
};

// Threads per workgroup (LSF_CHAIN_THREADS: 1024 / 512 / 256; measurement knob).  Smaller workgroups -- two or four
// independent ones per CU, so that one walks while another drains, reduces, publishes and waits -- measured SLOWER
// (256^3: 31.2 / 32.4 / 38.7 us per iteration with 1 / 2 / 4 workgroups per CU): the SIMDs issue oldest-wave-first, the
// younger workgroup of a CU takes 32 us for what the older does in 20, and everybody's window ends up waiting for it.
inline unsigned chain_threads() {
    static const unsigned v = [] {
        const char* e = getenv("LSF_CHAIN_THREADS");
        const int n = e ? atoi(e) : 0;
        return (n == 1024 || n == 512 || n == 256) ? (unsigned)n : (unsigned)kCuBlock;
    }();
    return v;
}

// One policy for the three entry points (scratch size, plan, launch): how a list of `count` entries is cut and dealt.
inline ChainShape chain_shape(long long count, int stages_wanted, unsigned cus) {
    ChainShape s;
    s.threads = chain_threads();
    s.units = (unsigned)((count + kWave - 1) / kWave);
    const unsigned slots = cus * (kCuBlock / s.threads);  // resident workgroups of the whole device
    unsigned wg = s.units / (s.threads / kWave);          // a wave-unit per wave at least
    if (wg < 1) wg = 1;
    if (wg > slots) wg = slots;
    if (wg > kXcds) wg -= wg % kXcds;
    s.workgroups = wg;
    s.stages = 1;
    if (stages_wanted > 1 && wg == slots && slots % (unsigned)stages_wanted == 0 &&
        s.units / kStageChunkUnits >= 2u * slots)  // a sweep through the list is longer than the pipeline of stages
        s.stages = (unsigned)stages_wanted;
    s.members = wg / s.stages;
    if (s.stages == 1) {
        s.chunks = wg;
    } else {
        unsigned want = (s.units + kStageChunkUnits - 1) / kStageChunkUnits;
        want = ((want + s.members - 1) / s.members) * s.members;
        while (want > kMaxChunks) want -= s.members;
        s.chunks = want;
    }
    s.progress_words = (s.chunks + 15u) & ~15u;
    return s;
}

#ifdef LSF_CHAIN_TRACE  // measurement builds only (tools/chain_trace.py): 100 MHz stamps per work item
__device__ unsigned long long* g_chain_trace = nullptr;  // [workgroup][item < kTraceItems][kTraceWords]
constexpr unsigned kTraceItems = 64, kTraceWords = 32;
#define LSF_CT(slot)                                                                                \
    do {                                                                                            \
        if (trace_row && threadIdx.x == 0) trace_row[slot] = __builtin_amdgcn_s_memrealtime();       \
    } while (0)
#else
#define LSF_CT(slot) do {} while (0)
#endif

struct ChainPlan {
    unsigned iterations, chunks, units, stages, members, progress_words, timeout_ticks;
    float reach_limit;
};

__device__ inline unsigned chunk_begin(unsigned c, unsigned base, unsigned extra) { return c * base + (c < extra ? c : extra); }

// windows[2 c], windows[2 c + 1] = first and last chunk whose voxels lie within `reach` voxel indices of chunk c's
// (the list is ascending, so a chunk is an index interval).  One workgroup; C <= kMaxChunks.
__global__ __launch_bounds__(kCuBlock) void chain_plan_kernel(const int* __restrict__ list, unsigned count, unsigned units,
                                                              unsigned chunks, long long reach, unsigned* __restrict__ windows) {
    __shared__ int s_first[kMaxChunks], s_last[kMaxChunks];
    const unsigned base = units / chunks, extra = units % chunks;
    for (unsigned c = threadIdx.x; c < chunks; c += blockDim.x) {
        const unsigned b = chunk_begin(c, base, extra), e = chunk_begin(c + 1, base, extra);
        const unsigned long long last = (unsigned long long)e * kWave;
        s_first[c] = list[(unsigned long long)b * kWave];
        s_last[c] = list[(last < count ? last : count) - 1];
    }
    __syncthreads();
    for (unsigned c = threadIdx.x; c < chunks; c += blockDim.x) {
        unsigned lo = c, hi = c;
        while (lo > 0 && (long long)s_last[lo - 1] + reach >= (long long)s_first[c]) --lo;
        while (hi + 1 < chunks && (long long)s_first[hi + 1] - reach <= (long long)s_last[c]) ++hi;
        windows[2 * c] = lo;
        windows[2 * c + 1] = hi;
    }
}

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
__global__ __launch_bounds__(kCuBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
void slavcheva_state_chain_kernel(vf4* state_a, vf4* state_b, const float* __restrict__ canonical, Grid g, Params p,
                                  lsf_iteration_record* records, const int* __restrict__ band_list,
                                  unsigned band_count, ChainPlan plan, unsigned* scratch) {
    __shared__ unsigned s_next_unit;
    __shared__ int s_abort;
    __shared__ unsigned long long s_max[kMaxBlockWaves];
    __shared__ double s_sum[3][kMaxBlockWaves];
    const unsigned waves = blockDim.x / kWave, wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const unsigned nb = gridDim.x, bid = blockIdx.x;
    // workgroups b, b + 8, ... share an XCD under round-robin dispatch (speed only): ranks are XCD-major, so that the
    // chunks of neighbouring ranks -- neighbouring z-ranges -- meet in one L2
    const unsigned rank = nb % kXcds == 0 ? (bid % kXcds) * (nb / kXcds) + bid / kXcds : bid;
    const unsigned stage = rank / plan.members, member = rank - stage * plan.members;
    gu32* control = (gu32*)scratch;
    gu32* progress = (gu32*)(scratch + kControlWords);
    const unsigned* windows = scratch + kControlWords + plan.progress_words;
    const unsigned base = plan.units / plan.chunks, extra = plan.units % plan.chunks;
    __shared__ unsigned s_worst_bits;  // the longest update this workgroup has produced (float bits, non-negative)
    if (threadIdx.x == 0) {
        s_abort = 0;
        s_worst_bits = 0u;
    }

    for (unsigned it = stage; it < plan.iterations; it += plan.stages) {
        const bool odd = it & 1u;
        const vf4* __restrict__ state_in = odd ? state_b : state_a;
        const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(odd ? state_a : state_b, 0, -1, 0x00020000);
        lsf_iteration_record* record = records + it;
        for (unsigned c = member; c < plan.chunks; c += plan.members) {
            const unsigned u_begin = chunk_begin(c, base, extra), u_end = chunk_begin(c + 1, base, extra);
#ifdef LSF_CHAIN_TRACE
            unsigned long long* trace_row = nullptr;
            {
                const unsigned item = (it / plan.stages) * ((plan.chunks + plan.members - 1) / plan.members) + c / plan.members;
                if (g_chain_trace && item < kTraceItems)
                    trace_row = g_chain_trace + ((unsigned long long)blockIdx.x * kTraceItems + item) * kTraceWords;
                if (trace_row && threadIdx.x == 0) {
                    trace_row[7] = ((unsigned long long)it << 32) | c;
                    trace_row[8] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
                }
            }
#endif
            LSF_CT(0);
            // ---- wait until the window's chunks have finished iteration it - 1 (ONE wave polls, relaxed, then ONE acquire)
            if (threadIdx.x == 0) s_next_unit = u_begin + 2u * waves;
            if (wave == 0 && it > 0) {
                const unsigned lo = windows[2 * c], hi = windows[2 * c + 1];
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    bool ok = true;
                    for (unsigned d = lo + lane; d <= hi; d += kWave)
                        ok &= __hip_atomic_load(progress + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= it;
                    if (__all(ok)) break;
#ifdef LSF_CHAIN_TRACE
                    if (trace_row && threadIdx.x == 0) trace_row[9] += 1;  // polls that did not match
#endif
                    const unsigned gone = __hip_atomic_load(control, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (gone != 0u || __builtin_amdgcn_s_memrealtime() - t0 > plan.timeout_ticks) {
                        if (lane == 0) {
                            s_abort = 1;
                            __hip_atomic_store(control, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            // an aborted launch's state is half written: the finalize pass behind it (skip_flag = control
                            // word 1) must leave the caller's fields alone exactly as after a reach violation
                            __hip_atomic_store(control + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            // the host reads records, not this scratch: an all-ones maximum decodes as NaN
                            atomicMax(reinterpret_cast<unsigned long long*>(&records[plan.iterations - 1].slot[0].max_packed),
                                      ~0ull);
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                LSF_CT(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LSF_CT(2);
            }
            __syncthreads();
            if (s_abort) return;
            LSF_CT(3);

            // ---- the chunk's wave-units: the software-pipelined INTERIOR walk of lsf_slavcheva_state.hip
            unsigned long long best = 0ull;
            double en[3] = {0.0, 0.0, 0.0};
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
                // write-through (aux 16 = sc1): the bytes are on their way to memory when the wave's vmcnt drains
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rsrc_out, (int)((unsigned)d.i * 16u), 0, 16);
            };
            auto interior_voxel = [&](unsigned i, const vf4& sc, float cn, bool listed, const Deferred& previous) {
                NbhStateFast<D> n;
                n.load(state_in, g, i, sc, false);
                __builtin_amdgcn_sched_barrier(0);
                finish(previous);
                __builtin_amdgcn_sched_barrier(0);
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
                for (int k = 0; k < 3; ++k) {
                    d.wv[k] = (k < D && in_band) ? (-gv[k]) * p.rate : 0.0f;
                    en[k] += counted ? e[k] : 0.0;
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
                const unsigned long long q = listed ? pack_max(len, i + g.index_offset) : 0ull;
                best = q > best ? q : best;
                return d;
            };
            auto entry = [&](unsigned unit, bool& listed) {
                const unsigned k = unit * kWave + lane;
                listed = unit < u_end && k < band_count;
                return (unsigned)band_list[k < band_count ? k : band_count - 1u];
            };
            auto grab = [&]() {
                unsigned v = 0u;
                if (lane == 0) v = atomicAdd(&s_next_unit, 1u);
                return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            };
            Deferred pending;
            pending.i = -1;
            unsigned u = u_begin + wave, u1 = u + waves, u2 = 0u;
            bool in0 = false, in1 = false, in2 = false;
            unsigned i0 = 0u, i1 = 0u;
            vf4 s0 = {0.0f, 0.0f, 0.0f, 0.0f};
            float c0 = 0.0f;
            if (u < u_end) {
                i0 = entry(u, in0);
                i1 = entry(u1, in1);
                s0 = state_in[i0];
                c0 = canonical[i0];
            }
            while (u < u_end) {
                u2 = grab();
                const vf4 s1 = state_in[i1];
                const float c1 = canonical[i1];
                const unsigned i2 = entry(u2, in2);
                pending = interior_voxel(i0, s0, c0, in0, pending);
                u = u1;
                u1 = u2;
                i0 = i1; in0 = in1; s0 = s1; c0 = c1;
                i1 = i2; in1 = in2;
            }
            finish(pending);
            // every storing wave drains BEFORE the barrier behind which one lane signals for all of them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LSF_CHAIN_TRACE
            if (trace_row && lane == 0) trace_row[16 + wave] = __builtin_amdgcn_s_memrealtime();
#endif
            if (c == 0 && threadIdx.x == 0) {  // the arg-max of an all-zero update: the first voxel of the launch's range
                const unsigned long long q = pack_max(0.0f, linear_index(g, 0, 0, g.z_begin));
                best = q > best ? q : best;
            }

            // ---- reduce, commit to the record of this iteration, publish
            const unsigned long long m = wave_max_u64(best);
            double s[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] = ENERGY != LSF_ENERGY_NONE ? wave_sum_f64(en[k]) : 0.0;
            if (lane == 0) {
                s_max[wave] = m;
#pragma unroll
                for (int k = 0; k < 3; ++k) s_sum[k][wave] = s[k];
            }
            __syncthreads();
            LSF_CT(4);
            if (wave == 0) {
                // the waves' partial results, one per lane (a serial loop over LDS words on one lane took 3 us per item)
                unsigned long long mm = wave_max_u64(lane < waves ? s_max[lane] : 0ull);
                double t[3] = {0.0, 0.0, 0.0};
                if (ENERGY != LSF_ENERGY_NONE) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) t[k] = wave_sum_f64(lane < waves ? s_sum[k][lane] : 0.0);
                }
                if (lane == 0) {
                    lsf_record_slot* slot = record_slot(record);
                    if (mm != 0ull) atomicMax(reinterpret_cast<unsigned long long*>(&slot->max_packed), mm);
                    if (ENERGY != LSF_ENERGY_NONE) {
                        double* dst[3] = {&slot->data_energy, &slot->smoothing_energy, &slot->level_set_energy};
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            if (t[k] != 0.0) atomicAdd(dst[k], t[k]);
                    }
                    const unsigned bits = (unsigned)(mm >> 32);
                    if (bits > s_worst_bits) s_worst_bits = bits;
                    __hip_atomic_store(progress + c, it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            LSF_CT(5);
        }
    }
    if (threadIdx.x == 0 && !(__uint_as_float(s_worst_bits) < plan.reach_limit))
        __hip_atomic_store(control + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct ChainArgs {
    unsigned blocks, threads;
    hipStream_t s;
    vf4 *state_a, *state_b;
    const float* canonical;
    Grid g;
    Params p;
    lsf_iteration_record* records;
    const int* band_list;
    unsigned band_count;
    ChainPlan plan;
    unsigned* scratch;
    int* resident;  // out: workgroups of this instantiation that fit one CU (occupancy query)
};

template <int D, int SMOOTH, bool LEVELSET, int DATA, int ENERGY>
void chain_one(const ChainArgs& a) {
    auto kernel = slavcheva_state_chain_kernel<D, SMOOTH, LEVELSET, DATA, ENERGY>;
    static int cached[64];  // per instantiation (the workgroup size is fixed per process) and device: 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), (int)a.threads, 0) != hipSuccess) n = 0;
        cached[dev] = n + 1;
    }
    const int fits = cached[dev] - 1;
    *a.resident = fits;
    if (fits < (int)(kCuBlock / a.threads)) return;
    hipLaunchKernelGGL(kernel, dim3(a.blocks), dim3(a.threads), 0, a.s, a.state_a, a.state_b, a.canonical, a.g, a.p,
                       a.records, a.band_list, a.band_count, a.plan, a.scratch);
}

template <int D, int SMOOTH, bool LEVELSET, int DATA>
void chain_energy(int energy, const ChainArgs& a) {
    switch (energy) {
        case LSF_ENERGY_DIRECT: chain_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_DIRECT>(a); break;
        case LSF_ENERGY_VECTORIZED: chain_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_VECTORIZED>(a); break;
        default: chain_one<D, SMOOTH, LEVELSET, DATA, LSF_ENERGY_NONE>(a); break;
    }
}

template <int D>
void chain_terms(const lsf_slavcheva_params* q, const ChainArgs& a) {
    const bool killing = q->smoothing_method == LSF_SMOOTHING_KILLING;
    const bool ls = q->level_set_enabled != 0;
    const bool fdm = q->data_method == LSF_DATA_THRESHOLDED_FDM;
    const int e = q->energy_mode;
#define LSF_PICK(S, L, DM) chain_energy<D, S, L, DM>(e, a)
    if (killing) {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_KILLING, false, LSF_DATA_BASIC); }
    } else {
        if (ls) { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, true, LSF_DATA_BASIC); }
        else    { if (fdm) LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_THRESHOLDED_FDM); else LSF_PICK(LSF_SMOOTHING_TIKHONOV, false, LSF_DATA_BASIC); }
    }
#undef LSF_PICK
}

// CUs of the current device; LSF_CHAIN_BLOCKS caps the chain's grid (measurements)
inline unsigned chain_units() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (const char* e = getenv("LSF_CHAIN_BLOCKS")) {
            const int v = atoi(e);
            if (v > 0 && v < n) n = v;
        }
        cached[dev] = n;
    }
    return (unsigned)cached[dev];
}

inline bool chain_list_ok(const lsf_grid* grid, const int32_t* band_list, int64_t band_count) {
    // the INTERIOR kernel's addressing: 32-bit byte offsets into the whole float4 state
    return band_list && band_count > 0 && band_count <= 0x7fffffffll &&
           16ll * grid->nz * grid->ny * grid->nx < 0xffffffffll;
}

}  // namespace

#ifdef LSF_CHAIN_TRACE
extern "C" int lsf_debug_set_chain_trace(void* rows) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &rows, sizeof(rows));
}
#endif

extern "C" int64_t lsf_state_chain_scratch_elements(int64_t band_count, int32_t stages) {
    if (band_count <= 0) return 0;
    const ChainShape s = chain_shape(band_count, stages, chain_units());
    return (int64_t)kControlWords + s.progress_words + 2ll * s.chunks;
}

extern "C" int lsf_state_chain_plan(const lsf_grid* grid, const int32_t* band_list, int64_t band_count, int32_t stages,
                                    int32_t* scratch, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!scratch || !chain_list_ok(grid, band_list, band_count)) return LSF_ERR_BAD_ARGUMENT;
    const ChainShape s = chain_shape(band_count, stages, chain_units());
    const long long row = grid->nx, slice = (long long)grid->nx * grid->ny;
    const long long reach = kReachSlices * ((grid->dims == 3 ? slice : 0ll) + row + 1);
    hipLaunchKernelGGL(chain_plan_kernel, dim3(1), dim3(kCuBlock), 0, as_stream(stream), band_list, (unsigned)band_count,
                       s.units, s.chunks, reach, reinterpret_cast<unsigned*>(scratch) + kControlWords + s.progress_words);
    return launch_status();
}

extern "C" int lsf_slavcheva_state_chain(float* state_a, float* state_b, const float* canonical, const lsf_grid* grid,
                                         const lsf_slavcheva_params* params, lsf_iteration_record* records,
                                         const int32_t* band_list, int64_t band_count, int32_t iterations,
                                         int32_t stages, int32_t* scratch, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!state_a || !state_b || state_a == state_b || !canonical || !params || !records || !scratch ||
        iterations < 0 || !chain_list_ok(grid, band_list, band_count) || grid->z_begin != 0 || grid->z_end != grid->nz)
        return LSF_ERR_BAD_ARGUMENT;
    if (iterations == 0) return 0;
    Grid g = make_grid(grid, 4);
    g.fast_ok = 1;
    g.wide_ok = 1;
    const ChainShape s = chain_shape(band_count, stages, chain_units());
    ChainPlan plan;
    plan.iterations = (unsigned)iterations;
    plan.chunks = s.chunks;
    plan.units = s.units;
    plan.stages = s.stages;
    plan.members = s.members;
    plan.progress_words = s.progress_words;
    // 2 s of the 100 MHz clock: healthy waits take microseconds (LSF_CHAIN_TIMEOUT_MS: debugging knob)
    static const unsigned timeout_ticks = [] {
        const char* e = getenv("LSF_CHAIN_TIMEOUT_MS");
        const long v = e ? atol(e) : 0;
        return (v > 0 && v <= 20000) ? (unsigned)v * 100000u : 200u * 1000u * 1000u;
    }();
    plan.timeout_ticks = timeout_ticks;
    plan.reach_limit = kReachLimit;
    Params p;
    p.lambda64 = params->isomorphic_enforcement_factor_f64;
    p.rate = params->rate;
    p.w_data = params->data_term_weight;
    p.w_smooth = params->smoothing_term_weight;
    p.w_level_set = params->level_set_term_weight;
    p.lambda32 = params->isomorphic_enforcement_factor;
    p.killing_c1 = params->killing_c1;
    p.zero_gradient_on_snap = params->zero_gradient_on_snap;
    // every polled word starts at zero in EVERY launch (control + progress: a block of its own at the allocation's
    // start, a multiple of 16 bytes)
    if (hipMemsetAsync(scratch, 0, (size_t)(kControlWords + s.progress_words) * 4u, as_stream(stream)) != hipSuccess)
        return launch_status();  // a hipError_t value, as the ABI's rules say (and the sticky error is cleared)
    int resident = 0;
    ChainArgs a{s.workgroups, s.threads, as_stream(stream), reinterpret_cast<vf4*>(state_a), reinterpret_cast<vf4*>(state_b),
                canonical, g, p, records, band_list, (unsigned)band_count, plan, reinterpret_cast<unsigned*>(scratch),
                &resident};
    if (grid->dims == 2) chain_terms<2>(params, a);
    else chain_terms<3>(params, a);
    if (resident < (int)(kCuBlock / s.threads)) return LSF_ERR_NOT_RESIDENT;
    return launch_status();
}

extern "C" int lsf_state_chain_shape(int64_t band_count, int32_t stages, int32_t* out4) {
    if (!out4 || band_count <= 0) return LSF_ERR_BAD_ARGUMENT;
    const ChainShape s = chain_shape(band_count, stages, chain_units());
    out4[0] = (int32_t)s.workgroups;
    out4[1] = (int32_t)s.stages;
    out4[2] = (int32_t)s.chunks;
    out4[3] = (int32_t)s.units;
    return 0;
}
