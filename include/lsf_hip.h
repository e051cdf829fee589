/*
 * lsf_hip.h -- C ABI of liblsf_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the per-voxel
 * warp-field gradient-descent path of KillingFusion / SobolevFusion-style non-rigid TSDF alignment.
 *
 * This is the drop-in boundary under the Python classes SlavchevaOptimizer2d / HierarchicalOptimizer2d/3d.
 * The reference (Algomorph/LevelSetFusion-Python) crosses its only language boundary through a boost-python
 * module `level_set_fusion_optimization` (un-vendored C++); the entry points below are what a binding for
 * THIS path binds instead.  Each one names the reference interface it replaces (path:line under the
 * reference checkout).
 *
 * Rules of the ABI
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 on success or a
 *     hipError_t value; nothing throws.  LSF_ERR_* (negative) flag argument errors detected on the host.
 *   - The library owns no FIELD memory: every field, state, list, record and scratch buffer is the caller's.  What it
 *     does own: per slab communicator (lsf_slab_comm_create ... _destroy) a HIP stream, a few events and two small
 *     face-count buffers (one hipMalloc, one hipHostMalloc); per host thread and device one timing-less event
 *     (lsf_state_run_begin).  Its only process-wide state is read-only after first use: the table of RCCL entry points
 *     the z-slab runtime binds once (under a mutex) and per-device cached device attributes.  All field pointers are
 *     DEVICE pointers supplied by the caller (e.g. torch.Tensor.data_ptr()), all launches are asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream); the functions that also WAIT say so
 *     (lsf_state_run_begin / _finish, lsf_slab_face_counts_end).  Re-entrant across streams, devices and host threads.
 *   - Layouts (MI355X-first):
 *       scalar fields      float32 [z][y][x]            (nz = 1 for 2-D)
 *       vector fields      float32 PLANAR [c][z][y][x]  c = 0:x(u) 1:y(v) 2:z(w); `dims` planes
 *       packed live field  float4  [z][y][x] = (live, d/dx live, d/dy live, d/dz live)   (gather operand)
 *       interleaved        float32 [z][y][x][c]         only at the API edge (lsf_interleave/deinterleave)
 *   - Slab support: a launch processes slices z in [z_begin, z_end) of an array that holds nz slices
 *     (owned slab + halo).  z_global_offset is added to z when a voxel index is reported (arg-max).
 *   - Arithmetic: float32, multiply and add rounded separately (-ffp-contract=off), in the operation order
 *     of oracle/lsf_oracle.py, so results are bit-identical to the oracle except for sum reductions.
 */
#ifndef LSF_HIP_H
#define LSF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSF_ABI_VERSION 4
#define LSF_MAX_KERNEL_TAPS 31

#define LSF_ERR_BAD_ARGUMENT (-1)
#define LSF_ERR_BAD_DIMS (-2)
#define LSF_ERR_KERNEL_TOO_LONG (-3)
#define LSF_ERR_RCCL_UNAVAILABLE (-4) /* librccl.so could not be bound at run time */
#define LSF_ERR_RCCL_FAILED (-5)      /* an RCCL call returned an error (its text goes to stderr) */
#define LSF_ERR_NOT_RESIDENT (-6)     /* lsf_hip_chain.h: the chain kernel's workgroups cannot all be resident on this device */

/* extents of one field as stored on this device */
typedef struct lsf_grid {
    int32_t dims;            /* 2 or 3 */
    int32_t nz, ny, nx;      /* allocated extents, nz = 1 when dims == 2 */
    int32_t z_begin, z_end;  /* slices processed by the launch, 0 <= z_begin <= z_end <= nz */
    int32_t z_global_offset; /* global z of local slice 0 (for reported voxel indices) */
    int32_t y_global_offset; /* global y of local row 0: slabs cut along y; 0 otherwise */
    /* lsf_slavcheva_state_iteration only: energies are accumulated for slices in [energy_z_begin, energy_z_end) --
     * a z-slab launch that recomputes halo slices must not count them twice.
     * energy_z_end <= energy_z_begin (e.g. both 0): every slice of the launch counts. */
    int32_t energy_z_begin, energy_z_end;
    /* Slabs cut along y (lsf_slavcheva_state_iteration on band lists and lsf_state_finalize_listed honour these; every
     * other entry point requires y_global_offset == 0 and ny_global in {0, ny}): the local array holds rows
     * [y_global_offset, y_global_offset + ny) of a volume with ny_global rows (0 = ny: not cut along y).  Reported voxel
     * indices are ((z + z_global_offset) * ny_global + y + y_global_offset) * nx + x, gather positions are formed from the
     * GLOBAL row (float32(y + y_global_offset) + displacement, as with z), and energies count rows in
     * [energy_y_begin, energy_y_end) only (energy_y_end <= energy_y_begin: every row). */
    int32_t ny_global;
    int32_t energy_y_begin, energy_y_end;
} lsf_grid;

/* per-iteration reduction record written by the iteration kernels (one record per iteration).
 * max_packed = (float bits of the maximum vector length << 32) | ~(linear voxel index): the largest value
 * with the smallest index wins an unsigned 64-bit max, which reproduces numpy's first-arg-max tie break
 * (hierarchical_optimizer2d.py:223-225, slavcheva_optimizer2d.py:213-215).  0 = iteration not executed.
 *
 * A record is kept as LSF_RECORD_SLOTS partial records 4 KiB apart (different memory channels); a block accumulates into slot
 * (block id mod 8), i.e. the slot of its XCD.  The record's VALUE is the max over the slots' max_packed and the sum
 * over the slots' energies -- the gate below, and the host when it decodes records, combine them that way.  (Atomics
 * of ~2000 blocks on one address serialise at ~12 ns each: 0.10 -> 0.055 ms for the band-list kernel at 256^3.) */
#define LSF_RECORD_SLOTS 8
typedef struct lsf_record_slot {
    uint64_t max_packed;
    double data_energy;      /* Slavcheva: sum over band of 0.5*diff^2 ; hierarchical: sum diff^2 */
    double smoothing_energy; /* un-weighted */
    double level_set_energy; /* un-weighted */
    uint64_t pad[508];
} lsf_record_slot; /* 4096 bytes: the slots of a record land in different L2 / memory channels (256 bytes apart, the
                      eight atomics streams of a launch's ~1000 blocks shared one: 0.0368 -> 0.0353 ms at 256^3) */
typedef struct lsf_iteration_record {
    lsf_record_slot slot[LSF_RECORD_SLOTS];
} lsf_iteration_record; /* 32 KiB */

/* HOST-side reading of records that have been copied to the host (no device involved): the value of each of n records,
 * combined over its partial slots as above.  slots = n x n_slots x slot_words int64 words, the first four words of every
 * slot being max_packed and the three energies (slot_words >= 4: 512 for whole records, 4 when only the used words were
 * copied; n_slots = LSF_RECORD_SLOTS, or a multiple of it when the slots of several ranks' records stand side by side).
 * Out, n entries each (energies3: n x 3): the maximum as a float, the voxel index it was found at, the energy sums
 * (slots added in ascending order), executed = 1 where the record's maximum word is non-zero. */
int lsf_records_decode(const int64_t *slots, int32_t n, int32_t n_slots, int32_t slot_words, float *max_value,
                       int64_t *argmax, double *energies3, uint8_t *executed);

/* Device-side convergence gate.  The reference tests its stop condition on the host after every iteration
 * (hierarchical_optimizer2d.py:169-171, slavcheva_optimizer2d.py:360-362); here every kernel of iteration i
 * looks at the record of iteration i-1 and turns itself into a no-op when that iteration already met the
 * stop condition (or was itself a no-op), so the host may enqueue iterations in batches without a sync and
 * still get exactly the reference's iteration count.  prev_record == NULL: always run.
 *   mode LSF_GATE_HIERARCHICAL: run iff NOT (max < a)            (a = maximum_warp_update_threshold)
 *   mode LSF_GATE_SLAVCHEVA:    run iff  a < max  AND  max < b   (a, b = lower / upper warp thresholds) */
#define LSF_GATE_HIERARCHICAL 0
#define LSF_GATE_SLAVCHEVA 1
#define LSF_GATE_OPEN 2 /* never closes; only names the previous record (lsf_hier_params::previous_max) */
typedef struct lsf_gate {
    const lsf_iteration_record *prev_record; /* DEVICE pointer or NULL */
    int32_t mode;
    float a;
    float b;
} lsf_gate;

/* ---------------------------------------------------------------------------------------------------- */
int lsf_abi_version(void);
/* first 16 hex digits of the SHA-256 over the NORMALISED text of this header (comments stripped, white space collapsed:
 * _build.py::abi_hash) as it stood when the library was compiled -- every struct and prototype goes into it, so a
 * binding written against another revision of the header is refused at load time (levelsetfusion-python_amd/_lib.py
 * compares it with the hash of the header it was written against) whether or not someone remembered to bump
 * LSF_ABI_VERSION.  "unknown" when compiled by hand. */
const char *lsf_abi_hash(void);
/* name of the code object's target, e.g. "gfx950" */
const char *lsf_target_arch(void);
/* identity of the build: the first 16 hex digits of the SHA-256 over the library's sources (csrc/, include/), set by the
 * build script; "unknown" when compiled by hand.  bench.py reports the committed rocprofv3 HBM-traffic figures only for
 * the build they were measured on (profiles/traffic.json carries the id). */
const char *lsf_build_id(void);

/* ---- layout helpers (API edge only) ---------------------------------------------------------------- */
/* [z][y][x][c] -> [c][z][y][x] and back; n_voxels = nz*ny*nx, channels = dims */
int lsf_deinterleave(const float *interleaved, float *planar, int64_t n_voxels, int32_t channels, void *stream);
int lsf_interleave(const float *planar, float *interleaved, int64_t n_voxels, int32_t channels, void *stream);

/* ---- z-slab halo staging (new design; the reference has no distributed code) -----------------
 * Copies `halo` consecutive slices starting at z_lo / z_hi of a scalar field [z][y][x] (channel 0, may be NULL) and of
 * the `planes` planes of a planar vector field (channels 1.., may be NULL) into (unpack = 0) or out of (unpack = 1) the
 * contiguous messages msg_lo / msg_hi (each [1 + planes][halo][ny][nx] floats; NULL = no neighbour on that side). */
int lsf_halo_copy(float *scalar, float *planar, float *msg_lo, float *msg_hi, const lsf_grid *grid, int32_t planes,
                  int32_t halo, int32_t z_lo, int32_t z_hi, int32_t unpack, void *stream);

/* ---- a1/a2: resample a scalar field under a warp ---------------------------------------------------
 * replaces nonrigid_opt/field_warping.py:67-85 (warp_field, oob_value = 1) and :88-109
 * (warp_field_replacement, oob_value = replacement) with utils/sampling.py:139-175,222-263. */
int lsf_warp_field(const float *field, const float *warp_planar, float *out, const lsf_grid *grid,
                   float oob_value, void *stream);

/* ---- a3: truncation-aware re-warp -------------------------------------------------------------------
 * replaces nonrigid_opt/field_warping.py:112-151 and the C++ twin cpp.warp_field_advanced called at
 * nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:227-228.  warp (and gradient, may be NULL) are zeroed in
 * place where the new value snaps to +-1.  flags: bit0 band_union_only, bit1 known_values_only,
 * bit2 substitute_original. */
int lsf_warp_field_advanced(const float *canonical, const float *live, float *warp_planar,
                            float *gradient_planar, float *new_live, const lsf_grid *grid, int32_t flags,
                            void *stream);

/* ---- a4 + packing: np.gradient of the live field, packed with it as float4 ------------------------
 * replaces np.gradient at hierarchical_optimizer2d.py:126 (+ the four pyramids' level-0 inputs :128-131) */
int lsf_pack_live_gradient(const float *live, float *packed4, const lsf_grid *grid, void *stream);

/* ---- a5/a6: pyramid restrict (2^D block mean, per channel) and prolong (repeat, no rescale) -------
 * replaces nonrigid_opt/hierarchical/pyramid.py:45-56 and hierarchical_optimizer2d.py:155-156.
 * `fine` describes the fine grid; the coarse grid has every extent halved (nz stays 1 in 2-D).
 * restrict: interleaved channels (1 = scalar, 4 = packed live field).  prolong: planar vector field. */
int lsf_restrict_mean(const float *fine, float *coarse, const lsf_grid *fine_grid, int32_t channels,
                      void *stream);
int lsf_prolong_repeat(const float *coarse_planar, float *fine_planar, const lsf_grid *fine_grid, void *stream);

/* LINEAR resampling strategy, 3-D only: replaces math_utils/resampling.py:29-80 (upsample2x_linear: 0.75 / 0.25
 * trilinear prolongation, edge padded) and :83-126 (downsample2x_linear: 4x4x4 window with the reference's literal
 * weights, edge padded).  `fine_grid` describes the FINE grid (even extents); interleaved channels 1 or 4. */
int lsf_upsample2x_linear(const float *coarse, float *fine, const lsf_grid *fine_grid, int32_t channels,
                          void *stream);
int lsf_downsample2x_linear(const float *fine, float *coarse, const lsf_grid *fine_grid, int32_t channels,
                            void *stream);

/* ---- a9/a10: one pass of the separable convolution along one axis (0 = x, 1 = y, 2 = z) ------------
 * replaces math_utils/convolution.py:70-111 (convolve_with_kernel) and :114-132 (…_preserve_zeros) pass by
 * pass: out[i] = sum_j k[j]*in[i + n/2 - j], zero padded, accumulated in float64 in tap order, stored
 * float32.  zero_mask_source (may be NULL): where |zero_mask_source| < 1e-6 the output is forced to 0
 * (convolution.py:118,123,127).  Operates on `planes` planes of a planar vector field. */
int lsf_convolve_axis(const float *in_planar, float *out_planar, const float *zero_mask_source,
                      const lsf_grid *grid, int32_t planes, int32_t axis, const double *taps_host,
                      int32_t n_taps, const lsf_gate *gate, void *stream);
/* the x and the y pass of a 3-D filter in one launch, without a zero mask (levels below lsf_convolve_xyz's size, where the
 * passes are launch-bound): the result equals lsf_convolve_axis along x, then along y, bit for bit, on the grid's z-range.
 * nx % 4 == 0, 3 / 5 / 7 / 9 taps (LSF_ERR_BAD_DIMS / LSF_ERR_KERNEL_TOO_LONG otherwise). */
int lsf_convolve_xy(const float *in_planar, float *out_planar, const lsf_grid *grid, int32_t planes,
                    const double *taps_host, int32_t n_taps, const lsf_gate *gate, void *stream);
/* the LAST pass of a hierarchical iteration's 3-D filter (axis 2; axis 1 is accepted too -- the reference's 2-D filter ends
 * with its x pass, convolution.py:77-83, which this does not cover; 3 / 5 / 7 / 9 taps, no zero mask): the
 * same pass, which also moves the warp by the filtered gradient it writes, component by component -- warp -= rate * out
 * (hierarchical_optimizer2d.py:220-222; what lsf_convolve_xyz does on large levels): lsf_hier_update is then called with
 * a NULL warp (the maximum only), or not at all when the next iteration takes the maximum (lsf_hier_params::previous_max).
 * LSF_ERR_BAD_DIMS for other tap counts. */
int lsf_convolve_axis_update(const float *in_planar, float *out_planar, float *warp_planar, float rate,
                             const lsf_grid *grid, int32_t planes, int32_t axis, const double *taps_host,
                             int32_t n_taps, const lsf_gate *gate, void *stream);

/* the three passes of a 3-D filter (x, then y, then z: convolution.py:94-105) in one launch, without a zero mask: the
 * result equals three lsf_convolve_axis calls bit for bit (x and y passes on every slice the z pass reads, z pass on the
 * grid's z-range: a z-slab filters its owned slices from raw input whose halo slices are valid).  nx % 4 == 0,
 * 3 / 5 / 7 / 9 taps; LSF_ERR_BAD_DIMS / LSF_ERR_KERNEL_TOO_LONG otherwise (the caller then runs the single passes).
 * warp_planar (may be NULL): the filtered field also moves the warp, warp -= rate * out, component by component -- the
 * hierarchical optimizer's update (hierarchical_optimizer2d.py:220-221) without another pass over out; the caller
 * then calls lsf_hier_update with warp_planar = NULL for the record's maximum. */
int lsf_convolve_xyz(const float *in_planar, float *out_planar, float *warp_planar, float rate, const lsf_grid *grid,
                     int32_t planes, const double *taps_host, int32_t n_taps, const lsf_gate *gate, void *stream);

/* the same zero-preserving pass (zero_mask_source required, 3 / 5 / 7 / 9 taps) at the voxels of a band list only */
int lsf_convolve_axis_listed(const float *in_planar, float *out_planar, const float *zero_mask_source,
                             const lsf_grid *grid, int32_t planes, int32_t axis, const double *taps_host,
                             int32_t n_taps, const lsf_gate *gate, const int32_t *band_list, int64_t band_count,
                             void *stream);

/* ---- hierarchical optimizer iteration ---------------------------------------------------------------
 * replaces one pass of the loop body nonrigid_opt/hierarchical/hierarchical_optimizer2d.py:184-225:
 *   resample live and its gradients under `warp` (a1,a2), data term (a7), Tikhonov = Laplacian of the
 *   previous gradient (a8), and -- when apply_update != 0 (no gradient kernel) -- warp -= rate*g and the
 *   max-update reduction (a11).  With a gradient kernel the caller runs lsf_convolve_axis passes on
 *   g_out and then lsf_hier_update.
 * gate (may be NULL): see lsf_gate.  g_prev_planar may be NULL when Tikhonov is off; g_out_planar may be NULL
 * when apply_update is set and the gradient is not wanted. */
typedef struct lsf_hier_params {
    float data_term_amplifier;
    float tikhonov_strength;   /* used when tikhonov_enabled */
    float rate;
    int32_t tikhonov_enabled;
    int32_t apply_update;
    int32_t compute_energy;    /* accumulate sum(diff^2) into the record's data_energy */
    int32_t previous_max;      /* 3-D, Tikhonov on, apply_update off, gate->prev_record set: ALSO write the maximum length
                                  (and arg-max) of g_prev -- the previous iteration's final gradient, which this kernel
                                  reads anyway -- into gate->prev_record.  For runs whose stop test cannot fire
                                  (threshold <= 0): the maximum is then only a log value and needs no pass of its own
                                  (lsf_hier_update with a NULL warp) except after the last iteration. */
    int32_t reserved;
    /* z-slab runs (3-D): the packed live field -- the only operand read through the data-dependent gather -- may hold MORE
     * slices than the grid: packed_nz > 0 says how many, packed_z_global_offset the global z of its slice 0 (0 for a copy
     * of the whole level replicated on every rank, SURVEY 8e: the gather then never leaves the device however far the
     * cumulative warp reaches).  packed_nz == 0: the packed field has the grid's own extent and offset. */
    int32_t packed_nz;
    int32_t packed_z_global_offset;
} lsf_hier_params;

int lsf_hier_iteration(const float *packed_live4, const float *canonical, float *warp_planar,
                       const float *g_prev_planar, float *g_out_planar, const lsf_grid *grid,
                       const lsf_hier_params *params, const lsf_gate *gate, lsf_iteration_record *record,
                       void *stream);

/* warp -= rate*g ; the record's max_packed = max |g|   (hierarchical_optimizer2d.py:220-225).  warp_planar may be
 * NULL: only the maximum (the warp was moved by lsf_convolve_xyz already). */
/* A whole 2-D level, K = iterations_per_launch (1..8) iterations per launch: temporal blocking through LDS (round 6;
 * BASELINE config 2 is launch-bound -- a 512^2 level is 1024 voxels per CU).  Replaces `iterations` calls of
 * lsf_hier_iteration (Tikhonov term, apply_update = 1, no energies; hierarchical_optimizer2d.py:184-225) with
 * ceil(iterations / K) launches whose workgroups each advance a 32 x 32 tile K iterations inside LDS, recomputing the
 * K - 1 - j rings of voxels around it that iteration j needs (identical arithmetic on identical inputs: identical results).
 * Launch b reads (warp_a, g_a) when b is even, (warp_b, g_b) when odd, and writes the other pair.  records[0..iterations):
 * one record per iteration (the caller has zeroed them), maxima only.
 * threshold <= 0 (the stop test cannot fire): after the call the warp and the last iteration's gradient are in the _a
 * buffers when ceil(iterations / K) is even, in the _b buffers otherwise.
 * threshold > 0 (hierarchical_optimizer2d.py:169-171: an iteration whose maximum is below it is the level's last): launch
 * b > 0 first looks at the K records of launch b - 1 and does nothing when one of them is below the threshold or empty.  The
 * launch in which the level converged is then the last to have written anything and its input pair is intact: the caller
 * reads the records, finds the first iteration j below the threshold and, unless j is its launch's last iteration, calls
 * again with that launch's input pair as (_a), its output pair as (_b), iterations = j + 1 - b K and threshold = 0.
 * With a gradient kernel (taps_host / n_taps = 3, 5, 7 or 9; NULL / 0: none; hierarchical_optimizer2d.py:213-216): the
 * launch also runs the filter's y and x passes (convolution.py:77-83) and the update behind them -- what lsf_hier_iteration
 * with apply_update = 0, two lsf_convolve_axis passes and lsf_hier_update do in four launches; params->apply_update must
 * then be 0 as it is for those.  An iteration consumes n_taps / 2 + 1 rings of the tile's surroundings instead of one:
 * iterations_per_launch * (n_taps / 2 + 1) <= 8 (two iterations per launch for seven taps), LSF_ERR_BAD_ARGUMENT otherwise.
 * Without the Tikhonov term (tikhonov_enabled = 0) a voxel's update depends on nothing around it but through the filter:
 * iterations_per_launch * (tikhonov_enabled + n_taps / 2) <= 8 in general.
 * LSF_ERR_BAD_DIMS for anything but dims = 2, compute_energy = 0, apply_update = (n_taps == 0). */
int lsf_hier_level_run_2d(const float *packed_live4, const float *canonical, float *warp_a, float *warp_b, float *g_a,
                          float *g_b, const lsf_grid *grid, const lsf_hier_params *params, const double *taps_host,
                          int32_t n_taps, lsf_iteration_record *records, int32_t iterations,
                          int32_t iterations_per_launch, float threshold, void *stream);
int lsf_hier_update(const float *g_planar, float *warp_planar, const lsf_grid *grid, float rate,
                    const lsf_gate *gate, lsf_iteration_record *record, void *stream);

/* ---- Slavcheva (KillingFusion / SobolevFusion) optimizer iteration ----------------------------------
 * replaces nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:163-236 (VECTORIZED) and :238-330 (DIRECT)
 * with data_term.py:169-227,334-358, smoothing_term.py:50-177, level_set_term.py:28-64.
 *
 * No Sobolev filter: lsf_slavcheva_state_iteration -- ONE kernel does gradient, warp = -g*rate, max-warp reduction,
 *   energies and the truncation-aware re-warp of the live field (a12-a18 + a3): the "fused per-voxel warp-update
 *   kernel", on the float4 state layout.
 * With a Sobolev filter: lsf_slavcheva_gradient (gradient + energies into g_out), lsf_convolve_axis passes
 *   (zero-preserving), lsf_slavcheva_update_rewarp -- planar fields. */
#define LSF_SMOOTHING_TIKHONOV 0
#define LSF_SMOOTHING_KILLING 1
#define LSF_DATA_BASIC 0
#define LSF_DATA_THRESHOLDED_FDM 2
#define LSF_ENERGY_NONE 0
#define LSF_ENERGY_DIRECT 1     /* per-voxel energies of DIRECT mode */
#define LSF_ENERGY_VECTORIZED 2 /* np.gradient-based smoothing energy of VECTORIZED mode */

typedef struct lsf_slavcheva_params {
    double isomorphic_enforcement_factor_f64; /* lambda as given (energies are accumulated in double) */
    float rate;
    float data_term_weight;
    float smoothing_term_weight;
    float level_set_term_weight;
    float isomorphic_enforcement_factor; /* float32(lambda) */
    float killing_c1;                    /* float32(-2*(1+lambda)), evaluated in double on the host */
    int32_t smoothing_method;
    int32_t data_method;
    int32_t level_set_enabled;
    int32_t energy_mode;
    int32_t zero_gradient_on_snap; /* DIRECT: 1 (field_warping.py:141) ; VECTORIZED: 0 */
    int32_t reserved;
} lsf_slavcheva_params;

/* band_list (may be NULL = every voxel of the z-range; LSF_BAND_ALL lists only): the SobolevFusion path on a band list --
 * gradient, the masked filter passes (lsf_convolve_axis_listed) and the update touch listed voxels only; the caller
 * zero-initialises the gradient / filter buffers and initialises BOTH ping-pong (live, warp) sets with (live, 0): a
 * zero-preserving filter cannot move gradient out of the band, so every other voxel keeps those values. */
int lsf_slavcheva_gradient(const float *live, const float *canonical, const float *warp_prev_planar,
                           float *g_out_planar, const lsf_grid *grid, const lsf_slavcheva_params *params,
                           const lsf_gate *gate, lsf_iteration_record *record, const int32_t *band_list,
                           int64_t band_count, void *stream);

/* Band lists for lsf_slavcheva_state_iteration: only listed voxels are visited.  This is exact, not an approximation:
 * a voxel outside the narrow-band union (|live| == |canonical| == 1, the test of slavcheva_optimizer2d.py:251-252 /
 * tsdf_set_routines.py:19-52) gets a zero gradient, hence warp 0 and live' = live, and so stays outside for the rest of
 * the optimisation.  The caller keeps canonical and params unchanged while a list is in use.  Records, live and warp
 * are identical with and without a list. */
#define LSF_BAND_ALL 0      /* every voxel of the narrow-band union */
#define LSF_BAND_INTERIOR 1 /* ... whose whole 3^D neighbourhood lies inside the (allocated) array */
#define LSF_BAND_BOUNDARY 2 /* ... the others (voxels on a face of the array) */

/* Band list of the grid's z-range [z_begin, z_end): the voxels with |live| != 1 or |canonical| != 1
 * (tsdf_set_routines.py:19-52; the `continue` of slavcheva_optimizer2d.py:251-252), in ascending index order --
 * all of them, or split into INTERIOR and BOUNDARY voxels.  An INTERIOR list lets lsf_slavcheva_state_iteration run a
 * kernel that never applies the reference's out-of-bounds rules (none can fire) and addresses all neighbours from one
 * per-lane offset; it requires 16 * nz * ny * nx < 2^32 (32-bit buffer offsets; LSF_ERR_BAD_ARGUMENT otherwise).
 * One iteration = one launch per non-empty list, all on the same record.
 *   1. lsf_band_count     counts per 1024-voxel chunk into scratch (lsf_band_scratch_elements(grid) int32 elements),
 *                         scans them, and writes the total to *count_out (device memory);
 *   2. the caller reads the total and allocates the list;
 *   3. lsf_band_list_fill writes the indices (same live / canonical / grid / subset / scratch as step 1). */
int64_t lsf_band_scratch_elements(const lsf_grid *grid);
int lsf_band_count(const float *live, const float *canonical, const lsf_grid *grid, int32_t subset,
                   int32_t *scratch, int64_t *count_out, void *stream);
int lsf_band_list_fill(const float *live, const float *canonical, const lsf_grid *grid, int32_t subset,
                       const int32_t *scratch, int32_t *list, void *stream);

/* ---- the fused iteration on the STATE layout (what the optimizers run when no Sobolev filter is configured) --------
 * state: float4 [z][y][x] = (live, u, v, w) (w = 0 in 2-D) -- the live field and the per-iteration warp travel together:
 * an iteration reads both through the same 3^D neighbourhood and writes both, so one 16-byte access replaces four
 * dword accesses to four planes.  band_list (may be NULL = every voxel of the z-range): ascending
 * voxel indices (z * ny + y) * nx + x from lsf_band_list_fill, band_count of them, band_subset the LSF_BAND_* it was
 * built with; state_out of BOTH ping-pong states must hold (live, 0) at unlisted voxels (lsf_state_pack with two
 * destinations does exactly that).  An INTERIOR list requires 16 * nz * ny * nx < 2^32.
 * lsf_state_pack: (live, warp planar or NULL = 0) -> state_a and, if not NULL, state_b, slices [z_begin, z_end).
 * lsf_state_unpack: state -> live and / or planar warp [c][z][y][x] and / or interleaved warp [z][y][x][c]. */
/* lsf_state_prepare: the start of an optimize() call in one pass over WHOLE arrays (z_begin = 0, z_end = nz): both
 * ping-pong states = (live, 0) (state_b may be NULL: the caller then writes it with lsf_state_pack behind its copy of
 * counts_out, where it overlaps the host's wait) and the counting step of lsf_band_count for the INTERIOR and the
 * BOUNDARY subset at once.
 * scratch: lsf_state_prepare_scratch_elements(grid) int32 (8-byte aligned); counts_out[0..4) (device) = INTERIOR and
 * BOUNDARY totals, then the number of voxels OUTSIDE the band with live = -canonical and the first of them (-1: none)
 * -- what lsf_state_finalize_listed needs to know about the voxels no list holds.  lsf_band_list_fill_prepared then writes one subset's list from the ballots the prepare pass kept
 * in scratch (16 bytes per 64 voxels) -- live and canonical are not read again.
 * state_a AND state_b NULL: the pass only counts; lsf_state_pack_needed then writes the two states (live, 0) ONLY where
 * an iteration can read them: the 1024-voxel chunks that have a voxel within `reach` voxels (per axis) of a chunk that
 * holds band voxels -- 3^D stencils reach 1, the re-warp gather of an update shorter than `reach` voxels reaches `reach`
 * (1 <= reach <= 8).  Everything else of the two buffers stays UNINITIALISED: a run is valid only while every
 * iteration's maximum update stays below `reach` (lsf_state_finalize_listed's guard_records check that on the device,
 * the host on the records it reads), and whole-state readers (lsf_state_unpack, lsf_state_finalize) must first complete the state with
 * invert = 1 (writes exactly the chunks the first call left out, from the verdicts it kept in scratch).  On a narrow
 * band this replaces 2 x 16 B per VOXEL of initialisation by 2 x 16 B per voxel NEAR THE BAND (a quarter of a 256^3
 * sphere pair). */
int64_t lsf_state_prepare_scratch_elements(const lsf_grid *grid);
int lsf_state_pack_needed(const float *live, float *state_a, float *state_b, const lsf_grid *grid, int32_t *scratch,
                          int32_t reach, int32_t invert, void *stream);

int lsf_band_list_fill_prepared(const lsf_grid *grid, int32_t subset, const int32_t *scratch, int32_t *list,
                                void *stream);
int lsf_state_prepare(const float *live, const float *canonical, float *state_a, float *state_b,
                      const lsf_grid *grid, int32_t *scratch, int64_t *counts_out, void *stream);
int lsf_state_pack(const float *live, const float *warp_planar, float *state_a, float *state_b,
                   const lsf_grid *grid, void *stream);
int lsf_state_unpack(const float *state, float *live_out, float *warp_planar_out, float *warp_interleaved_out,
                     const lsf_grid *grid, void *stream);
/* end of an optimize() call in one pass over slices [z_begin, z_end): state -> live_out, planar and / or interleaved warp
 * (each may be NULL) and -- when statistics16 is not NULL -- the convergence statistics of
 * lsf_warp_statistics (statistics16[0..8)) and lsf_tsdf_difference_statistics (statistics16[8..16)) of the final
 * fields (a20; cpp.build_warp_delta_statistics_2d / build_tsdf_difference_statistics_2d, slavcheva_optimizer2d.py:394-398).
 * scratch: lsf_state_finalize_scratch_elements(grid) doubles of device memory. */
int64_t lsf_state_finalize_scratch_elements(const lsf_grid *grid);
int lsf_state_finalize(const float *state, const float *canonical, float *live_out, float *warp_planar_out,
                       float *warp_interleaved_out, const lsf_grid *grid, float lower_threshold,
                       double *statistics16, double *scratch, void *stream);
/* the same pass for final fields that are planar (the SobolevFusion path: live [z,]y,x and warp [c][z,]y,x): live_out
 * (may be NULL, must not be live), the interleaved warp and the statistics in one pass instead of a copy, an interleave
 * and the two statistics kernels.  scratch as lsf_state_finalize. */
int lsf_planar_finalize(const float *live, const float *warp_planar, const float *canonical, float *live_out,
                        float *warp_interleaved_out, const lsf_grid *grid, float lower_threshold,
                        double *statistics16, double *scratch, void *stream);

/* lsf_state_finalize for whole arrays whose band lists are at hand: only the listed voxels are visited -- live_out must
 * already hold the INPUT live field and warp_interleaved_out zeros (nothing else can have changed); the statistics
 * take the unlisted voxels from lsf_state_prepare's counts_out[2..4): opposite_count of them have
 * |canonical - live| = 2, the first one at voxel first_opposite (-1: none), the others 0.  scratch as lsf_state_finalize.
 * skip_flag (may be NULL): a DEVICE word; when it is non-zero as the pass runs, live_out and warp_interleaved_out are
 * left untouched (and the statistics are meaningless) -- the chain add-on (include/lsf_hip_chain.h) raises such a word.
 * guard_records (may be NULL): the pass also leaves everything untouched when one of guard_records[0..guard_count) was
 * executed with a maximum update that is not below guard_limit (NaN included): the run's sparsely initialised states
 * (lsf_state_pack_needed) may then have been read where they were never written. */
int lsf_state_finalize_listed(const float *state, const float *canonical, float *live_out, float *warp_interleaved_out,
                              const lsf_grid *grid, const int32_t *const *band_lists, const int64_t *band_counts,
                              int32_t n_lists, int64_t opposite_count, int64_t first_opposite, float lower_threshold,
                              double *statistics16, double *scratch, const int32_t *skip_flag,
                              const lsf_iteration_record *guard_records, int32_t guard_count, float guard_limit,
                              void *stream);
int lsf_slavcheva_state_iteration(const float *state_in, const float *canonical, float *state_out,
                                  const lsf_grid *grid, const lsf_slavcheva_params *params, const lsf_gate *gate,
                                  lsf_iteration_record *record, const int32_t *band_list, int64_t band_count,
                                  int32_t band_subset, void *stream);

/* ---- the fused iteration over BOXES (3-D, INTERIOR band voxels) -----------------------------
 * The same pass of slavcheva_optimizer2d.py:238-330 as lsf_slavcheva_state_iteration over an INTERIOR band list, same
 * results, with the work cut differently: a box is 4 x 4 x 4 voxels whose lowest corner (x0, y0, z0) has coordinates that
 * are multiples of 4 -- origin = (z0 * ny + y0) * nx + x0 --, bit (lz * 4 + ly) * 4 + lx of `mask` says whether voxel
 * (x0 + lx, y0 + ly, z0 + lz) is an INTERIOR band voxel (LSF_BAND_INTERIOR: its whole 3^3 neighbourhood inside the array).
 * Boxes in ascending order of origin, every INTERIOR band voxel in exactly one box; voxels on the faces of the array keep
 * their BOUNDARY list and lsf_slavcheva_state_iteration.  A wave stages a box and its one-voxel shell through LDS instead
 * of loading 18 neighbours per voxel.  Requires dims = 3, extents that are multiples of 4, nz * ny * nx < 2^28
 * (LSF_ERR_BAD_DIMS otherwise).  lsf_band_boxes_count / _fill build the boxes from the ballots lsf_state_prepare kept in its
 * scratch: _count writes the number of boxes to *count_out (device) and keeps per-group counts in box_scratch
 * (lsf_band_boxes_scratch_elements(grid) int32), the caller reads the count, allocates, and _fill writes the boxes. */
typedef struct lsf_band_box {
    int32_t origin;
    int32_t reserved;
    uint64_t mask;
} lsf_band_box;
int64_t lsf_band_boxes_scratch_elements(const lsf_grid *grid);
/* subset: LSF_BAND_INTERIOR (the boxes of lsf_slavcheva_state_iteration_boxes) or LSF_BAND_ALL (every band voxel, those on
 * the faces included: lsf_sobolev_state_update_boxes); the same value in both calls */
int lsf_band_boxes_count(const lsf_grid *grid, int32_t subset, const int32_t *prepare_scratch, int32_t *box_scratch,
                         int64_t *count_out, void *stream);
int lsf_band_boxes_fill(const lsf_grid *grid, int32_t subset, const int32_t *prepare_scratch, const int32_t *box_scratch,
                        lsf_band_box *boxes, void *stream);
/* the canonical values of the boxes' voxels, box by box: canonical_boxed[64 b + (lz * 4 + ly) * 4 + lx] -- what the box walk
 * reads instead of the [z][y][x] array (256 contiguous bytes per box instead of sixteen rows of 16 bytes); once per call */
int lsf_band_boxes_canonical(const float *canonical, const lsf_grid *grid, const lsf_band_box *boxes, int64_t box_count,
                             float *canonical_boxed, void *stream);
int lsf_slavcheva_state_iteration_boxes(const float *state_in, const float *canonical_boxed, float *state_out,
                                        const lsf_grid *grid, const lsf_slavcheva_params *params, const lsf_gate *gate,
                                        lsf_iteration_record *record, const lsf_band_box *boxes, int64_t box_count,
                                        void *stream);

/* ---- a whole fixed-count call of the fused path, enqueued by the library in two host calls ---------------------------
 * replaces the LOOP of slavcheva_optimizer2d.py:354-388 (min_iterations == max_iterations: its stop test :360-362 cannot
 * fire) around the calls above, for whole volumes on band lists: lsf_state_run_begin launches lsf_state_prepare and the
 * initialisation of the two states (sparse_reach > 0: lsf_state_pack_needed with that reach, else both states in full,
 * the second one behind the copy of the list sizes when second_state_late != 0) and RETURNS when the four totals of
 * lsf_state_prepare are in totals_host; the caller sizes the two lists from them (totals_host[0] INTERIOR, [1] BOUNDARY
 * entries) and calls lsf_state_run_finish, which launches the list fills, `iterations` x lsf_slavcheva_state_iteration
 * per non-empty list (ungated; iteration i reads state[i % 2], writes the other, reduces into records[i], which the
 * caller has zeroed; with `boxes` -- room for totals_host[4] of them, box_scratch given to lsf_state_run_begin, and
 * `box_canonical`, room for 64 floats per box -- the INTERIOR voxels are walked box by box instead,
 * lsf_band_boxes_canonical + lsf_slavcheva_state_iteration_boxes: same results) and lsf_state_finalize_listed of the final state into live_out (which must hold the input live field:
 * the pass writes listed voxels only; statistics16 / finalize_scratch as there, may be NULL; with sparse states the pass
 * guards itself with the records), copies the records' used words and the statistics to the host in ONE transfer and
 * RETURNS when the stream has drained, the records decoded into `result` (lsf_records_decode).  The same launches in the
 * same order as the calls made one by one: identical results.  Both functions block the calling thread only (no
 * device-wide synchronisation); totals_host and words_host must be page-locked host memory; words_host and words_device
 * hold iterations x LSF_RECORD_SLOTS x 4 + 16 int64: the used words of every record slot, then the 16 statistics (as
 * doubles; zeros without statistics16).
 * Threshold-terminated calls (round 6; `loop` not NULL and min_iterations < max_iterations: every reference caller's
 * default, slavcheva_optimizer2d.py:360-362 -- min 1, max 100, lower threshold 0.1): `iterations` = max_iterations is the
 * number of records; iteration i >= min_iterations is launched behind the device-side gate on record i - 1 (lsf_gate,
 * LSF_GATE_SLAVCHEVA with the two thresholds), `check_interval` iterations at a time; after each batch the function reads
 * the batch's records (one small transfer, one wait) and stops enqueueing when the reference's loop would have ended --
 * so the executed count, every record and the final fields are the reference's, and at most check_interval - 1 gated
 * no-op launches are wasted.  The finalize pass then reads state[executed % 2]. */
typedef struct lsf_run_loop {
    int32_t min_iterations;  /* >= 1 (with 0 the reference never enters its loop: the caller handles that) */
    int32_t max_iterations;  /* the loop runs while it < min or (it < max and lower < max_warp < upper) */
    float lower_threshold, upper_threshold;
    int32_t check_interval;  /* >= 1 */
    int32_t reserved;
} lsf_run_loop;
typedef struct lsf_state_run {
    const float *live;        /* the input live field; read again by the sparse initialisation */
    const float *canonical;
    float *state[2];          /* the two ping-pong states, nz * ny * nx float4 each, contents undefined on entry */
    int32_t *prepare_scratch; /* lsf_state_prepare_scratch_elements(grid) int32 */
    int64_t *totals_device;   /* 5 int64 */
    int64_t *totals_host;     /* 5 int64, page-locked: lsf_state_prepare's four totals, then the number of boxes (or 0) */
    lsf_grid grid;            /* a whole volume: z_begin = 0, z_end = nz, no offsets */
    int32_t sparse_reach;     /* 0: both states are written in full */
    int32_t second_state_late;
    int32_t *box_scratch;     /* NULL, or lsf_band_boxes_scratch_elements(grid) int32: lsf_state_run_begin then also counts the
                                 boxes of lsf_slavcheva_state_iteration_boxes (totals_host[4]) */
    int32_t box_all;          /* 0: the boxes of the INTERIOR band voxels (lsf_state_run_finish); 1: of ALL band voxels
                                 (LSF_BAND_ALL: lsf_sobolev_run_finish) */
    int32_t reserved;
} lsf_state_run;
typedef struct lsf_state_run_result {
    float *max_value;    /* host arrays of `iterations` entries (energies3: 3 per iteration), as lsf_records_decode */
    int64_t *argmax;
    double *energies3;
    uint8_t *executed;
    int32_t final_state;    /* out: index of the state that holds the result (iterations % 2) */
    int32_t n_lists;        /* out: launches per iteration */
    int32_t reach_exceeded; /* out: sparse states and an update of sparse_reach voxels or more -- live_out was left alone
                               (the finalize pass's guard) and the call has to be repeated on full states */
    int32_t compact_faces;  /* out (lsf_slab_run_finish): 1 = only the band voxels of the faces travelled, 0 = whole slices
                               (no message buffer given, or a neighbour's face counts disagreed), -1 = nothing was exchanged */
} lsf_state_run_result;
int lsf_state_run_begin(const lsf_state_run *run, void *stream);
int lsf_state_run_finish(const lsf_state_run *run, const lsf_slavcheva_params *params, int32_t *list_interior,
                         int32_t *list_boundary, lsf_band_box *boxes, float *box_canonical, lsf_iteration_record *records,
                         int32_t iterations, const lsf_run_loop *loop /* NULL: `iterations` ungated launches */,
                         float *live_out,
                         float lower_threshold, double *statistics16, double *finalize_scratch, int64_t *words_device,
                         int64_t *words_host, lsf_state_run_result *result, void *stream);

/* The SobolevFusion counterpart of lsf_state_run_finish (round 6): the loop of slavcheva_optimizer2d.py:354-388 WITH a Sobolev
 * filter, for whole 3-D volumes of whole boxes (the conditions of lsf_sobolev_state_update_boxes), behind a
 * lsf_state_run_begin with box_scratch and box_all = 1.  Launches the list fills, the ascending list of ALL band voxels
 * (lsf_merge_sorted_runs into list_all -- room for totals_host[0] + totals_host[1] entries -- when both lists have entries),
 * the boxes (room for totals_host[4]), then per iteration lsf_sobolev_state_gradient_x (bricks) into g4_a and
 * lsf_sobolev_state_update_boxes (the final gradient into g4_b: in the last iteration of a fixed count, in every iteration
 * of a threshold-terminated call), the listed finalize pass and the read-backs; `loop`, records, words, result as
 * lsf_state_run_finish.  g4_a / g4_b: nz * ny * nx float4 each, ZERO-filled by the caller.  Same launches in the same order as
 * the calls made one by one: identical results. */
int lsf_sobolev_run_finish(const lsf_state_run *run, const lsf_slavcheva_params *params, const double *taps_host,
                           int32_t n_taps, int32_t *list_interior, int32_t *list_boundary, int32_t *list_all,
                           lsf_band_box *boxes, float *g4_a, float *g4_b, lsf_iteration_record *records, int32_t iterations,
                           const lsf_run_loop *loop, float *live_out, float lower_threshold, double *statistics16,
                           double *finalize_scratch, int64_t *words_device, int64_t *words_host,
                           lsf_state_run_result *result, void *stream);

/* ---- the SobolevFusion iteration on the float4 layouts (band lists) ------------------------------
 * replaces one pass of slavcheva_optimizer2d.py:163-236 / :238-330 WITH a Sobolev filter (math_utils/convolution.py:
 * 114-132) exactly as lsf_slavcheva_gradient + lsf_convolve_axis_listed x D + lsf_slavcheva_update_rewarp do on planar
 * fields -- same results -- with one vector-memory instruction per neighbour / tap instead of one per component:
 *   state  float4 [z][y][x] = (live, u, v, w)        as lsf_slavcheva_state_iteration (two ping-pong copies, both
 *                                                    (live, 0) at unlisted voxels: lsf_state_prepare / lsf_state_pack)
 *   g4     float4 [z][y][x] = (g_x, g_y, g_z, m)     raw gradient (m = 0), filter intermediates (m = mask bits 1 | 2 | 4 of the
 *                                                    three components, 8: a listed voxel), final gradient (m = 0); the caller
 *                                                    zero-initialises them once (unlisted voxels are never written)
 * One iteration = lsf_sobolev_state_gradient, lsf_convolve_axis_listed4 for every axis but the last (3-D: x, y; 2-D: y),
 * lsf_sobolev_state_update for the last (3-D: z, 2-D: x) -- each once per band list (ascending voxel indices of any
 * LSF_BAND_* subset; first_list != 0 on the call that stands for the unlisted voxels' zero update in the arg-max).
 * 3 / 5 / 7 / 9 taps (LSF_ERR_KERNEL_TOO_LONG otherwise); 16 * nz * ny * nx need not fit 32 bits.
 * zero_mask_source4: the RAW gradient for the FIRST pass (its own input: math_utils/convolution.py:118 computes the mask
 * before filtering) -- that pass leaves the three verdicts as bits in the fourth component of out4 --, NULL for every
 * later pass (and for lsf_sobolev_state_update behind at least one pass): the mask is then read from the fourth
 * component of in4's centre tap and handed on, one 16-byte load per voxel and pass less. */
int lsf_sobolev_state_gradient(const float *state, const float *canonical, float *g_raw4, const lsf_grid *grid,
                               const lsf_slavcheva_params *params, const lsf_gate *gate, lsf_iteration_record *record,
                               const int32_t *band_list, int64_t band_count, void *stream);
/* 3-D: lsf_sobolev_state_gradient and the x pass (the FIRST pass of a volume, math_utils/convolution.py:94-105) in ONE
 * launch: out4 = what lsf_convolve_axis_listed4(raw, out4, raw, axis 0) would hold, bit for bit, mask bits included; the
 * raw gradient is never stored.  band_list must then hold EVERY band voxel whose gradient can be non-zero (one
 * ascending list of the whole band, e.g. LSF_BAND_ALL): a tap at a voxel that is not in THIS list counts as zero.
 * out_bricks != 0 (extents that are multiples of 4: LSF_ERR_BAD_DIMS otherwise): out4 is laid out in BRICKS of 4 x 4 x 4
 * voxels, 1 KB each -- voxel (x, y, z) at float4 index (((z/4 * ny/4 + y/4) * nx/4 + x/4) * 64 + (z%4 * 4 + y%4) * 4 + x%4)
 * --, the layout lsf_sobolev_state_update_boxes stages its footprints from (a box's footprint is then 23 contiguous pieces
 * instead of 100 rows of 64 bytes a row's pitch apart). */
int lsf_sobolev_state_gradient_x(const float *state, const float *canonical, float *out4, const lsf_grid *grid,
                                 const lsf_slavcheva_params *params, const double *taps_host, int32_t n_taps,
                                 const lsf_gate *gate, lsf_iteration_record *record, const int32_t *band_list,
                                 int64_t band_count, int32_t out_bricks, void *stream);
/* zeros at the voxels of a band list in a float4 buffer (bricks != 0: in the brick layout of lsf_sobolev_state_gradient_x): a
 * gradient buffer whose non-zero entries all lie at the voxels of a PREVIOUS call's lists is made all-zero again without a
 * fill of the whole buffer (new design: the reference allocates its gradient per call, slavcheva_optimizer2d.py:343-346) */
int lsf_zero_listed4(float *field4, const lsf_grid *grid, const int32_t *band_list, int64_t band_count, int32_t bricks,
                     void *stream);
int lsf_convolve_axis_listed4(const float *in4, float *out4, const float *zero_mask_source4, const lsf_grid *grid,
                              int32_t axis, const double *taps_host, int32_t n_taps, const lsf_gate *gate,
                              const int32_t *band_list, int64_t band_count, void *stream);
/* Walk order of the LAST pass of a volume (along z): the ascending band list regrouped strip by strip -- `strips` strips of
 * ceil(ny / strips) rows, each swept through all slices, ascending inside a strip -- so that the z -/+ taps of a workgroup
 * are lines its own XCD has just read (results do not depend on the order).  out: band_count
 * entries; scratch: 3 * n_strips * nz int32 with n_strips = ceil(ny / ceil(ny / strips)) <= strips. */
int lsf_band_list_strip_major(const int32_t *band_list, int64_t band_count, const lsf_grid *grid, int32_t strips,
                              int32_t *out, int32_t *scratch, void *stream);
/* g_out4 of lsf_sobolev_state_update may be NULL: the iteration's filtered gradient is then not stored (a caller that
 * knows which iteration is the last one -- a fixed iteration count -- only needs that one's). */
int lsf_sobolev_state_update(const float *in4, const float *zero_mask_source4, const float *state_in, float *state_out,
                             float *g_out4, const lsf_grid *grid, const lsf_slavcheva_params *params, int32_t axis,
                             const double *taps_host, int32_t n_taps, const lsf_gate *gate,
                             lsf_iteration_record *record, const int32_t *band_list, int64_t band_count,
                             int32_t first_list, void *stream);
/* 3-D whole volumes: everything behind the FIRST pass in ONE launch, box by box -- the y pass, the z pass, the update and the
 * re-warp of slavcheva_optimizer2d.py:208-236 / math_utils/convolution.py:94-132 / field_warping.py:112-151, i.e. what
 * lsf_convolve_axis_listed4(in4, tmp, NULL, axis 1) followed by lsf_sobolev_state_update(tmp, NULL, ..., axis 2, ...,
 * first_list = 1) over one list of the whole band leave in state_out, g_out4 and the record, bit for bit, without `tmp`:
 * a wave stages the x-filtered gradient of a box's filter footprint (4 x (4 + 2c) x (4 + 2c) float4, c = n_taps / 2) and
 * the state's shell of the box through LDS.  in4: the output of
 * lsf_sobolev_state_gradient_x with out_bricks = 1 (bit 8 of its fourth component marks a listed voxel; every other brick
 * zero); g_out4 ([z][y][x] like the states) must not be in4 (other boxes
 * still read it) and may be NULL; boxes: lsf_band_boxes_count / _fill with LSF_BAND_ALL.  Requires dims = 3, extents that
 * are multiples of 4, the whole array as launch range, nz * ny * nx < 2^27 (LSF_ERR_BAD_DIMS otherwise). */
int lsf_sobolev_state_update_boxes(const float *in4, const float *state_in, float *state_out, float *g_out4,
                                   const lsf_grid *grid, const lsf_slavcheva_params *params, const double *taps_host,
                                   int32_t n_taps, const lsf_gate *gate, lsf_iteration_record *record,
                                   const lsf_band_box *boxes, int64_t box_count, void *stream);

/* ---- z-slab runtime of the fused path for multi-GPU runs (new design) -----------------------------
 * One process per GPU; rank r owns z-slices [z_begin, z_end) of its local array and keeps `halo` slices of its
 * neighbours on either interior side.  lsf_slab_state_iteration enqueues ONE whole iteration with one host call:
 *   1. lsf_slavcheva_state_iteration over the boundary parts (the `halo` owned slices next to each neighbour),
 *   2. the exchange of those slices of state_out with the neighbours -- ncclSend / ncclRecv in one group on the
 *      communicator's own HIP stream (RCCL over xGMI), each face one contiguous run of halo * ny * nx float4 --
 *   3. while the interior parts run on `stream`; 4. `stream` then waits for the halos.
 * Exchange groups: with a halo of k slices the faces need to travel only every k-th iteration when the iterations in
 * between recompute the neighbours' slices they still have valid inputs for (iteration j of a group runs over the owned
 * range widened by k - 1 - j slices, energies limited to the owned range by lsf_grid::energy_z_*; each iteration
 * consumes one slice of validity while every warp update stays below one voxel).  Those iterations pass exchange = 0.
 * RCCL is bound with dlopen (rccl_library_path, else "librccl.so" as already mapped into the process).
 * Communicator: rank 0 calls lsf_slab_unique_id, the 128 bytes travel to the other ranks by any means (the Python side
 * broadcasts them with torch.distributed), every rank calls lsf_slab_comm_create on its current device. */
#define LSF_SLAB_LAUNCH 0            /* launches only (boundary parts, then interior parts): inside an exchange group */
#define LSF_SLAB_EXCHANGE 1          /* boundary parts -> exchange || interior parts -> the stream waits for the halos */
#define LSF_SLAB_EXCHANGE_DEFERRED 2 /* the same, but the wait is left to the next call, which must be a RESUME */
#define LSF_SLAB_RESUME 3            /* "boundary" parts = what does not read the halos (runs while the previous call's
                                        exchange is in flight), wait for that exchange, then the "interior" parts */
typedef struct lsf_slab_comm lsf_slab_comm;
typedef struct lsf_slab_layout {
    int32_t nz, ny, nx;       /* local array (owned slab + halos) */
    int32_t z_begin, z_end;   /* owned slices */
    int32_t halo;
    int32_t lo_rank, hi_rank; /* neighbour ranks, -1 = none (end of the volume) */
} lsf_slab_layout;
typedef struct lsf_slab_part {
    lsf_grid grid;             /* z-range of this part */
    const int32_t *band_list[2]; /* NULL = dense walk */
    int64_t band_count[2];
    int32_t band_subset[2];
    int32_t n_lists;           /* 1 or 2 launches (INTERIOR + BOUNDARY band lists) */
    int32_t reserved;
} lsf_slab_part;
/* compact faces (optional): only the band voxels of the boundary slices travel -- every other voxel of a slab face
 * never changes.  [0] = lower side, [1] = upper side.  send_list: ascending local voxel indices of the band voxels in the
 * `halo` owned slices next to that neighbour; recv_list: the same for this rank's halo slices on that side (the
 * neighbour's send list, shifted: both ranks derive the lists from identical initial data, the caller verifies that the
 * counts agree before the first exchange); send_msg / recv_msg: device buffers of 4 * count floats. */
typedef struct lsf_slab_faces {
    const int32_t *send_list[2], *recv_list[2];
    float *send_msg[2], *recv_msg[2];
    int64_t send_count[2], recv_count[2];
} lsf_slab_faces;
/* helper of the compact-face plan: out[f] = the ascending merge of the ascending runs a[f] (n_a[f] entries) and b[f]
 * (n_b[f]), all entries distinct, for f < pairs <= 4, in ONE launch (a face's band voxels are the face slices' entries
 * of the INTERIOR and of the BOUNDARY list: both ranks must enumerate them in the same -- ascending -- order). */
int lsf_merge_sorted_runs(const int32_t *const *a, const int64_t *n_a, const int32_t *const *b, const int64_t *n_b,
                          int32_t *const *out, int32_t pairs, void *stream);
int lsf_slab_unique_id(const char *rccl_library_path, uint8_t *id_out128);
int lsf_slab_comm_create(const char *rccl_library_path, const uint8_t *id128, int32_t rank, int32_t world,
                         lsf_slab_comm **out);
int lsf_slab_comm_destroy(lsf_slab_comm *comm);
/* what RCCL itself says about the communicator: ncclCommUserRank and ncclCommCount (bench.py puts them on its N > 1 line) */
int lsf_slab_comm_info(lsf_slab_comm *comm, int32_t *rank_out, int32_t *nranks_out);
/* The cross-check of a call's compact faces ("the caller verifies that the counts agree"), without a host round trip in
 * front of the launches: _begin hands this rank's four counts (send lower, send upper, recv lower, recv upper) to an
 * ncclAllGather on the communicator's stream and returns at once; _end waits for it and copies every rank's four counts,
 * rank-major, into table[4 * world].  Every rank sees every row and so reaches the same verdict.  Every rank of the
 * communicator must call the pair at the same point of its call sequence (it is a collective). */
int lsf_slab_face_counts_begin(lsf_slab_comm *comm, const int64_t *counts4);
int lsf_slab_face_counts_end(lsf_slab_comm *comm, int64_t *table);
int lsf_slab_state_iteration(lsf_slab_comm *comm, const float *state_in, const float *canonical, float *state_out,
                             const lsf_slab_layout *layout, const lsf_slab_part *boundary_parts, int32_t n_boundary,
                             const lsf_slab_part *interior_parts, int32_t n_interior,
                             const lsf_slavcheva_params *params, const lsf_gate *gate, lsf_iteration_record *record,
                             int32_t exchange /* LSF_SLAB_* */,
                             const lsf_slab_faces *faces /* NULL: whole slices travel */, void *stream);

/* ---- a whole fixed-count call of a z-SLAB rank, enqueued by the library in two host calls (round 6) ---------------------
 * lsf_state_run_begin / _finish for one rank of a z-slab run: the loop of slavcheva_optimizer2d.py:354-388 over this
 * rank's slab, with the schedule the calls above implement one iteration at a time -- exchange groups of `halo`
 * iterations (iteration j of a group runs over the owned range widened by halo - 1 - j slices on every interior side;
 * the group's last iteration computes the boundary slices first, sends the faces on the communicator's stream while the
 * interior runs, and leaves the wait to the next iteration, LSF_SLAB_EXCHANGE_DEFERRED / LSF_SLAB_RESUME), compact faces
 * cross-checked with the neighbours (lsf_slab_face_counts_*; whole slices when a neighbour disagrees), the listed
 * finalize pass, and ONE gather of every rank's record words (ncclAllGather) so that all ranks decode the same records.
 * Caller shape: one rank of run_hierarchical_optimizer3d_multipair.py:403-432's loop, the volume cut along z.
 *   base: as lsf_state_run for the LOCAL array (owned slices + halos, grid.z_begin = 0, z_end = nz; z_global_offset as the
 *         slab's); slices must be a multiple of 1024 voxels (the cut positions are per-chunk prefix counts of the
 *         counting pass); box_scratch must be NULL; totals_device / totals_host hold 5 + 2 * LSF_SLAB_MAX_CUTS int64.
 *   lsf_slab_run_begin: launches the counting pass and the states' initialisation and RETURNS when the list sizes and the
 *         positions of the schedule's z cuts are on the host; it fills the out members: the caller allocates
 *         index_scratch (out_index_entries int32) and face_messages (4 * out_face_entries + 16 floats; NULL = whole faces).
 *   lsf_slab_run_finish: list fills, `iterations` iterations, lsf_state_finalize_listed into live_out (which holds the
 *         input live field), the record gather; RETURNS when both streams have drained, the records of ALL ranks decoded
 *         into `result` (energies summed over the ranks' owned slices, maxima over everything computed).  The caller
 *         checks the maxima: an update of one voxel or more inside an exchange group has read invalid halo slices
 *          -- every rank sees the same numbers and repeats the call on a wider slab.
 *         words_device: (1 + world) * iterations * LSF_RECORD_SLOTS * 4 int64; words_host (page-locked): world * ... . */
#define LSF_SLAB_MAX_CUTS 72
typedef struct lsf_slab_run {
    lsf_state_run base;
    lsf_slab_layout layout;
    int32_t exchange_interval;    /* iterations per exchange: layout.halo (exchange groups) or 1 */
    int32_t n_cuts;               /* out: the slices at which the band lists are cut, ascending ... */
    int32_t cut_slices[LSF_SLAB_MAX_CUTS];
    int64_t cut_entries[2][LSF_SLAB_MAX_CUTS]; /* out: ... and the number of INTERIOR / BOUNDARY entries in front of each */
    int64_t out_index_entries;    /* out: int32 elements of index_scratch lsf_slab_run_finish needs */
    int64_t out_face_entries;     /* out: band voxels of the four faces (send lower, send upper, recv lower, recv upper) */
} lsf_slab_run;
int lsf_slab_run_begin(lsf_slab_run *run, void *stream);
int lsf_slab_run_finish(const lsf_slab_run *run, lsf_slab_comm *comm, const lsf_slavcheva_params *params,
                        int32_t *list_interior, int32_t *list_boundary, int32_t *index_scratch, float *face_messages,
                        lsf_iteration_record *records, int32_t iterations, float *live_out, int64_t *words_device,
                        int64_t *words_host, lsf_state_run_result *result, void *stream);

/* the LAST pass of the zero-preserving filter at the voxels of a band list (axis = 2 in 3-D, 0 in 2-D:
 * math_utils/convolution.py:94-111,114-132; 3 / 5 / 7 / 9 taps) fused with lsf_slavcheva_update_rewarp
 * (slavcheva_optimizer2d.py:208-236): in_planar = the output of the passes before, zero_mask_source = the RAW gradient,
 * g_out_planar receives the final gradient; results are those of lsf_convolve_axis_listed + lsf_slavcheva_update_rewarp. */
int lsf_slavcheva_filter_update_rewarp(const float *in_planar, const float *zero_mask_source, const float *live,
                                       float *g_out_planar, float *warp_out_planar, float *live_out,
                                       const lsf_grid *grid, const lsf_slavcheva_params *params, int32_t axis,
                                       const double *taps_host, int32_t n_taps, const lsf_gate *gate,
                                       lsf_iteration_record *record, const int32_t *band_list, int64_t band_count,
                                       void *stream);

int lsf_slavcheva_update_rewarp(const float *live, const float *canonical, float *g_planar /* inout */,
                                float *warp_out_planar, float *live_out, const lsf_grid *grid,
                                const lsf_slavcheva_params *params, const lsf_gate *gate,
                                lsf_iteration_record *record, const int32_t *band_list /* may be NULL */,
                                int64_t band_count, void *stream);

/* ---- a20: convergence statistics ---------------------------------------------------------------------
 * replaces cpp.build_warp_delta_statistics_2d / build_tsdf_difference_statistics_2d
 * (slavcheva_optimizer2d.py:394-398; known answer tests/test_slavcheva_optimizer.py:141-145).
 * out (device, 8 doubles each):
 *   warp:  [count_band, count_above_lo, max_len, sum_len, sum_len^2, argmax_linear_index, 0, scratch]
 *   tsdf:  [count, min, max, sum, sum^2, argmax_linear_index, 0, scratch]    of |canonical - live|   */
int lsf_warp_statistics(const float *warp_planar, const float *canonical, const float *live,
                        const lsf_grid *grid, float lower_threshold, double *out8, void *stream);
int lsf_tsdf_difference_statistics(const float *canonical, const float *live, const lsf_grid *grid,
                                   double *out8, void *stream);

/* ---- a21: TSDF from a depth image, nearest pixel (the input stage of the path) ------------------------
 * replaces tsdf/generation.py:130-207 (generate_2d_tsdf_field_from_depth_image_no_interpolation: grid dims 2,
 * field[y][x] with the field's y index as depth axis and y_voxel = 0, depth row image_y_coordinate) and
 * :356-437 (generate_3d_tsdf_field_from_depth_image: grid dims 3, field[z][y][x]); tsdf/common.py:34-47.
 * depth_image: DEVICE uint16 [image_height][image_width]. */
typedef struct lsf_tsdf_params {
    double intrinsics[4];        /* fx, fy, cx, cy */
    double depth_unit_ratio;     /* metres per depth unit */
    double voxel_size;           /* metres */
    double narrow_band_half_width; /* metres: narrow_band_width_voxels / 2 * voxel_size */
    float extrinsic[16];         /* camera_extrinsic_matrix, row-major 4x4, float32 */
    int32_t array_offset[3];     /* voxels (x, y, z) */
    int32_t image_width, image_height;
    int32_t image_y_coordinate;  /* 2-D only */
    float default_value;
    int32_t intrinsics_are_f32;  /* project in float32 (float32 intrinsic matrix) or float64 */
} lsf_tsdf_params;

int lsf_tsdf_generate_nearest(const uint16_t *depth_image, float *field, const lsf_grid *grid,
                              const lsf_tsdf_params *params, void *stream);

/* the two bilinear 2-D variants: replaces tsdf/generation.py:78-128 (generate_2d_tsdf_field_from_depth_image_bilinear_
 * image_space, method 1: blend the depth of the two pixels around the projection, one TSDF value) and :18-75
 * (…_bilinear_tsdf_space, method 2 = FilteringMethod.BILINEAR_VOXEL_SPACE: a TSDF value per pixel, then blend); grid
 * dims 2 only, out-of-image taps read 1 as in utils/sampling.py:35-55. */
int lsf_tsdf_generate_bilinear(const uint16_t *depth_image, float *field, const lsf_grid *grid,
                               const lsf_tsdf_params *params, int32_t method, void *stream);

/* EWA filters: replaces tsdf/ewa.py:230-353 (generate_tsdf_2d_ewa_image, method 3), :358-481 (…_ewa_tsdf, method 4),
 * :485-624 (…_ewa_tsdf_inclusive, method 5) for grid dims 2, and :59-184 (generate_tsdf_3d_ewa_image, method 3) for
 * grid dims 3 -- there field[a][b][c] has world x on array axis 0 and the depth axis on array axis 2 (the reference's
 * deliberate flip, tsdf/ewa.py:115-119).  params->intrinsics is ignored; the full matrix travels in ewa. */
typedef struct lsf_ewa_params {
    double covariance_camera_space[9]; /* R * (gaussian_covariance_scale*voxel_size*I) * R^T, row-major, float64 */
    double squared_radius_threshold;   /* 4 * gaussian_covariance_scale * voxel_size */
    float intrinsic_matrix[9];         /* row-major 3x3, float32 */
    int32_t method;                    /* 3 EWA_IMAGE_SPACE, 4 EWA_VOXEL_SPACE, 5 EWA_VOXEL_SPACE_INCLUSIVE */
} lsf_ewa_params;

int lsf_tsdf_generate_ewa(const uint16_t *depth_image, float *field, const lsf_grid *grid,
                          const lsf_tsdf_params *params, const lsf_ewa_params *ewa, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LSF_HIP_H */
