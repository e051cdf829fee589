/*
 * lsf_hip_chain.h -- C ABI of the OPTIONAL add-on library liblsf_chain.so: the chain kernel, K fused iterations of the
 * KillingFusion warp update per launch.  Measured 4 % SLOWER than one launch per iteration (DESIGN.md section 5 and 7,
 * round 3), so the product library liblsf_hip.so does not carry it; the add-on is built by
 * levelsetfusion-python_amd/_build.py::build_chain (also from __graft_entry__.build), loaded only when LSF_CHAIN=1 asks
 * for it (levelsetfusion-python_amd/_lib.py::chain_lib) and exercised by tests/test_gpu_chain.py.  Types and error codes
 * are those of lsf_hip.h.
 */
#ifndef LSF_HIP_CHAIN_H
#define LSF_HIP_CHAIN_H

#include "lsf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- K fused iterations in ONE launch: the chain kernel ------------------------------------------
 * replaces K consecutive passes of the loop body nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330 (DIRECT; the
 * VECTORIZED form :163-236 with its parameter block) for runs whose stop test (:360-362) cannot fire in between: iteration
 * j = 0 .. iterations - 1 reads state_a (j even) or state_b (j odd), writes the other and reduces into records[j], exactly
 * as lsf_slavcheva_state_iteration would in K launches -- bit for bit while every warp update stays below 2 voxels.
 * One CU-sized workgroup per CU stays resident; a workgroup waits only for the few neighbouring list chunks its stencils
 * and re-warp gather reach (progress words in `scratch`), never for the whole chip.  Requirements: an INTERIOR band list
 * of the WHOLE array (z_begin = 0, z_end = nz; no BOUNDARY voxels besides it), 16 * nz * ny * nx < 2^32.
 *   scratch   lsf_state_chain_scratch_elements(band_count, stages) int32 of device memory (16-byte aligned);
 *             lsf_state_chain_plan fills its dependency windows ONCE per list, every lsf_slavcheva_state_chain call on
 *             that list (same band_count and stages) reuses them and zeroes the words it polls.
 *   stages    1: every CU owns one chunk and runs all iterations on it.  S > 1 (long lists only, else treated as 1): the
 *             CUs form S groups, group s runs iterations s, s + S, ... over all chunks, so that an iteration's output is
 *             consumed from the Infinity Cache by the next iteration instead of travelling through HBM.
 *   scratch[0] != 0 after the launch: a wait timed out (records[iterations - 1] then decodes to a NaN maximum);
 *   scratch[1] != 0: an update of 2 voxels or more -- the result is NOT the reference's; lsf_state_finalize_listed(...,
 *             skip_flag = scratch + 1) then leaves the caller's fields untouched and the caller repeats the call with
 *             lsf_slavcheva_state_iteration.  The same verdict follows from the records' maxima.
 * Returns LSF_ERR_NOT_RESIDENT (nothing launched) when a CU cannot hold one 1024-thread workgroup of the kernel. */
int64_t lsf_state_chain_scratch_elements(int64_t band_count, int32_t stages);
int lsf_state_chain_shape(int64_t band_count, int32_t stages, int32_t *out4 /* workgroups, stages, chunks, wave-units */);
int lsf_state_chain_plan(const lsf_grid *grid, const int32_t *band_list, int64_t band_count, int32_t stages,
                         int32_t *scratch, void *stream);
int lsf_slavcheva_state_chain(float *state_a, float *state_b, const float *canonical, const lsf_grid *grid,
                              const lsf_slavcheva_params *params, lsf_iteration_record *records,
                              const int32_t *band_list, int64_t band_count, int32_t iterations, int32_t stages,
                              int32_t *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LSF_HIP_CHAIN_H */
