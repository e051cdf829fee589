// A whole KillingFusion-style optimize() call of a whole volume -- a fixed iteration count or the reference's default
// threshold-terminated loop --, enqueued by the LIBRARY in two host calls
// (include/lsf_hip.h: lsf_state_run_begin / lsf_state_run_finish).  Reference loop: nonrigid_opt/slavcheva/
// slavcheva_optimizer2d.py:354-388 -- one Python iteration of which is ONE kernel launch here; caller shape
// run_hierarchical_optimizer3d_multipair.py:403-432 (a loop of such calls over independent pairs).
//
// Nothing new runs on the device: the two functions issue the launches of lsf_state_prepare, lsf_state_pack_needed /
// lsf_state_pack, lsf_band_list_fill_prepared, lsf_slavcheva_state_iteration x K and lsf_state_finalize_listed in the order
// the Python engine issues them, on the caller's stream -- results are those launches' results, bit for bit.  What changes
// is the host side: K + 8 foreign calls, their argument marshalling and three tensor-level read-backs become two calls
// that run without the interpreter lock (ctypes releases it), so that a second pair's call can be enqueued by another
// host thread meanwhile (experiment/multipair.run_pairs with several optimizers).
#include "lsf_device.h"

#include <cstring>

using namespace lsf;

namespace {

// the four used words of every slot of every record (a record is 8 slots 4 KiB apart) and, behind them, the 16 doubles of
// the convergence statistics, gathered for ONE copy to the host
__global__ __launch_bounds__(kBlock) void records_used_words_kernel(const long long* __restrict__ records,
                                                                    const long long* __restrict__ statistics16,
                                                                    long long* __restrict__ out, int n_slots, int tail) {
    const int k = blockIdx.x * kBlock + threadIdx.x;  // (slot, word), then (tail != 0) the statistics
    if (k < n_slots * 4) out[k] = records[(long long)(k >> 2) * (sizeof(lsf_record_slot) / 8) + (k & 3)];
    else if (tail && k < n_slots * 4 + 16) out[k] = statistics16 ? statistics16[k - n_slots * 4] : 0ll;
}

// one timing-less event per host thread and device, created on first use (an event per call costs a create / destroy pair)
inline hipEvent_t thread_event() {
    thread_local hipEvent_t events[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!events[dev] && hipEventCreateWithFlags(&events[dev], hipEventDisableTiming) != hipSuccess) events[dev] = nullptr;
    return events[dev];
}

inline bool run_ok(const lsf_state_run* r) {
    return r && r->live && r->canonical && r->state[0] && r->state[1] && r->state[0] != r->state[1] && r->prepare_scratch &&
           r->totals_device && r->totals_host && r->sparse_reach >= 0 && r->sparse_reach <= 8;
}

}  // namespace

extern "C" int lsf_state_run_begin(const lsf_state_run* run, void* stream) {
    if (!run_ok(run)) return LSF_ERR_BAD_ARGUMENT;
    const lsf_grid* g = &run->grid;
    if (int e = check_grid(g)) return e;
    if (g->z_begin != 0 || g->z_end != g->nz) return LSF_ERR_BAD_ARGUMENT;  // whole volumes
    hipStream_t s = as_stream(stream);
    const bool sparse = run->sparse_reach > 0;
    // the counting pass; it also writes the states unless they are initialised near the band only (below).  With
    // second_state_late the second state is written behind the copy of the list sizes, where it overlaps the host's wait
    float* b_in_pass = (sparse || run->second_state_late) ? nullptr : run->state[1];
    if (int e = lsf_state_prepare(run->live, run->canonical, sparse ? nullptr : run->state[0], b_in_pass, g,
                                  run->prepare_scratch, run->totals_device, stream))
        return e;
    // (optional) the boxes of the box walk are counted behind the pass, so that their number comes back with the list sizes
    if (run->box_scratch) {
        if (int e = lsf_band_boxes_count(g, run->box_all ? LSF_BAND_ALL : LSF_BAND_INTERIOR, run->prepare_scratch, run->box_scratch,
                                         run->totals_device + 4, stream))
            return e;
    } else if (hipMemsetAsync(run->totals_device + 4, 0, sizeof(int64_t), s) != hipSuccess) {
        return (int)hipGetLastError();
    }
    if (hipMemcpyAsync(run->totals_host, run->totals_device, 5 * sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess)
        return (int)hipGetLastError();
    hipEvent_t sizes = thread_event();
    if (!sizes) return (int)hipGetLastError();
    int status = 0;
    if (hipEventRecord(sizes, s) != hipSuccess) status = (int)hipGetLastError();
    if (!status && sparse)
        status = lsf_state_pack_needed(run->live, run->state[0], run->state[1], g, run->prepare_scratch, run->sparse_reach, 0,
                                       stream);
    else if (!status && run->second_state_late)
        status = lsf_state_pack(run->live, nullptr, run->state[1], nullptr, g, stream);
    if (!status && hipEventSynchronize(sizes) != hipSuccess) status = (int)hipGetLastError();
    return status;
}

extern "C" int lsf_state_run_finish(const lsf_state_run* run, const lsf_slavcheva_params* params, int32_t* list_interior,
                                    int32_t* list_boundary, lsf_band_box* boxes, float* box_canonical,
                                    lsf_iteration_record* records, int32_t iterations, const lsf_run_loop* loop,
                                    float* live_out, float lower_threshold, double* statistics16,
                                    double* finalize_scratch, int64_t* words_device, int64_t* words_host,
                                    lsf_state_run_result* result, void* stream) {
    if (!run_ok(run) || !params || !records || iterations < 1 || !live_out || !words_device || !words_host || !result ||
        (statistics16 && !finalize_scratch) || !result->max_value || !result->argmax ||
        !result->energies3 || !result->executed)
        return LSF_ERR_BAD_ARGUMENT;
    if (loop && (loop->min_iterations < 1 || loop->check_interval < 1 ||
                 (loop->max_iterations > loop->min_iterations ? loop->max_iterations : loop->min_iterations) != iterations))
        return LSF_ERR_BAD_ARGUMENT;
    if (run->box_all && boxes) return LSF_ERR_BAD_ARGUMENT;  // those are lsf_sobolev_run_finish's boxes
    const lsf_grid* g = &run->grid;
    if (int e = check_grid(g)) return e;
    const int64_t n_interior = run->totals_host[0], n_boundary = run->totals_host[1];
    if (n_interior < 0 || n_boundary < 0 || n_interior > 0x7fffffffll || n_boundary > 0x7fffffffll ||
        (n_interior && !list_interior) || (n_boundary && !list_boundary))
        return LSF_ERR_BAD_ARGUMENT;
    hipStream_t s = as_stream(stream);
    // the lists, from the ballots the counting pass kept; an empty list is dropped -- but never both (a launch over an
    // empty BOUNDARY list still contributes the arg-max of an all-zero update)
    const int32_t* lists[2];
    int64_t counts[2];
    int32_t subsets[2];
    int n_lists = 0;
    if (n_interior) {
        if (int e = lsf_band_list_fill_prepared(g, LSF_BAND_INTERIOR, run->prepare_scratch, list_interior, stream)) return e;
        lists[n_lists] = list_interior; counts[n_lists] = n_interior; subsets[n_lists++] = LSF_BAND_INTERIOR;
    }
    if (n_boundary || !n_lists) {
        if (n_boundary)
            if (int e = lsf_band_list_fill_prepared(g, LSF_BAND_BOUNDARY, run->prepare_scratch, list_boundary, stream)) return e;
        // (an empty list is never dereferenced; the pointer only has to be non-NULL to mean "a list walk")
        lists[n_lists] = n_boundary ? list_boundary : reinterpret_cast<const int32_t*>(run->prepare_scratch);
        counts[n_lists] = n_boundary; subsets[n_lists++] = LSF_BAND_BOUNDARY;
    }
    // the INTERIOR voxels box by box when the caller asks for it (same results; DESIGN.md section 5: the walk that pays
    // when the band's states do not fit the Infinity Cache)
    const int64_t n_boxes = boxes && run->box_scratch ? run->totals_host[4] : 0;
    if (n_boxes > 0) {
        if (!box_canonical) return LSF_ERR_BAD_ARGUMENT;
        if (int e = lsf_band_boxes_fill(g, LSF_BAND_INTERIOR, run->prepare_scratch, run->box_scratch, boxes, stream)) return e;
        if (int e = lsf_band_boxes_canonical(run->canonical, g, boxes, n_boxes, box_canonical, stream)) return e;
    }
    const bool sparse = run->sparse_reach > 0;
    const int n_slots = iterations * LSF_RECORD_SLOTS;
    // The iterations: iteration i reads state[i % 2] and writes the other.  A fixed count: all of them ungated, at once.
    // With a stop test that can fire (slavcheva_optimizer2d.py:360-362) iteration i >= min_iterations sits behind the
    // device-side gate on record i - 1, and the host looks at the records every check_interval iterations.
    const bool fixed = !loop || loop->min_iterations >= iterations;
    int32_t it = 0, n_exec = fixed ? iterations : 0;
    while (it < iterations) {
        const int32_t batch = fixed ? iterations : (loop->check_interval < iterations - it ? loop->check_interval : iterations - it);
        for (int32_t i = it; i < it + batch; ++i) {
            lsf_gate gate_i{records + (i > 0 ? i - 1 : 0), LSF_GATE_SLAVCHEVA, loop ? loop->lower_threshold : 0.0f,
                            loop ? loop->upper_threshold : 0.0f};
            const lsf_gate* gate = (fixed || i < loop->min_iterations) ? nullptr : &gate_i;
            for (int k = 0; k < n_lists; ++k) {
                int e;
                if (n_boxes > 0 && subsets[k] == LSF_BAND_INTERIOR)
                    e = lsf_slavcheva_state_iteration_boxes(run->state[i % 2], box_canonical, run->state[(i + 1) % 2], g, params,
                                                            gate, records + i, boxes, n_boxes, stream);
                else
                    e = lsf_slavcheva_state_iteration(run->state[i % 2], run->canonical, run->state[(i + 1) % 2], g, params,
                                                      gate, records + i, lists[k], counts[k], subsets[k], stream);
                if (e) return e;
            }
        }
        it += batch;
        if (fixed) break;
        // the batch's records (their used words) to the host, one wait; then the reference's loop condition
        const int first_slot = (it - batch) * LSF_RECORD_SLOTS, batch_slots = batch * LSF_RECORD_SLOTS;
        hipLaunchKernelGGL(records_used_words_kernel, dim3((batch_slots * 4 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                           reinterpret_cast<const long long*>(records + (it - batch)), nullptr,
                           reinterpret_cast<long long*>(words_device) + (size_t)first_slot * 4, batch_slots, 0);
        if (int e = launch_status()) return e;
        if (hipMemcpyAsync(words_host + (size_t)first_slot * 4, words_device + (size_t)first_slot * 4,
                           (size_t)batch_slots * 4 * sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess)
            return (int)hipGetLastError();
        if (hipStreamSynchronize(s) != hipSuccess) return (int)hipGetLastError();
        if (int e = lsf_records_decode(words_host, it, LSF_RECORD_SLOTS, 4, result->max_value, result->argmax,
                                       result->energies3, result->executed))
            return e;
        n_exec = 0;
        while (n_exec < it && result->executed[n_exec]) ++n_exec;
        if (n_exec < it) break;  // the gate closed inside the batch
        const float m = result->max_value[n_exec - 1];
        if (n_exec >= loop->min_iterations && !(loop->lower_threshold < m && m < loop->upper_threshold)) break;
    }
    // the end of the call behind the last iteration: the listed voxels' live values into the caller's array (which holds
    // the input everywhere else) and the convergence statistics; with sparsely initialised states the pass looks at the
    // records first and leaves everything alone when an update outran what was initialised
    const float* final_state = run->state[n_exec % 2];
    int64_t listed_counts[2] = {0, 0};
    const int32_t* listed[2] = {nullptr, nullptr};
    int n_listed = 0;
    for (int k = 0; k < n_lists; ++k)
        if (counts[k]) { listed[n_listed] = lists[k]; listed_counts[n_listed++] = counts[k]; }
    if (int e = lsf_state_finalize_listed(final_state, run->canonical, live_out, nullptr, g, listed, listed_counts, n_listed,
                                          run->totals_host[2], run->totals_host[3], lower_threshold, statistics16,
                                          finalize_scratch, nullptr, sparse ? records : nullptr, sparse ? iterations : 0,
                                          (float)run->sparse_reach, stream))
        return e;
    hipLaunchKernelGGL(records_used_words_kernel, dim3((n_slots * 4 + 16 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                       reinterpret_cast<const long long*>(records), reinterpret_cast<const long long*>(statistics16),
                       reinterpret_cast<long long*>(words_device), n_slots, 1);
    if (int e = launch_status()) return e;
    if (hipMemcpyAsync(words_host, words_device, ((size_t)n_slots * 4 + 16) * sizeof(int64_t), hipMemcpyDeviceToHost, s) !=
        hipSuccess)
        return (int)hipGetLastError();
    if (hipStreamSynchronize(s) != hipSuccess) return (int)hipGetLastError();
    if (int e = lsf_records_decode(words_host, iterations, LSF_RECORD_SLOTS, 4, result->max_value, result->argmax,
                                   result->energies3, result->executed))
        return e;
    result->final_state = n_exec % 2;
    result->n_lists = n_lists;
    result->reach_exceeded = 0;
    result->compact_faces = -1;
    for (int32_t i = 0; i < iterations && sparse; ++i)
        if (result->executed[i] && !(result->max_value[i] < (float)run->sparse_reach)) result->reach_exceeded = 1;
    return 0;
}

// the same call with the SobolevFusion iteration (gradient + x pass over the list of all band voxels, then y pass, z pass,
// update and re-warp box by box): engine_run._optimize_run(sobolev=True)
extern "C" int lsf_sobolev_run_finish(const lsf_state_run* run, const lsf_slavcheva_params* params, const double* taps_host,
                                      int32_t n_taps, int32_t* list_interior, int32_t* list_boundary, int32_t* list_all,
                                      lsf_band_box* boxes, float* g4_a, float* g4_b, lsf_iteration_record* records,
                                      int32_t iterations, const lsf_run_loop* loop, float* live_out, float lower_threshold,
                                      double* statistics16, double* finalize_scratch, int64_t* words_device,
                                      int64_t* words_host, lsf_state_run_result* result, void* stream) {
    if (!run_ok(run) || !run->box_scratch || !run->box_all || !params || !taps_host || !boxes || !g4_a || !g4_b ||
        g4_a == g4_b || !records || iterations < 1 || !live_out || !words_device || !words_host || !result ||
        (statistics16 && !finalize_scratch) || !result->max_value || !result->argmax || !result->energies3 ||
        !result->executed)
        return LSF_ERR_BAD_ARGUMENT;
    if (loop && (loop->min_iterations < 1 || loop->check_interval < 1 ||
                 (loop->max_iterations > loop->min_iterations ? loop->max_iterations : loop->min_iterations) != iterations))
        return LSF_ERR_BAD_ARGUMENT;
    const lsf_grid* g = &run->grid;
    if (int e = check_grid(g)) return e;
    if (g->dims != 3) return LSF_ERR_BAD_DIMS;
    const int64_t n_interior = run->totals_host[0], n_boundary = run->totals_host[1], n_boxes = run->totals_host[4];
    if (n_interior < 0 || n_boundary < 0 || n_boxes < 0 || n_interior + n_boundary > 0x7fffffffll ||
        (n_interior && !list_interior) || (n_boundary && !list_boundary) || (n_interior && n_boundary && !list_all))
        return LSF_ERR_BAD_ARGUMENT;
    hipStream_t s = as_stream(stream);
    if (n_interior)
        if (int e = lsf_band_list_fill_prepared(g, LSF_BAND_INTERIOR, run->prepare_scratch, list_interior, stream)) return e;
    if (n_boundary)
        if (int e = lsf_band_list_fill_prepared(g, LSF_BAND_BOUNDARY, run->prepare_scratch, list_boundary, stream)) return e;
    // ONE ascending list of the whole band for the fused gradient + x pass (a tap at a voxel of another list would count as zero)
    const int32_t* all = n_interior ? list_interior : list_boundary;
    const int64_t n_all = n_interior + n_boundary;
    if (n_interior && n_boundary) {
        const int32_t* a[1] = {list_interior};
        const int32_t* b[1] = {list_boundary};
        int32_t* out[1] = {list_all};
        const int64_t na[1] = {n_interior}, nb[1] = {n_boundary};
        if (int e = lsf_merge_sorted_runs(a, na, b, nb, out, 1, stream)) return e;
        all = list_all;
    }
    if (n_boxes > 0)
        if (int e = lsf_band_boxes_fill(g, LSF_BAND_ALL, run->prepare_scratch, run->box_scratch, boxes, stream)) return e;
    const bool sparse = run->sparse_reach > 0;
    const int n_slots = iterations * LSF_RECORD_SLOTS;
    const bool fixed = !loop || loop->min_iterations >= iterations;
    int32_t it = 0, n_exec = fixed ? iterations : 0;
    while (it < iterations) {
        const int32_t batch = fixed ? iterations : (loop->check_interval < iterations - it ? loop->check_interval : iterations - it);
        for (int32_t i = it; i < it + batch; ++i) {
            lsf_gate gate_i{records + (i > 0 ? i - 1 : 0), LSF_GATE_SLAVCHEVA, loop ? loop->lower_threshold : 0.0f,
                            loop ? loop->upper_threshold : 0.0f};
            const lsf_gate* gate = (fixed || i < loop->min_iterations) ? nullptr : &gate_i;
            const float* s_in = run->state[i % 2];
            float* s_out = run->state[(i + 1) % 2];
            if (n_all)
                if (int e = lsf_sobolev_state_gradient_x(s_in, run->canonical, g4_a, g, params, taps_host, n_taps, gate,
                                                         records + i, all, n_all, 1, stream))
                    return e;
            // the filtered gradient is an OUTPUT of the last executed iteration only: a fixed count stores just that one
            float* g_out = (!fixed || i == iterations - 1) ? g4_b : nullptr;
            if (int e = lsf_sobolev_state_update_boxes(g4_a, s_in, s_out, g_out, g, params, taps_host, n_taps, gate, records + i,
                                                       boxes, n_boxes, stream))
                return e;
        }
        it += batch;
        if (fixed) break;
        const int first_slot = (it - batch) * LSF_RECORD_SLOTS, batch_slots = batch * LSF_RECORD_SLOTS;
        hipLaunchKernelGGL(records_used_words_kernel, dim3((batch_slots * 4 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                           reinterpret_cast<const long long*>(records + (it - batch)), nullptr,
                           reinterpret_cast<long long*>(words_device) + (size_t)first_slot * 4, batch_slots, 0);
        if (int e = launch_status()) return e;
        if (hipMemcpyAsync(words_host + (size_t)first_slot * 4, words_device + (size_t)first_slot * 4,
                           (size_t)batch_slots * 4 * sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess)
            return (int)hipGetLastError();
        if (hipStreamSynchronize(s) != hipSuccess) return (int)hipGetLastError();
        if (int e = lsf_records_decode(words_host, it, LSF_RECORD_SLOTS, 4, result->max_value, result->argmax,
                                       result->energies3, result->executed))
            return e;
        n_exec = 0;
        while (n_exec < it && result->executed[n_exec]) ++n_exec;
        if (n_exec < it) break;
        const float m = result->max_value[n_exec - 1];
        if (n_exec >= loop->min_iterations && !(loop->lower_threshold < m && m < loop->upper_threshold)) break;
    }
    const int32_t* listed[2] = {nullptr, nullptr};
    int64_t listed_counts[2] = {0, 0};
    int n_listed = 0;
    if (n_interior) { listed[n_listed] = list_interior; listed_counts[n_listed++] = n_interior; }
    if (n_boundary) { listed[n_listed] = list_boundary; listed_counts[n_listed++] = n_boundary; }
    if (int e = lsf_state_finalize_listed(run->state[n_exec % 2], run->canonical, live_out, nullptr, g, listed, listed_counts,
                                          n_listed, run->totals_host[2], run->totals_host[3], lower_threshold, statistics16,
                                          finalize_scratch, nullptr, sparse ? records : nullptr, sparse ? iterations : 0,
                                          (float)run->sparse_reach, stream))
        return e;
    hipLaunchKernelGGL(records_used_words_kernel, dim3((n_slots * 4 + 16 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                       reinterpret_cast<const long long*>(records), reinterpret_cast<const long long*>(statistics16),
                       reinterpret_cast<long long*>(words_device), n_slots, 1);
    if (int e = launch_status()) return e;
    if (hipMemcpyAsync(words_host, words_device, ((size_t)n_slots * 4 + 16) * sizeof(int64_t), hipMemcpyDeviceToHost, s) !=
        hipSuccess)
        return (int)hipGetLastError();
    if (hipStreamSynchronize(s) != hipSuccess) return (int)hipGetLastError();
    if (int e = lsf_records_decode(words_host, iterations, LSF_RECORD_SLOTS, 4, result->max_value, result->argmax,
                                   result->energies3, result->executed))
        return e;
    result->final_state = n_exec % 2;
    result->n_lists = n_listed;
    result->reach_exceeded = 0;
    result->compact_faces = -1;
    for (int32_t i = 0; i < iterations && sparse; ++i)
        if (result->executed[i] && !(result->max_value[i] < (float)run->sparse_reach)) result->reach_exceeded = 1;
    return 0;
}
