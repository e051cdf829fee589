// z-slab runtime of the fused Slavcheva path for multi-GPU runs (one process per GPU): ONE host call enqueues a whole
// iteration -- boundary-slice launches, the halo exchange over RCCL (xGMI) on a second HIP stream, the interior
// launches -- instead of ~10 Python-level calls into torch.distributed (measured on one GPU with self send / recv,
// tools/slab_nccl_loopback.py: 150 us of host time per iteration through batch_isend_irecv against a 40 us kernel).
// The reference has no distributed code; this is new design, DESIGN.md section 6.
//
// RCCL is bound at run time (dlopen of the librccl.so the process already uses -- PyTorch ships its own copy), so
// liblsf_hip.so has no link-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>

#include "../../include/lsf_hip.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

// The one piece of process-wide state in the library: the table of RCCL entry points, bound ONCE per process (a mutex
// serialises the attempts; a failed attempt leaves the table empty so that a later call may name another path).  Once
// bound the table is read-only, so concurrent iteration calls on different communicators never race on it.
int load_rccl(const char* path) {
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    if (g_rccl.handle) return 0;
    RcclApi api;
    const char* candidates[] = {path, "librccl.so", "librccl.so.1"};
    for (const char* c : candidates) {
        if (!c || !c[0]) continue;
        // the copy already mapped into this process first (same soname), else load it
        void* h = dlopen(c, RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            api.handle = h;
            break;
        }
    }
    if (!api.handle) return LSF_ERR_RCCL_UNAVAILABLE;
    auto bind = [&](auto& fn, const char* name) {
        fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name));
        return fn != nullptr;
    };
    const bool ok = bind(api.GetUniqueId, "ncclGetUniqueId") && bind(api.CommInitRank, "ncclCommInitRank") &&
                    bind(api.CommDestroy, "ncclCommDestroy") && bind(api.GroupStart, "ncclGroupStart") &&
                    bind(api.GroupEnd, "ncclGroupEnd") && bind(api.Send, "ncclSend") &&
                    bind(api.Recv, "ncclRecv") && bind(api.AllGather, "ncclAllGather") &&
                    bind(api.GetErrorString, "ncclGetErrorString");
    if (!ok) return LSF_ERR_RCCL_UNAVAILABLE;
    g_rccl = api;  // the handle is stored last of all members' owner: readers test it first
    return 0;
}

}  // namespace

struct lsf_slab_comm {
    ncclComm_t comm;
    int rank, world;
    hipStream_t comm_stream;
    hipEvent_t boundary_done[2], halos_done[2];
    unsigned parity;
    int pending;  // index of the halos_done event a LSF_SLAB_EXCHANGE_DEFERRED call left for the next call, or -1
    // face counts of the current call (lsf_slab_face_counts_*): [0, 4) this rank's, [4, 4 + 4 * world) every rank's
    long long* counts_dev;
    long long* counts_host;  // pinned
    hipEvent_t counts_done;
    bool counts_in_flight;
    bool gather_on_main;  // the face gather behind the boundary launches on the launch stream (default), not on comm_stream
};

#define LSF_RCCL_CHECK(call)                                                                       \
    do {                                                                                           \
        const ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                                   \
            fprintf(stderr, "liblsf_hip: %s failed: %s\n", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return LSF_ERR_RCCL_FAILED;                                                            \
        }                                                                                          \
    } while (0)

#define LSF_HIP_CHECK(call)               \
    do {                                  \
        const hipError_t e_ = (call);     \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

extern "C" int lsf_slab_unique_id(const char* rccl_library_path, uint8_t* id_out128) {
    if (!id_out128) return LSF_ERR_BAD_ARGUMENT;
    if (int e = load_rccl(rccl_library_path)) return e;
    ncclUniqueId id;
    LSF_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id_out128, &id, sizeof(id));
    return 0;
}

extern "C" int lsf_slab_comm_create(const char* rccl_library_path, const uint8_t* id128, int32_t rank, int32_t world,
                                    lsf_slab_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return LSF_ERR_BAD_ARGUMENT;
    if (int e = load_rccl(rccl_library_path)) return e;
    lsf_slab_comm* c = new (std::nothrow) lsf_slab_comm();
    if (!c) return LSF_ERR_BAD_ARGUMENT;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    if (g_rccl.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) {
        delete c;
        return LSF_ERR_RCCL_FAILED;
    }
    c->rank = rank;
    c->world = world;
    c->parity = 0;
    c->pending = -1;
    c->counts_dev = c->counts_host = nullptr;
    c->counts_in_flight = false;
    // the face gather runs on the launch stream, right behind the boundary launches (on the communication stream it
    // started 13 us late and took 21 us beside the interior launch: 2.36-2.40 against 2.29-2.39 ms per slab call,
    // profiles/r04_slab_rccl_loopback.txt)
    c->gather_on_main = true;
    hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&c->boundary_done[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->halos_done[k], hipEventDisableTiming);
    }
    const size_t words = 4u + 4u * (size_t)world;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->counts_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->counts_dev), words * sizeof(long long));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->counts_host), words * sizeof(long long), 0);
    if (e != hipSuccess) {  // (the struct was value-initialised: whatever is still null was never created)
        if (c->counts_dev) (void)hipFree(c->counts_dev);
        if (c->counts_host) (void)hipHostFree(c->counts_host);
        if (c->counts_done) (void)hipEventDestroy(c->counts_done);
        for (int k = 0; k < 2; ++k) {
            if (c->boundary_done[k]) (void)hipEventDestroy(c->boundary_done[k]);
            if (c->halos_done[k]) (void)hipEventDestroy(c->halos_done[k]);
        }
        if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
        g_rccl.CommDestroy(c->comm);
        delete c;
        return (int)e;
    }
    *out = c;
    return 0;
}

extern "C" int lsf_slab_comm_destroy(lsf_slab_comm* c) {
    if (!c) return 0;
    (void)hipStreamSynchronize(c->comm_stream);
    for (int k = 0; k < 2; ++k) {
        (void)hipEventDestroy(c->boundary_done[k]);
        (void)hipEventDestroy(c->halos_done[k]);
    }
    (void)hipEventDestroy(c->counts_done);
    (void)hipFree(c->counts_dev);
    (void)hipHostFree(c->counts_host);
    (void)hipStreamDestroy(c->comm_stream);
    if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
    return 0;
}

extern "C" int lsf_slab_face_counts_begin(lsf_slab_comm* c, const int64_t* counts4) {
    if (!c || !counts4) return LSF_ERR_BAD_ARGUMENT;
    if (c->counts_in_flight) {
        // a call that planned its faces but never exchanged them (fewer iterations than one exchange group) never asked
        // for the table: every rank is in that position together, so the stale collective is simply waited for and dropped
        LSF_HIP_CHECK(hipEventSynchronize(c->counts_done));
        c->counts_in_flight = false;
    }
    for (int k = 0; k < 4; ++k) c->counts_host[k] = counts4[k];
    LSF_HIP_CHECK(hipMemcpyAsync(c->counts_dev, c->counts_host, 4 * sizeof(long long), hipMemcpyHostToDevice, c->comm_stream));
    LSF_RCCL_CHECK(g_rccl.AllGather(c->counts_dev, c->counts_dev + 4, 4, ncclInt64, c->comm, c->comm_stream));
    LSF_HIP_CHECK(hipMemcpyAsync(c->counts_host + 4, c->counts_dev + 4, 4 * (size_t)c->world * sizeof(long long),
                                 hipMemcpyDeviceToHost, c->comm_stream));
    LSF_HIP_CHECK(hipEventRecord(c->counts_done, c->comm_stream));
    c->counts_in_flight = true;
    return 0;
}

extern "C" int lsf_slab_face_counts_end(lsf_slab_comm* c, int64_t* table) {
    if (!c || !table || !c->counts_in_flight) return LSF_ERR_BAD_ARGUMENT;
    c->counts_in_flight = false;
    LSF_HIP_CHECK(hipEventSynchronize(c->counts_done));
    for (int k = 0; k < 4 * c->world; ++k) table[k] = c->counts_host[4 + k];
    return 0;
}

typedef float vf4 __attribute__((ext_vector_type(4)));

// compact faces: only the band voxels of the boundary / halo slices travel (everything else never changes)
__global__ __launch_bounds__(256) void face_gather_kernel(const vf4* __restrict__ state, const int* __restrict__ list_a,
                                                          unsigned n_a, vf4* __restrict__ msg_a,
                                                          const int* __restrict__ list_b, unsigned n_b,
                                                          vf4* __restrict__ msg_b) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_a) msg_a[k] = state[list_a[k]];
    else if (k - n_a < n_b) msg_b[k - n_a] = state[list_b[k - n_a]];
}

__global__ __launch_bounds__(256) void face_scatter_kernel(vf4* __restrict__ state, const int* __restrict__ list_a,
                                                           unsigned n_a, const vf4* __restrict__ msg_a,
                                                           const int* __restrict__ list_b, unsigned n_b,
                                                           const vf4* __restrict__ msg_b) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_a) state[list_a[k]] = msg_a[k];
    else if (k - n_a < n_b) state[list_b[k - n_a]] = msg_b[k - n_a];
}

// merge of two ascending runs of distinct voxel indices into one ascending run, several pairs per launch: element j of
// a run lands at j + (the number of elements of the other run below it) -- one binary search per element.  (The face
// lists are the union of the INTERIOR and the BOUNDARY list's entries of the face slices; torch.sort(torch.cat(...)) did
// this with ~15 small launches and ~0.1 ms of host time per face, in front of the second iteration.)
struct MergePairs {
    const int* a[4];
    const int* b[4];
    int* out[4];
    unsigned na[4], nb[4], first[5];
};

__device__ inline unsigned count_below(const int* __restrict__ run, unsigned n, int v) {
    unsigned lo = 0u, hi = n;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (run[mid] < v) lo = mid + 1u;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void merge_runs_kernel(MergePairs m, int pairs) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m.first[pairs]) return;
    int f = 0;
    while (f + 1 < pairs && k >= m.first[f + 1]) ++f;
    unsigned j = k - m.first[f];
    if (j < m.na[f]) {
        const int v = m.a[f][j];
        m.out[f][j + count_below(m.b[f], m.nb[f], v)] = v;
    } else {
        j -= m.na[f];
        const int v = m.b[f][j];
        m.out[f][j + count_below(m.a[f], m.na[f], v)] = v;
    }
}

// Send order: lower boundary -> lower neighbour, upper halo <- upper neighbour, upper boundary -> upper neighbour, lower
// halo <- lower neighbour.  Between distinct peers the order inside a group is irrelevant; when a rank is its own
// neighbour (the one-GPU loop-back of a z-periodic stack) sends and receives pair up in order, and this order pairs the
// lower boundary with the upper halo -- the same voxels of a periodic stack -- so counts match and the physics is right.
static int gather_faces(const float* state, const lsf_slab_layout* L, const lsf_slab_faces* F, hipStream_t s) {
    const bool lo = L->lo_rank >= 0, hi = L->hi_rank >= 0;
    const unsigned n_slo = lo ? (unsigned)F->send_count[0] : 0u, n_shi = hi ? (unsigned)F->send_count[1] : 0u;
    if (n_slo + n_shi > 0)
        hipLaunchKernelGGL(face_gather_kernel, dim3((n_slo + n_shi + 255) / 256), dim3(256), 0, s,
                           reinterpret_cast<const vf4*>(state), F->send_list[0], n_slo,
                           reinterpret_cast<vf4*>(F->send_msg[0]), F->send_list[1], n_shi,
                           reinterpret_cast<vf4*>(F->send_msg[1]));
    return (int)hipGetLastError();
}

// the messages are gathered already (gather_faces): wire + scatter on s
static int exchange_compact(lsf_slab_comm* c, float* state, const lsf_slab_layout* L, const lsf_slab_faces* F,
                            hipStream_t s) {
    const bool lo = L->lo_rank >= 0, hi = L->hi_rank >= 0;
    const unsigned n_slo = lo ? (unsigned)F->send_count[0] : 0u, n_shi = hi ? (unsigned)F->send_count[1] : 0u;
    const unsigned n_rlo = lo ? (unsigned)F->recv_count[0] : 0u, n_rhi = hi ? (unsigned)F->recv_count[1] : 0u;
    LSF_RCCL_CHECK(g_rccl.GroupStart());
    if (lo && n_slo) LSF_RCCL_CHECK(g_rccl.Send(F->send_msg[0], (size_t)n_slo * 4, ncclFloat, L->lo_rank, c->comm, s));
    if (hi && n_rhi) LSF_RCCL_CHECK(g_rccl.Recv(F->recv_msg[1], (size_t)n_rhi * 4, ncclFloat, L->hi_rank, c->comm, s));
    if (hi && n_shi) LSF_RCCL_CHECK(g_rccl.Send(F->send_msg[1], (size_t)n_shi * 4, ncclFloat, L->hi_rank, c->comm, s));
    if (lo && n_rlo) LSF_RCCL_CHECK(g_rccl.Recv(F->recv_msg[0], (size_t)n_rlo * 4, ncclFloat, L->lo_rank, c->comm, s));
    LSF_RCCL_CHECK(g_rccl.GroupEnd());
    if (n_rlo + n_rhi > 0)
        hipLaunchKernelGGL(face_scatter_kernel, dim3((n_rlo + n_rhi + 255) / 256), dim3(256), 0, s,
                           reinterpret_cast<vf4*>(state), F->recv_list[0], n_rlo,
                           reinterpret_cast<const vf4*>(F->recv_msg[0]), F->recv_list[1], n_rhi,
                           reinterpret_cast<const vf4*>(F->recv_msg[1]));
    return (int)hipGetLastError();
}

// exchange of the state's halo slices with the z-neighbours: each face is one contiguous run of halo * ny * nx float4
static int exchange_state(lsf_slab_comm* c, float* state, const lsf_slab_layout* L, hipStream_t s) {
    const size_t slice = (size_t)L->ny * L->nx * 4;  // floats per z-slice of the state
    const size_t count = slice * (size_t)L->halo;
    LSF_RCCL_CHECK(g_rccl.GroupStart());
    // same order as exchange_compact (see there)
    if (L->lo_rank >= 0) LSF_RCCL_CHECK(g_rccl.Send(state + slice * L->z_begin, count, ncclFloat, L->lo_rank, c->comm, s));
    if (L->hi_rank >= 0) LSF_RCCL_CHECK(g_rccl.Recv(state + slice * L->z_end, count, ncclFloat, L->hi_rank, c->comm, s));
    if (L->hi_rank >= 0)
        LSF_RCCL_CHECK(g_rccl.Send(state + slice * (L->z_end - L->halo), count, ncclFloat, L->hi_rank, c->comm, s));
    if (L->lo_rank >= 0)
        LSF_RCCL_CHECK(g_rccl.Recv(state + slice * (L->z_begin - L->halo), count, ncclFloat, L->lo_rank, c->comm, s));
    LSF_RCCL_CHECK(g_rccl.GroupEnd());
    return 0;
}

extern "C" int lsf_merge_sorted_runs(const int32_t* const* a, const int64_t* n_a, const int32_t* const* b,
                                     const int64_t* n_b, int32_t* const* out, int32_t pairs, void* stream) {
    (void)hipGetLastError();
    if (pairs < 0 || pairs > 4 || (pairs && (!a || !n_a || !b || !n_b || !out))) return LSF_ERR_BAD_ARGUMENT;
    MergePairs m;
    unsigned total = 0u;
    for (int f = 0; f < pairs; ++f) {
        if (n_a[f] < 0 || n_b[f] < 0 || n_a[f] + n_b[f] > 0x7fffffffll || (n_a[f] && !a[f]) || (n_b[f] && !b[f]) ||
            ((n_a[f] + n_b[f]) && !out[f]))
            return LSF_ERR_BAD_ARGUMENT;
        m.a[f] = a[f]; m.b[f] = b[f]; m.out[f] = out[f];
        m.na[f] = (unsigned)n_a[f]; m.nb[f] = (unsigned)n_b[f];
        m.first[f] = total;
        total += m.na[f] + m.nb[f];
    }
    for (int f = pairs; f < 4; ++f) { m.a[f] = m.b[f] = nullptr; m.out[f] = nullptr; m.na[f] = m.nb[f] = 0u; }
    for (int f = pairs; f <= 4; ++f) m.first[f] = total;
    if (total == 0u) return 0;
    hipLaunchKernelGGL(merge_runs_kernel, dim3((total + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), m,
                       pairs);
    return (int)hipGetLastError();
}

static int launch_parts(const float* state_in, const float* canonical, float* state_out, const lsf_slab_part* parts,
                        int32_t n, const lsf_slavcheva_params* params, const lsf_gate* gate,
                        lsf_iteration_record* record, void* stream) {
    for (int32_t k = 0; k < n; ++k) {
        const lsf_slab_part& p = parts[k];
        for (int32_t j = 0; j < p.n_lists; ++j)
            if (int e = lsf_slavcheva_state_iteration(state_in, canonical, state_out, &p.grid, params, gate, record,
                                                      p.band_list[j], p.band_count[j], p.band_subset[j], stream))
                return e;
    }
    return 0;
}

extern "C" int lsf_slab_state_iteration(lsf_slab_comm* comm, const float* state_in, const float* canonical,
                                        float* state_out, const lsf_slab_layout* layout,
                                        const lsf_slab_part* boundary_parts, int32_t n_boundary,
                                        const lsf_slab_part* interior_parts, int32_t n_interior,
                                        const lsf_slavcheva_params* params, const lsf_gate* gate,
                                        lsf_iteration_record* record, int32_t exchange, const lsf_slab_faces* faces,
                                        void* stream) {
    if (exchange == LSF_SLAB_LAUNCH || exchange == LSF_SLAB_RESUME) {
        // an iteration inside an exchange group: plain launches, nothing on the wire.  RESUME: the first part does not
        // touch the halos (the owned slices at least one slice away from them) and runs while the previous call's
        // exchange is still in flight; the launch stream waits for that exchange only before the second part
        if (!state_in || !canonical || !state_out || !params || !record) return LSF_ERR_BAD_ARGUMENT;
        if (int e = launch_parts(state_in, canonical, state_out, boundary_parts, n_boundary, params, gate, record, stream))
            return e;
        if (exchange == LSF_SLAB_RESUME) {
            if (!comm) return LSF_ERR_BAD_ARGUMENT;
            if (comm->pending >= 0)
                LSF_HIP_CHECK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), comm->halos_done[comm->pending], 0));
            comm->pending = -1;
        }
        return launch_parts(state_in, canonical, state_out, interior_parts, n_interior, params, gate, record, stream);
    }
    if (exchange != LSF_SLAB_EXCHANGE && exchange != LSF_SLAB_EXCHANGE_DEFERRED) return LSF_ERR_BAD_ARGUMENT;
    if (!comm || !state_in || !canonical || !state_out || !layout || !params || !record) return LSF_ERR_BAD_ARGUMENT;
    if ((n_boundary > 0 && !boundary_parts) || (n_interior > 0 && !interior_parts)) return LSF_ERR_BAD_ARGUMENT;
    if (layout->halo < 1 || layout->z_begin - (layout->lo_rank >= 0 ? layout->halo : 0) < 0 ||
        layout->z_end + (layout->hi_rank >= 0 ? layout->halo : 0) > layout->nz || layout->z_end - layout->z_begin < layout->halo ||
        layout->lo_rank >= comm->world || layout->hi_rank >= comm->world)
        return LSF_ERR_BAD_ARGUMENT;
    hipStream_t main = reinterpret_cast<hipStream_t>(stream);
    const unsigned k = comm->parity++ & 1u;
    // 1. the slices the neighbours are waiting for
    if (int e = launch_parts(state_in, canonical, state_out, boundary_parts, n_boundary, params, gate, record, stream))
        return e;
    // 2. their exchange on the communication stream, while ...  (compact faces: the messages are gathered on the LAUNCH
    //    stream, right behind the boundary parts -- 4 us there; on the communication stream the gather started ~13 us
    //    after the boundary parts ended and took 21 us squeezed in beside the interior part's one-workgroup-per-CU grid:
    //    kernel trace of the loop-back, profiles/r04_slab_rccl_loopback.txt)
    if (faces && comm->gather_on_main)
        if (int e = gather_faces(state_out, layout, faces, main)) return e;
    LSF_HIP_CHECK(hipEventRecord(comm->boundary_done[k], main));
    LSF_HIP_CHECK(hipStreamWaitEvent(comm->comm_stream, comm->boundary_done[k], 0));
    if (faces && !comm->gather_on_main)
        if (int e = gather_faces(state_out, layout, faces, comm->comm_stream)) return e;
    if (int e = faces ? exchange_compact(comm, state_out, layout, faces, comm->comm_stream)
                      : exchange_state(comm, state_out, layout, comm->comm_stream))
        return e;
    LSF_HIP_CHECK(hipEventRecord(comm->halos_done[k], comm->comm_stream));
    // 3. ... the interior runs on the launch stream
    if (int e = launch_parts(state_in, canonical, state_out, interior_parts, n_interior, params, gate, record, stream))
        return e;
    // 4. the next iteration reads the halos: wait now, or leave that to the next call (LSF_SLAB_RESUME), which first
    //    launches what does not depend on them
    if (exchange == LSF_SLAB_EXCHANGE_DEFERRED) {
        comm->pending = (int)k;
        return 0;
    }
    LSF_HIP_CHECK(hipStreamWaitEvent(main, comm->halos_done[k], 0));
    return 0;
}
