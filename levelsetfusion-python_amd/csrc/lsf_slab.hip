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
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
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
                    bind(api.CommCount, "ncclCommCount") && bind(api.CommUserRank, "ncclCommUserRank") &&
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

extern "C" int lsf_slab_comm_info(lsf_slab_comm* c, int32_t* rank_out, int32_t* nranks_out) {
    if (!c || !rank_out || !nranks_out) return LSF_ERR_BAD_ARGUMENT;
    int rank = -1, count = -1;  // RCCL's own answer (ncclCommUserRank / ncclCommCount), not what the creator passed in
    LSF_RCCL_CHECK(g_rccl.CommUserRank(c->comm, &rank));
    LSF_RCCL_CHECK(g_rccl.CommCount(c->comm, &count));
    *rank_out = rank;
    *nranks_out = count;
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

// ======================================================================================================================
// A whole fixed-count call of a z-slab rank, enqueued by the library (include/lsf_hip.h: lsf_slab_run_begin / _finish;
// round 6).  The schedule is the one SlabMixin._plan_slab (engine_slab.py) builds for lsf_slab_state_iteration, here
// derived from the layout and the cut positions in C: the 50 Python -> C calls of a call, the 13 part descriptors, the
// face plan's tensor bookkeeping and two torch.distributed collectives become two foreign calls.
// ======================================================================================================================
namespace {

struct SlabCuts {
    int chunk[LSF_SLAB_MAX_CUTS];  // first 1024-voxel chunk of the cut slice (== chunks for a cut at the end of the array)
    int n, chunks;
};

// cut_entries[which][k] = list entries of subset `which` in front of cut slice k: the counting pass's exclusive per-chunk
// prefix counts (two arrays, chunks + 1 apart, at the start of its scratch), or the subset's total for a cut at the end
__global__ void slab_cuts_kernel(const int* __restrict__ prepare_scratch, const long long* __restrict__ totals, SlabCuts c,
                                 long long* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= 2 * c.n) return;
    const int which = k / c.n, j = k % c.n;
    out[which * LSF_SLAB_MAX_CUTS + j] =
        c.chunk[j] >= c.chunks ? totals[which] : (long long)prepare_scratch[(size_t)which * (c.chunks + 1) + c.chunk[j]];
}

__global__ __launch_bounds__(256) void slab_record_words_kernel(const long long* __restrict__ records,
                                                                long long* __restrict__ out, int n_slots) {
    const int k = blockIdx.x * 256 + threadIdx.x;  // (slot, word)
    if (k < n_slots * 4) out[k] = records[(long long)(k >> 2) * (sizeof(lsf_record_slot) / 8) + (k & 3)];
}

inline bool slab_run_ok(const lsf_slab_run* r) {
    const lsf_state_run& b = r->base;
    const lsf_slab_layout& L = r->layout;
    const lsf_grid& g = b.grid;
    return b.live && b.canonical && b.state[0] && b.state[1] && b.state[0] != b.state[1] && b.prepare_scratch &&
           b.totals_device && b.totals_host && !b.box_scratch && b.sparse_reach >= 0 && b.sparse_reach <= 8 &&
           g.dims == 3 && g.z_begin == 0 && g.z_end == g.nz && g.y_global_offset == 0 &&
           (g.ny_global == 0 || g.ny_global == g.ny) && L.nz == g.nz && L.ny == g.ny && L.nx == g.nx && L.halo >= 1 &&
           L.z_begin >= (L.lo_rank >= 0 ? L.halo : 0) && L.z_end + (L.hi_rank >= 0 ? L.halo : 0) <= L.nz &&
           L.z_end - L.z_begin >= 2 * L.halo && ((long long)g.ny * g.nx) % 1024 == 0 &&
           (r->exchange_interval == 1 || r->exchange_interval == L.halo);
}

// the slices at which the schedule cuts the lists (SlabMixin._slab_cut_slices): every boundary of a widened, boundary,
// interior or resume range, ascending, inside the array
inline int slab_cut_slices(const lsf_slab_layout& L, int32_t* out) {
    bool mark[4096] = {};
    if (L.nz + 1 > 4096) return -1;
    auto add = [&](int z) { if (z >= 0 && z <= L.nz) mark[z] = true; };
    for (int e = 0; e <= L.halo; ++e) { add(L.z_begin - e); add(L.z_end + e); }
    add(L.z_begin + L.halo); add(L.z_end - L.halo); add(L.z_begin + 1); add(L.z_end - 1);
    int n = 0;
    for (int z = 0; z <= L.nz; ++z)
        if (mark[z]) {
            if (n == LSF_SLAB_MAX_CUTS) return -1;
            out[n++] = z;
        }
    return n;
}

struct CutTable {
    const lsf_slab_run* run;
    int64_t at(int which, int z) const {  // entries of list `which` in front of slice z (z must be a cut slice)
        for (int k = 0; k < run->n_cuts; ++k)
            if (run->cut_slices[k] == z) return run->cut_entries[which][k];
        return -1;
    }
};

// entries of both lists inside [z0, z1)
inline int64_t entries_in(const CutTable& t, int z0, int z1) {
    return (t.at(0, z1) - t.at(0, z0)) + (t.at(1, z1) - t.at(1, z0));
}

struct Range { int z0, z1; };

// the launch of a phase over up to two z-ranges (ascending, disjoint): per band list ONE run of entries -- the list is
// sorted by voxel index, so a z-range is a contiguous run of it, and two ranges are concatenated into the scratch once
struct PartBuilder {
    const lsf_slab_run* run;
    CutTable cuts;
    const int32_t* list[2];
    int64_t total[2];
    int32_t* scratch;
    int64_t scratch_used, scratch_size;
    hipStream_t stream;
    int error = 0;

    // returns the number of parts (0 or 1) written to *out
    int build(const Range* ranges, int n_ranges, lsf_slab_part* out) {
        Range r[2];
        int n = 0;
        for (int k = 0; k < n_ranges; ++k)
            if (ranges[k].z1 > ranges[k].z0) r[n++] = ranges[k];
        if (!n) return 0;
        const lsf_slab_layout& L = run->layout;
        lsf_slab_part p;
        memset(&p, 0, sizeof(p));
        p.grid = run->base.grid;
        p.grid.z_begin = r[0].z0;
        p.grid.z_end = r[n - 1].z1;
        p.grid.energy_z_begin = L.z_begin;
        p.grid.energy_z_end = L.z_end;
        p.n_lists = 0;
        for (int which = 0; which < 2; ++which) {
            if (!total[which]) continue;
            int64_t first[2], count[2], sum = 0;
            int pieces = 0;
            for (int k = 0; k < n; ++k) {
                const int64_t a = cuts.at(which, r[k].z0), b = cuts.at(which, r[k].z1);
                if (a < 0 || b < a) { error = LSF_ERR_BAD_ARGUMENT; return 0; }
                if (b > a) { first[pieces] = a; count[pieces++] = b - a; sum += b - a; }
            }
            if (!pieces) continue;
            const int32_t* at = list[which] + first[0];
            if (pieces == 2) {
                if (scratch_used + sum > scratch_size) { error = LSF_ERR_BAD_ARGUMENT; return 0; }
                int32_t* dst = scratch + scratch_used;
                scratch_used += sum;
                if (hipMemcpyAsync(dst, list[which] + first[0], (size_t)count[0] * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                    hipMemcpyAsync(dst + count[0], list[which] + first[1], (size_t)count[1] * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
                    error = (int)hipGetLastError();
                    return 0;
                }
                at = dst;
            }
            p.band_list[p.n_lists] = at;
            p.band_count[p.n_lists] = sum;
            p.band_subset[p.n_lists++] = which == 0 ? LSF_BAND_INTERIOR : LSF_BAND_BOUNDARY;
        }
        if (!p.n_lists) {  // at least one (empty) list, so that the launch still reports the arg-max of an all-zero update
            const int which = total[0] ? 0 : 1;
            p.band_list[0] = list[which] ? list[which] : reinterpret_cast<const int32_t*>(run->base.prepare_scratch);
            p.band_count[0] = 0;
            p.band_subset[0] = which == 0 ? LSF_BAND_INTERIOR : LSF_BAND_BOUNDARY;
            p.n_lists = 1;
        }
        *out = p;
        return 1;
    }
};

// how many int32 of index scratch a call needs: the two-range parts' concatenations (boundary part of an exchange, outer
// part of a resume) and the merged face lists
inline void slab_scratch_sizes(const lsf_slab_run* run, int64_t* index_entries, int64_t* face_entries) {
    const lsf_slab_layout& L = run->layout;
    const CutTable t{run};
    const bool lo = L.lo_rank >= 0, hi = L.hi_rank >= 0;
    const int h = L.halo, e_last = run->exchange_interval - 1;
    int64_t faces = 0, concat = 0;
    if (lo) faces += entries_in(t, L.z_begin, L.z_begin + h) + entries_in(t, L.z_begin - h, L.z_begin);
    if (hi) faces += entries_in(t, L.z_end - h, L.z_end) + entries_in(t, L.z_end, L.z_end + h);
    if (lo && hi) {
        concat += entries_in(t, L.z_begin, L.z_begin + h) + entries_in(t, L.z_end - h, L.z_end);  // boundary part
        if (e_last > 0)
            concat += entries_in(t, L.z_begin - e_last, L.z_begin + 1) + entries_in(t, L.z_end - 1, L.z_end + e_last);
    }
    *face_entries = faces;
    *index_entries = concat + faces + 16;
}

}  // namespace

extern "C" int lsf_slab_run_begin(lsf_slab_run* run, void* stream) {
    if (!run || !slab_run_ok(run)) return LSF_ERR_BAD_ARGUMENT;
    const lsf_state_run& b = run->base;
    const lsf_grid* g = &b.grid;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool sparse = b.sparse_reach > 0;
    run->n_cuts = slab_cut_slices(run->layout, run->cut_slices);
    if (run->n_cuts < 2) return LSF_ERR_BAD_ARGUMENT;
    float* b_in_pass = (sparse || b.second_state_late) ? nullptr : b.state[1];
    if (int e = lsf_state_prepare(b.live, b.canonical, sparse ? nullptr : b.state[0], b_in_pass, g, b.prepare_scratch,
                                  b.totals_device, stream))
        return e;
    SlabCuts c;
    const long long per_slice = ((long long)g->ny * g->nx) / 1024;
    c.n = run->n_cuts;
    c.chunks = (int)(per_slice * g->nz);
    for (int k = 0; k < c.n; ++k) c.chunk[k] = (int)(per_slice * run->cut_slices[k]);
    hipLaunchKernelGGL(slab_cuts_kernel, dim3(1), dim3(2 * LSF_SLAB_MAX_CUTS), 0, s, b.prepare_scratch,
                       reinterpret_cast<const long long*>(b.totals_device), c,
                       reinterpret_cast<long long*>(b.totals_device) + 5);
    LSF_HIP_CHECK(hipGetLastError());
    LSF_HIP_CHECK(hipMemsetAsync(b.totals_device + 4, 0, sizeof(int64_t), s));
    LSF_HIP_CHECK(hipMemcpyAsync(b.totals_host, b.totals_device, (5 + 2 * LSF_SLAB_MAX_CUTS) * sizeof(int64_t),
                                 hipMemcpyDeviceToHost, s));
    hipEvent_t sizes = nullptr;
    LSF_HIP_CHECK(hipEventCreateWithFlags(&sizes, hipEventDisableTiming));
    int status = 0;
    if (hipEventRecord(sizes, s) != hipSuccess) status = (int)hipGetLastError();
    // the states are written BEHIND the copy of the sizes: the card fills them while the host wakes up on the sizes
    if (!status && sparse)
        status = lsf_state_pack_needed(b.live, b.state[0], b.state[1], g, b.prepare_scratch, b.sparse_reach, 0, stream);
    else if (!status && b.second_state_late)
        status = lsf_state_pack(b.live, nullptr, b.state[1], nullptr, g, stream);
    if (!status && hipEventSynchronize(sizes) != hipSuccess) status = (int)hipGetLastError();
    (void)hipEventDestroy(sizes);
    if (status) return status;
    for (int which = 0; which < 2; ++which)
        for (int k = 0; k < run->n_cuts; ++k) run->cut_entries[which][k] = b.totals_host[5 + which * LSF_SLAB_MAX_CUTS + k];
    slab_scratch_sizes(run, &run->out_index_entries, &run->out_face_entries);
    return 0;
}

extern "C" int lsf_slab_run_finish(const lsf_slab_run* run, lsf_slab_comm* comm, const lsf_slavcheva_params* params,
                                   int32_t* list_interior, int32_t* list_boundary, int32_t* index_scratch,
                                   float* face_messages, lsf_iteration_record* records, int32_t iterations,
                                   float* live_out, int64_t* words_device, int64_t* words_host,
                                   lsf_state_run_result* result, void* stream) {
    if (!run || !slab_run_ok(run) || !comm || !params || !records || iterations < 1 || !live_out || !words_device ||
        !words_host || !result || !result->max_value || !result->argmax || !result->energies3 || !result->executed ||
        !index_scratch)
        return LSF_ERR_BAD_ARGUMENT;
    const lsf_state_run& b = run->base;
    const lsf_slab_layout& L = run->layout;
    const lsf_grid* g = &b.grid;
    if (L.lo_rank >= comm->world || L.hi_rank >= comm->world) return LSF_ERR_BAD_ARGUMENT;
    const int64_t n_interior = b.totals_host[0], n_boundary = b.totals_host[1];
    if (n_interior < 0 || n_boundary < 0 || n_interior > 0x7fffffffll || n_boundary > 0x7fffffffll ||
        (n_interior && !list_interior) || (n_boundary && !list_boundary))
        return LSF_ERR_BAD_ARGUMENT;
    hipStream_t main = reinterpret_cast<hipStream_t>(stream);
    if (n_interior)
        if (int e = lsf_band_list_fill_prepared(g, LSF_BAND_INTERIOR, b.prepare_scratch, list_interior, stream)) return e;
    if (n_boundary)
        if (int e = lsf_band_list_fill_prepared(g, LSF_BAND_BOUNDARY, b.prepare_scratch, list_boundary, stream)) return e;

    const bool lo = L.lo_rank >= 0, hi = L.hi_rank >= 0;
    const int h = L.halo, k_group = run->exchange_interval, e_last = k_group - 1;
    PartBuilder pb{run, CutTable{run}, {list_interior, list_boundary}, {n_interior, n_boundary}, index_scratch, 0,
                   run->out_index_entries, main};

    // ---- the schedule's parts (SlabMixin._plan_slab) ----------------------------------------------------------------
    lsf_slab_part widened[32];  // [e]: the owned range widened by e slices on every interior side
    int n_widened[32];
    if (k_group > 32) return LSF_ERR_BAD_ARGUMENT;
    for (int e = 0; e < k_group; ++e) {
        const Range r{L.z_begin - (lo ? e : 0), L.z_end + (hi ? e : 0)};
        n_widened[e] = pb.build(&r, 1, &widened[e]);
    }
    const int z_lo = L.z_begin + (lo ? h : 0), z_hi = L.z_end - (hi ? h : 0);
    const bool exchanges = iterations > k_group;  // is there an iteration j == k - 1 with another one behind it?
    lsf_slab_part x_boundary, x_interior, r_first, r_second;
    int n_xb = 0, n_xi = 0, n_r1 = 0, n_r2 = 0;
    if (exchanges) {
        const Range bnd[2] = {{L.z_begin, lo ? z_lo : L.z_begin}, {hi ? z_hi : L.z_end, L.z_end}};
        const Range inner{z_lo, z_hi};
        n_xb = pb.build(bnd, 2, &x_boundary);
        n_xi = pb.build(&inner, 1, &x_interior);
        if (k_group > 1) {
            const int in_lo = L.z_begin + (lo ? 1 : 0), in_hi = L.z_end - (hi ? 1 : 0);
            const Range first{in_lo, in_hi};
            const Range outer[2] = {{lo ? L.z_begin - e_last : in_lo, in_lo}, {in_hi, hi ? L.z_end + e_last : in_hi}};
            n_r1 = pb.build(&first, 1, &r_first);
            n_r2 = pb.build(outer, 2, &r_second);
        }
    }
    if (pb.error) return pb.error;

    // ---- compact faces: the band voxels of the boundary / halo slices, ascending (both lists merged) ------------------
    lsf_slab_faces faces;
    memset(&faces, 0, sizeof(faces));
    bool compact = exchanges && face_messages != nullptr;
    if (compact) {
        const CutTable& t = pb.cuts;
        const Range face_ranges[4] = {{L.z_begin, L.z_begin + h}, {L.z_end - h, L.z_end},      // send lower, upper
                                      {L.z_begin - h, L.z_begin}, {L.z_end, L.z_end + h}};     // recv lower, upper
        const bool present[4] = {lo, hi, lo, hi};
        const int32_t* face_list[4] = {nullptr, nullptr, nullptr, nullptr};
        int64_t face_count[4] = {0, 0, 0, 0};
        const int32_t* ma[4]; const int32_t* mb[4]; int32_t* mo[4];
        int64_t mna[4], mnb[4];
        int merges = 0;
        for (int f = 0; f < 4; ++f) {
            if (!present[f]) continue;
            const int64_t a0 = t.at(0, face_ranges[f].z0), a1 = t.at(0, face_ranges[f].z1);
            const int64_t b0 = t.at(1, face_ranges[f].z0), b1 = t.at(1, face_ranges[f].z1);
            if (a0 < 0 || a1 < a0 || b0 < 0 || b1 < b0) return LSF_ERR_BAD_ARGUMENT;
            face_count[f] = (a1 - a0) + (b1 - b0);
            if (a1 > a0 && b1 > b0) {
                if (pb.scratch_used + face_count[f] > pb.scratch_size) return LSF_ERR_BAD_ARGUMENT;
                ma[merges] = list_interior + a0; mna[merges] = a1 - a0;
                mb[merges] = list_boundary + b0; mnb[merges] = b1 - b0;
                mo[merges] = index_scratch + pb.scratch_used;
                face_list[f] = mo[merges++];
                pb.scratch_used += face_count[f];
            } else if (a1 > a0) {
                face_list[f] = list_interior + a0;
            } else if (b1 > b0) {
                face_list[f] = list_boundary + b0;
            }
        }
        if (merges)
            if (int e = lsf_merge_sorted_runs(ma, mna, mb, mnb, mo, merges, stream)) return e;
        float* msg = face_messages;
        const int32_t* none = reinterpret_cast<const int32_t*>(b.prepare_scratch);  // a valid address for an empty face
        for (int f = 0; f < 4; ++f) {
            const int side = f & 1;
            (f < 2 ? faces.send_list : faces.recv_list)[side] = face_list[f] ? face_list[f] : none;
            (f < 2 ? faces.send_count : faces.recv_count)[side] = face_count[f];
            (f < 2 ? faces.send_msg : faces.recv_msg)[side] = msg;
            msg += 4 * (face_count[f] > 0 ? face_count[f] : 1);
        }
        const int64_t mine[4] = {faces.send_count[0], faces.send_count[1], faces.recv_count[0], faces.recv_count[1]};
        if (int e = lsf_slab_face_counts_begin(comm, mine)) return e;
    }

    // ---- the iterations ---------------------------------------------------------------------------------------------
    bool faces_checked = !compact;
    for (int32_t i = 0; i < iterations; ++i) {
        const float* s_in = b.state[i % 2];
        float* s_out = b.state[(i + 1) % 2];
        const int j = i % k_group;
        const bool exchange = j == k_group - 1 && i + 1 < iterations;
        const bool resume = k_group > 1 && j == 0 && i > 0;
        int e = 0;
        if (exchange) {
            if (!faces_checked) {
                // the neighbours' counts (the collective started in front of the first iteration has long finished): a rank's
                // lower boundary lands in its lower neighbour's UPPER halo, its upper boundary in the upper one's LOWER halo
                int64_t table[4 * 64];
                if (comm->world > 64) return LSF_ERR_BAD_ARGUMENT;
                if (int ee = lsf_slab_face_counts_end(comm, table)) return ee;
                bool ok = true;
                if (comm->world == 1) ok = table[0] == table[3] && table[1] == table[2];  // the loop-back: its own neighbour
                for (int r = 0; r + 1 < comm->world; ++r)
                    ok = ok && table[4 * r + 1] == table[4 * (r + 1) + 2] && table[4 * (r + 1) + 0] == table[4 * r + 3];
                if (!ok) compact = false;  // every rank sees every row: all of them fall back to whole slices together
                faces_checked = true;
            }
            e = lsf_slab_state_iteration(comm, s_in, b.canonical, s_out, &L, &x_boundary, n_xb, &x_interior, n_xi, params,
                                         nullptr, records + i, k_group > 1 ? LSF_SLAB_EXCHANGE_DEFERRED : LSF_SLAB_EXCHANGE,
                                         compact ? &faces : nullptr, stream);
        } else if (resume) {
            e = lsf_slab_state_iteration(comm, s_in, b.canonical, s_out, &L, &r_first, n_r1, &r_second, n_r2, params, nullptr,
                                         records + i, LSF_SLAB_RESUME, nullptr, stream);
        } else {
            const int w = j == k_group - 1 ? 0 : k_group - 1 - j;
            e = lsf_slab_state_iteration(comm, s_in, b.canonical, s_out, &L, nullptr, 0, &widened[w], n_widened[w], params,
                                         nullptr, records + i, LSF_SLAB_LAUNCH, nullptr, stream);
        }
        if (e) return e;
    }
    if (compact && !faces_checked) {  // (cannot happen: compact implies an exchange) -- never leave the collective open
        int64_t table[4 * 64];
        (void)lsf_slab_face_counts_end(comm, table);
    }

    // ---- the end of the call: final live values of the listed voxels into the caller's array, the records of all ranks --
    const bool sparse = b.sparse_reach > 0;
    const int32_t* listed[2] = {nullptr, nullptr};
    int64_t listed_counts[2] = {0, 0};
    int n_listed = 0;
    if (n_interior) { listed[n_listed] = list_interior; listed_counts[n_listed++] = n_interior; }
    if (n_boundary) { listed[n_listed] = list_boundary; listed_counts[n_listed++] = n_boundary; }
    if (int e = lsf_state_finalize_listed(b.state[iterations % 2], b.canonical, live_out, nullptr, g, listed, listed_counts,
                                          n_listed, 0, -1, 0.0f, nullptr, nullptr, nullptr, sparse ? records : nullptr,
                                          sparse ? iterations : 0, (float)b.sparse_reach, stream))
        return e;
    const int n_slots = iterations * LSF_RECORD_SLOTS;
    const size_t words = (size_t)n_slots * 4;
    hipLaunchKernelGGL(slab_record_words_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, main,
                       reinterpret_cast<const long long*>(records), reinterpret_cast<long long*>(words_device), n_slots);
    LSF_HIP_CHECK(hipGetLastError());
    // every rank's words side by side on every rank: one collective on the communicator's stream (all of a
    // communicator's operations are issued there, in the same order on every rank)
    const unsigned k = comm->parity++ & 1u;
    LSF_HIP_CHECK(hipEventRecord(comm->boundary_done[k], main));
    LSF_HIP_CHECK(hipStreamWaitEvent(comm->comm_stream, comm->boundary_done[k], 0));
    LSF_RCCL_CHECK(g_rccl.AllGather(words_device, words_device + words, words, ncclInt64, comm->comm, comm->comm_stream));
    LSF_HIP_CHECK(hipMemcpyAsync(words_host, words_device + words, words * (size_t)comm->world * sizeof(int64_t),
                                 hipMemcpyDeviceToHost, comm->comm_stream));
    LSF_HIP_CHECK(hipStreamSynchronize(comm->comm_stream));
    LSF_HIP_CHECK(hipStreamSynchronize(main));
    comm->pending = -1;
    // rank-major [world][iterations][slots][4] -> per iteration the ranks' slots side by side (in place would overlap: go
    // through the decoder one iteration at a time)
    {
        int64_t row[64 * LSF_RECORD_SLOTS * 4];
        if (comm->world > 64) return LSF_ERR_BAD_ARGUMENT;
        for (int32_t i = 0; i < iterations; ++i) {
            for (int r = 0; r < comm->world; ++r)
                memcpy(row + (size_t)r * LSF_RECORD_SLOTS * 4,
                       words_host + ((size_t)r * iterations + i) * LSF_RECORD_SLOTS * 4, LSF_RECORD_SLOTS * 4 * sizeof(int64_t));
            if (int e = lsf_records_decode(row, 1, LSF_RECORD_SLOTS * comm->world, 4, result->max_value + i,
                                           result->argmax + i, result->energies3 + 3 * (size_t)i, result->executed + i))
                return e;
        }
    }
    result->final_state = iterations % 2;
    result->n_lists = n_listed;
    result->reach_exceeded = 0;
    result->compact_faces = exchanges ? (compact ? 1 : 0) : -1;
    for (int32_t i = 0; i < iterations && sparse; ++i)
        if (result->executed[i] && !(result->max_value[i] < (float)b.sparse_reach)) result->reach_exceeded = 1;
    return 0;
}
