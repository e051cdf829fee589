// Probe helper of tools/probe_zsplit_pipeline.py: the A | M | B pipeline of a fixed-count run captured into ONE HIP graph by
// the HIP API itself (stream capture across three streams, fork / join by events) and launched on the caller's stream.
// The iteration launches go through the library's own entry point (a function pointer handed in by the caller).
// hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/probe/zsplit_graph.hip -o tools/probe/bin/libzsplit_graph.so
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef int (*iteration_fn)(const float*, const float*, float*, const void* grid, const void* params, const void* gate,
                            void* record, const int32_t* band_list, int64_t band_count, int32_t band_subset, void* stream);

struct Plan {
    hipGraphExec_t exec = nullptr;
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
};

#define CHECK(call)                                                                  \
    do {                                                                             \
        const hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "zsplit_graph: %s -> %s\n", #call, hipGetErrorString(e_)); \
            return nullptr;                                                          \
        }                                                                            \
    } while (0)

// parts: 0 = A, 1 = B, 2 = M.  pipelined == 0: one launch per iteration over part 0 (= the whole list) on one stream.
extern "C" void* zsplit_build(void* fn_ptr, float* state0, float* state1, const float* canonical, const void* grid,
                              const void* params, char* records, int64_t record_bytes, int32_t iterations,
                              const int32_t* const* lists, const int64_t* counts, int32_t subset, int32_t pipelined) {
    iteration_fn fn = reinterpret_cast<iteration_fn>(fn_ptr);
    Plan* p = new Plan();
    for (auto& s : p->s) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> events;
    auto new_event = [&]() {
        hipEvent_t e = nullptr;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        events.push_back(e);
        return e;
    };
    float* st[2] = {state0, state1};
    hipGraph_t graph = nullptr;
    for (int k = 0; k < 3 * iterations + 4; ++k) new_event();  // every event exists before the capture begins
    size_t next_event = 0;
    auto take_event = [&]() { return events[next_event++]; };
    if (pipelined == 2) {
        // no graph: the same dependencies enqueued directly, behind whatever the caller's stream holds (records[-1] unused)
        p->exec = nullptr;
    }
    if (pipelined != 2) CHECK(hipStreamBeginCapture(p->s[0], hipStreamCaptureModeGlobal));
    if (!pipelined) {
        for (int i = 0; i < iterations; ++i)
            if (fn(st[i % 2], canonical, st[(i + 1) % 2], grid, params, nullptr, records + (int64_t)i * record_bytes, lists[0],
                   counts[0], subset, p->s[0]))
                return nullptr;
    } else {
        hipEvent_t fork = take_event();
        CHECK(hipEventRecord(fork, p->s[0]));
        CHECK(hipStreamWaitEvent(p->s[1], fork, 0));
        CHECK(hipStreamWaitEvent(p->s[2], fork, 0));
        hipEvent_t done[3] = {nullptr, nullptr, nullptr};
        const int needs[3][2] = {{2, -1}, {2, -1}, {0, 1}};  // A <- M, B <- M, M <- A, B (of the previous iteration)
        for (int i = 0; i < iterations; ++i) {
            hipEvent_t now[3];
            for (int k = 0; k < 3; ++k) {
                for (int j = 0; j < 2; ++j)
                    if (needs[k][j] >= 0 && done[needs[k][j]]) CHECK(hipStreamWaitEvent(p->s[k], done[needs[k][j]], 0));
                if (fn(st[i % 2], canonical, st[(i + 1) % 2], grid, params, nullptr, records + (int64_t)i * record_bytes,
                       lists[k], counts[k], subset, p->s[k]))
                    return nullptr;
                now[k] = take_event();
                CHECK(hipEventRecord(now[k], p->s[k]));
            }
            for (int k = 0; k < 3; ++k) done[k] = now[k];
        }
        CHECK(hipStreamWaitEvent(p->s[0], done[1], 0));
        CHECK(hipStreamWaitEvent(p->s[0], done[2], 0));
    }
    if (pipelined == 2) {
        CHECK(hipStreamSynchronize(p->s[0]));
        return p;
    }
    CHECK(hipStreamEndCapture(p->s[0], &graph));
    CHECK(hipGraphInstantiate(&p->exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    return p;
}

extern "C" int zsplit_launch(void* plan, void* stream) {
    return (int)hipGraphLaunch(static_cast<Plan*>(plan)->exec, reinterpret_cast<hipStream_t>(stream));
}
