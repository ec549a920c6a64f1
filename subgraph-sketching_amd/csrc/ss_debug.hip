// ss_debug.hip -- measurement-only entry points (include/subgraph_sketch_debug.h): HIP-event brackets around launches,
// recorded on the stream the kernel is launched on.  Not part of the drop-in boundary.  The span list is process-wide,
// guarded by a mutex (launches may come from any thread / stream) and capped at SS_PROFILE_MAX_EVENTS.
#include <atomic>
#include <mutex>
#include <vector>

#include "ss_common.hpp"
#include "subgraph_sketch_debug.h"

namespace ss {

namespace {
std::atomic<uint32_t> g_mask{0};
std::mutex g_mutex;
struct Span {
    hipEvent_t start, stop;
    int tag;
};
std::vector<Span> g_spans;
std::atomic<int64_t> g_hub_calls{0};
std::atomic<uint32_t> g_every{1};                 // ss_profile_sample: every n-th launch of an enabled family is timed
std::atomic<uint32_t> g_seen[SS_PROF_TAGS] = {};  // launches of each family since the last ss_profile_enable
}  // namespace

void note_hub_call() { g_hub_calls.fetch_add(1, std::memory_order_relaxed); }

ProfileSpan::ProfileSpan(hipStream_t s, int tag_, bool attached_) : stream(s), tag(tag_), attached(attached_)
{
    if (tag < 0 || tag >= SS_PROF_TAGS || !((g_mask.load(std::memory_order_relaxed) >> tag) & 1u)) return;
    const uint32_t every = g_every.load(std::memory_order_relaxed);
    if (every > 1 && g_seen[tag].fetch_add(1, std::memory_order_relaxed) % every != 0) return;  // (the 1st, (n+1)-th, ... launch)
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_spans.size() >= (size_t)SS_PROFILE_MAX_EVENTS) return;
    }
    if (hipEventCreate(&start) != hipSuccess) { start = nullptr; return; }
    if (hipEventCreate(&stop) != hipSuccess) { (void)hipEventDestroy(start); start = nullptr; return; }
    if (!attached) (void)hipEventRecord(start, stream);
}

ProfileSpan::~ProfileSpan()
{
    if (!start) return;
    if (!attached) (void)hipEventRecord(stop, stream);
    std::lock_guard<std::mutex> lock(g_mutex);
    g_spans.push_back(Span{start, stop, tag});
}

}  // namespace ss

extern "C" int64_t ss_debug_hub_calls(int32_t reset)
{
    return reset ? ss::g_hub_calls.exchange(0, std::memory_order_relaxed) : ss::g_hub_calls.load(std::memory_order_relaxed);
}

extern "C" int ss_profile_enable(uint32_t tag_mask)
{
    ss::g_mask.store(tag_mask, std::memory_order_relaxed);
    for (auto &c : ss::g_seen) c.store(0, std::memory_order_relaxed);
    return SS_OK;
}

extern "C" int ss_profile_sample(int32_t every)
{
    if (every < 1) return SS_ERR_INVALID_ARG;
    ss::g_every.store((uint32_t)every, std::memory_order_relaxed);
    return SS_OK;
}

extern "C" int ss_profile_read(int32_t tag, float *mean_ms_out, int32_t *launches_out)
{
    if (!mean_ms_out || !launches_out || tag < 0 || tag >= SS_PROF_TAGS) return SS_ERR_INVALID_ARG;
    std::vector<ss::Span> mine;
    {
        std::lock_guard<std::mutex> lock(ss::g_mutex);
        std::vector<ss::Span> rest;
        for (auto &sp : ss::g_spans) (sp.tag == tag ? mine : rest).push_back(sp);
        ss::g_spans.swap(rest);
    }
    double total = 0.0;
    int n = 0;
    for (auto &sp : mine) {  // synchronising outside the lock: launches of other families keep recording
        float ms = 0.0f;
        if (hipEventSynchronize(sp.stop) == hipSuccess && hipEventElapsedTime(&ms, sp.start, sp.stop) == hipSuccess) {
            total += ms;
            ++n;
        }
        (void)hipEventDestroy(sp.start);
        (void)hipEventDestroy(sp.stop);
    }
    *mean_ms_out = n ? (float)(total / n) : 0.0f;
    *launches_out = n;
    return SS_OK;
}

namespace {
template <typename F>
int time_launches(hipStream_t stream, int reps, float *ms_out, F &&launch)
{
    if (reps < 1 || !ms_out) return SS_ERR_INVALID_ARG;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return SS_ERR_LAUNCH;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return SS_ERR_LAUNCH; }
    int rc = launch();  // warm-up (code object load, caches)
    if (rc == SS_OK) {
        (void)hipEventRecord(e0, stream);
        for (int r = 0; r < reps && rc == SS_OK; ++r) rc = launch();
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) rc = SS_ERR_LAUNCH;
        float ms = 0.0f;
        if (rc == SS_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = SS_ERR_LAUNCH;
        *ms_out = ms / (float)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
}  // namespace

extern "C" int ss_time_propagate(const ss_csr_graph *graph, const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                                 const uint8_t *hll_in, uint8_t *hll_out, int32_t M,
                                 float *cards_out, int64_t cards_stride, const ss_hll_params *prm, void *stream,
                                 int32_t reps, float *ms_out)
{
    return time_launches((hipStream_t)stream, reps, ms_out, [&]() {
        return ss_propagate(graph, mh_in, mh_out, P, hll_in, hll_out, M, cards_out, cards_stride, prm,
                            stream);
    });
}

extern "C" int ss_time_pair_features(const int64_t *links, int64_t B, int64_t N, int32_t h,
                                     const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                                     const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                                     float *out, void *stream, int32_t reps, float *ms_out)
{
    return time_launches((hipStream_t)stream, reps, ms_out, [&]() {
        return ss_pair_features(links, B, N, h, mh, P, hll, cards, cards_stride, prm, flags, out, nullptr, nullptr, nullptr,
                                nullptr, stream);
    });
}
