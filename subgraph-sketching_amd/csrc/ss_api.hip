// ss_api.hip -- library identification and HIP-event launch-duration probes (used by bench.py for the
// roofline figure: events are recorded on the very stream the kernel is launched on).
#include "ss_common.hpp"

extern "C" int ss_version(void) { return 110; /* 0.1.1: ss_csr_graph row ranges + mega rows, ss_csr_build mega outputs, heuristics / spmm entry points */ }

extern "C" const char *ss_error_string(int code)
{
    switch (code) {
        case SS_OK: return "ok";
        case SS_ERR_INVALID_ARG: return "invalid argument";
        case SS_ERR_LAUNCH: return "HIP launch / runtime error";
        case SS_ERR_WORKSPACE: return "workspace too small";
        case SS_ERR_UNSUPPORTED: return "unsupported parameter combination";
        default: return "unknown error";
    }
}

namespace {
template <typename F>
int time_launches(hipStream_t stream, int reps, float *ms_out, F &&launch)
{
    if (reps < 1 || !ms_out) return SS_ERR_INVALID_ARG;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return SS_ERR_LAUNCH;
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
