/*
 * subgraph_sketch_debug.h -- measurement-only entry points of libsubgraph_sketch.so.
 *
 * NOT part of the drop-in boundary (include/subgraph_sketch.h): nothing here replaces reference code.  bench.py and the
 * probes under tools/ use these to time launches with HIP events recorded on the very stream a kernel is launched on
 * (torch.cuda.Event only sees torch's current stream).  Implemented in csrc/ss_debug.hip; the event list is guarded by a
 * mutex and capped (SS_PROFILE_MAX_EVENTS launches; later launches are not recorded until ss_profile_read drains it).
 */
#ifndef SUBGRAPH_SKETCH_DEBUG_H
#define SUBGRAPH_SKETCH_DEBUG_H

#include "subgraph_sketch.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SS_PROFILE_MAX_EVENTS 65536

/* Live launch-duration measurement: while a kernel family's bit is enabled, every launch of it -- from any thread, on
 * any stream -- is bracketed by HIP events recorded on its own stream.  ss_profile_read(tag) synchronises the events of
 * that family, returns their mean duration (ms) and count through host pointers, and removes them from the list. */
#define SS_PROF_MINHASH_HOP 0     /* ss::propagate_kernel<128,256>, MinHash table hop (the dominant kernel of a build)   */
#define SS_PROF_HLL_HOP 1         /* ss::hll_propagate_row16_kernel, HLL table hop + cardinalities                       */
#define SS_PROF_FIRST_HOP_MH 2    /* ss::first_hop_kernel<..., true, false>                                              */
#define SS_PROF_FIRST_HOP_HLL 3   /* ss::hll_first_hop_kernel                                                            */
#define SS_PROF_PAIRS 4           /* ss::pair_features_kernel                                                            */
#define SS_PROF_CSR 5             /* all launches of one ss_csr_build                                                    */
#define SS_PROF_HUB 6             /* hub units as launches of their own (propagate_hub_kernel, first_hop_hub_kernel)     */
#define SS_PROF_FUSED 7           /* ss::fused_hop_persistent_kernel (MinHash first hop + HLL table hop in one launch)     */
#define SS_PROF_MINHASH_ROWS 8    /* ss_minhash_hop_rows: the MinHash table hop of a list of rows                          */
#define SS_PROF_TAGS 9
int ss_profile_enable(uint32_t tag_mask);   /* bit t enables family t; 0 disables everything */
/* Library calls since the last reset that were given hub lists and served hub units -- hosted by their row launches (which leave no
 * SS_PROF_HUB span) or as launches of their own; reset != 0 returns the count and clears it.  For the tests of the hub hints. */
int64_t ss_debug_hub_calls(int32_t reset);
/* Dedicated helper workgroups an ss_csr_build appends to its finish launch (SS_CSR_HELPERS caps it; `stream` is ignored: since
 * round 5 nothing waits for a helper, so a CU mask on the stream no longer changes the schedule). */
int ss_debug_csr_helpers(void *stream);
/* Cross-workgroup waits of any CSR build of this process that gave up after their ~2 s bound (ss_csr.hip "who may wait for whom":
 * only running workgroups are ever waited for, so this stays 0; the tests of the multi-process and CU-masked builds assert it).
 * -1: the counter could not be read. */
int ss_debug_csr_protocol_faults(void);
int ss_profile_read(int32_t tag, float *mean_ms_out, int32_t *launches_out);
/* Time only every `every`-th launch of an enabled family (the first after ss_profile_enable, then each `every`-th): a timed launch costs
 * its step about 5 us (the completion signal the events are filled from), which is 1 % of the step bench.py times.  Default 1. */
int ss_profile_sample(int32_t every);

/* Launch-duration probe for bench.py: records HIP events around `reps` back-to-back launches of the
 * same ss_propagate / ss_pair_features call ON `stream` and returns the mean milliseconds per launch in
 * *ms_out (host pointer).  Synchronises the stream. */
int ss_time_propagate(const ss_csr_graph *graph, const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                      const uint8_t *hll_in, uint8_t *hll_out, int32_t M,
                      float *cards_out, int64_t cards_stride, const ss_hll_params *prm, void *stream,
                      int32_t reps, float *ms_out);
int ss_time_pair_features(const int64_t *links, int64_t B, int64_t N, int32_t h,
                          const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                          const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                          float *out, void *stream, int32_t reps, float *ms_out);

/* ss_pair_features_grouped with its kernel forced (which = 0: the ordinary kernel walking `order`, 1: the run-aware kernel);
 * the product entry point chooses per hop count.  For tools/probe_pair_runs.py and the parity test of both kernels. */
int ss_pair_features_grouped_kernel(int32_t which, const int64_t *links, const int32_t *order, int64_t B, int64_t N, int32_t h,
                                    const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                                    const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                                    const float *degrees, float *out, int32_t *err_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SUBGRAPH_SKETCH_DEBUG_H */
