// ss_api.hip -- library identification (the measurement-only probes live in ss_debug.hip / subgraph_sketch_debug.h).
#include "ss_common.hpp"

extern "C" int ss_version(void) { return 129; /* 0.2.9: + ss_table_digest, ss_csr_protocol_faults; a CSR build whose bounded wait gave up sets bit 1 of err_flag (SS_CSR_ERR_PROTOCOL); 0.2.8: dense buckets of the CSR build are worked off inside the finish launch (per-bucket sync words in the workspace, no launches of their own), + ss_debug_csr_protocol_faults; 0.2.7: mega_rows entries are SS_MEGA_DESC_WORDS = 8 words (a ticket per sketch side), hub units hosted by the row launches, + ss_debug_hub_calls; 0.2.6: + ss_csr_group_ids / ss_csr_sort_rows / ss_gcn_degree / ss_sign_spmm, CSR build by tile-sort levels (another workspace layout); 0.2.5: ss_csr_graph carries the peers' tables of a peer-write build (mirror_*), + ss_group_links_by_source / ss_pair_features_grouped / ss_gather_links / ss_scatter_feature_rows / ss_csr_build_cached; 0.2.4: SS_MEGA_SLICE 4096 -> 1024, dense fine buckets of the CSR build split by edges (larger workspace); 0.2.3: + ss_minhash_hop_rows; 0.2.2: ss_fused_hop_stage owns the hop-1 HLL rows too; 0.2.1: + ss_fused_hop_stage; 0.2.0: probes moved to subgraph_sketch_debug.h (tagged ss_profile_*), mega-row hand-off drained before the ticket */ }

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
