/*
 * subgraph_sketch.h -- C ABI of the MI355X (gfx950) subgraph-sketching engine.
 *
 * The reference (melifluos/subgraph-sketching) has no FFI layer: its hot path is the Python class
 * `ElphHashes` in src/hashing.py.  This header is the boundary a replacement binds to; each entry
 * point names the reference code it replaces (file:line under /root/reference).  The Python host
 * class in subgraph-sketching_amd/hashing.py is the only in-tree caller (through ctypes); the
 * reference-side binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  Every data pointer is a DEVICE pointer unless the
 *     comment says "host".  No torch types.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and returns 0 on success or a negative SS_ERR_* code (argument / launch errors are detected
 *     synchronously on the host; nothing is thrown).
 *   - Canonical sketch layout in HBM ("packed"): MinHash rows are uint32[P] (every reference value
 *     is < 2^32, hashing.py:59,122), HyperLogLog rows are uint8[M], M = 2^p (values 0..64-p,
 *     hashing.py:75-76).  Row-major [N, P] / [N, M], rows 16-byte aligned (P % 4 == 0, p >= 4).
 *   - The caller owns every buffer, including workspaces.
 */
#ifndef SUBGRAPH_SKETCH_H
#define SUBGRAPH_SKETCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_OK 0
#define SS_ERR_INVALID_ARG (-1)   /* null pointer / size / unsupported P, p, h                    */
#define SS_ERR_LAUNCH (-2)        /* hipGetLastError() != hipSuccess after a launch                */
#define SS_ERR_WORKSPACE (-3)     /* workspace too small                                           */
#define SS_ERR_UNSUPPORTED (-4)   /* parameter combination has no kernel                           */

#define SS_MAX_HOPS 3             /* hashing.py:54 */
#define SS_MAX_TABLE 512          /* max entries of the HLL++ bias tables (datasketch ships <= 200) */

/* flags of ss_pair_features (hashing.py:56,67) */
#define SS_FLAG_USE_ZERO_ONE 1u
#define SS_FLAG_FLOOR_SF 2u

/* HyperLogLog++ estimator constants -- everything ElphHashes.__init__ takes from datasketch
 * (hashing.py:69-80) plus two host-derived helpers.  All fp32 values are rounded on the host exactly
 * as torch rounds the reference's Python scalars. */
typedef struct ss_hll_params {
    int32_t p;              /* hll_p; m = 1 << p                                   (hashing.py:65-66) */
    int32_t n_tbl;          /* entries in raw_est / bias, 6 <= n_tbl <= SS_MAX_TABLE                  */
    float alpha_mm;         /* fp32(alpha * m^2)                                   (hashing.py:228)   */
    float threshold;        /* fp32(hll_threshold)                                 (hashing.py:78)    */
    int32_t lc_min_zeros;   /* linear counting is returned iff V >= lc_min_zeros (V = #zero registers,
                               V > 0); derived on the host from lc_table <= threshold (hashing.py:220-226) */
    int32_t reserved;
    const float *raw_est;   /* device [n_tbl]: estimate_vector SORTED ascending     (hashing.py:80)    */
    const float *bias;      /* device [n_tbl]: bias_vector, permuted like raw_est  (hashing.py:79)    */
    const float *lc_table;  /* device [m+1]: lc_table[V] = m*log(m/V) in fp32, V>=1 (hashing.py:194-195) */
} ss_hll_params;

/* library / build identification */
int ss_version(void);
const char *ss_error_string(int code);

/* Hop-0 MinHash rows.  Replaces ElphHashes.initialise_minhash (hashing.py:118-124); a/b are the
 * permutation parameters of _init_permutations (hashing.py:106-116, drawn on the host with numpy's
 * RandomState(1)), device uint64[P].  Writes rows for nodes first_node .. first_node+n-1
 * (node id i hashes the value i+1, hashing.py:121). */
int ss_minhash_init(uint32_t *out, int64_t first_node, int64_t n, const uint64_t *a, const uint64_t *b, int32_t P,
                    void *stream);

/* Hop-0 HyperLogLog rows.  Replaces ElphHashes.initialise_hll + _get_hll_rank (hashing.py:126-137,
 * 91-104): one non-zero register per row. */
int ss_hll_init(uint8_t *out, int64_t first_node, int64_t n, int32_t p, void *stream);

/* Destination-grouped adjacency resident on the device (built by ss_csr_build, consumed by the propagation
 * kernels).  The struct itself lives on the host and is read at launch. */
typedef struct ss_csr_graph {
    const int64_t *rowptr;            /* device int64[N+1]                                                    */
    const int32_t *col;               /* device int32[E]: source ids grouped by destination                    */
    int64_t num_nodes;                /* N                                                                     */
    int64_t n_self_loops;             /* rows i < n_self_loops also receive their own row (implicit self loop)  */
    const int64_t *n_self_loops_dev;  /* device int64 (nullable): overrides n_self_loops, read by the kernels   */
    int32_t hub_threshold;            /* rows with more than this many in-edges are "hub rows" ...              */
    int32_t reserved;
    const int32_t *hub_rows;          /* ... listed here (device int32[*hub_count], nullable) and processed by  */
    const int32_t *hub_count;         /* a 16-wave cooperative kernel instead of a single wavefront             */
    const int32_t *mega_rows;         /* device int32[4 * max]: {row, first_slice, n_slices, done} per "mega row" */
    const int32_t *mega_count;        /* device int32[2]: {mega rows, slices}.  Rows with more than SS_MEGA_SLICE   */
    void *mega_scratch;               /* neighbours are walked slice by slice by ALL hub workgroups; the partial     */
                                      /* rows go through mega_scratch (SS_MEGA_SLOT_BYTES each), the last slice to   */
                                      /* finish combines them.  All three nullable together (then such rows are hub rows).  */
    int64_t row_begin;                /* destination rows [row_begin, row_end) are computed by ss_propagate /    */
    int64_t row_end;                  /* ss_first_hop (multi-GPU destination-range sharding);       */
                                      /* row_end == 0 means all rows.  Outputs are indexed by the GLOBAL row id.  */
    /* Peer-write build (the row-sharded build without an exchange step): every row a launch finishes is ALSO    */
    /* stored into these tables of the other ranks (device pointers into peers' memory, mapped through hipIpc / torch's       */
    /* CUDA-IPC; same shapes and strides as the launch's own mh_out / hll_out / cards_out; entries of a sketch the launch does */
    /* not produce are ignored).  The stores travel over xGMI while the kernel runs; the hop boundary then needs a cross-rank   */
    /* barrier only.  n_mirrors == 0: none.                                                                                   */
    int32_t n_mirrors;
    int32_t reserved2;
    uint32_t *mirror_mh[7];           /* SS_MAX_MIRRORS */
    uint8_t *mirror_hll[7];
    float *mirror_cards[7];
    /* Hub-row report (nullable): the first-hop launches from node ids -- the HLL first-hop kernel and the MinHash rows kernel   */
    /* of ss_first_hop, the HLL first-hop launch of ss_fused_hop_stage (not its fused kernel: a stage called with cards1_out ==  */
    /* NULL, the deferred first hop, reports nothing) -- store *report_hub_count + report_mega_count[0], the counters            */
    /* ss_csr_build left on the device, into *hub_report, a device-VISIBLE int32 such as pinned host memory, at their very start */
    /* (the store is long complete when the launch ends).  The word must stay valid until those launches have run.  The host    */
    /* may read it later WITHOUT synchronising, as a hint: a shape whose earlier build listed no such rows is propagated with    */
    /* hub_rows = NULL -- no hub units are served (leading workgroups that find nothing to do on an unskewed graph), every row   */
    /* is walked by its row kernel, so a stale hint costs time, never correctness.                                              */
    int32_t *hub_report;
    const int32_t *report_hub_count;
    const int32_t *report_mega_count;
} ss_csr_graph;
#define SS_MAX_MIRRORS 7

/* CSR-by-destination of an edge list.  Replaces the message materialisation of
 * torch_geometric MessagePassing.propagate as used by hashing.py:34,44 (flow source -> target).
 *   src/dst: device int64[E] (edge_index[0], edge_index[1]);  rowptr: device int64[N+1];
 *   col: device int32[E] (source ids grouped by destination, order inside a row unspecified).
 *   n_self_loops_out: device int64 (nullable) <- max(edge_index) + 1 (0 for E == 0): the number of self loops
 *   torch_geometric.utils.add_self_loops(edge_index) appends when num_nodes is not given (hashing.py:148), so
 *   the host never has to synchronise on edge_index.max().
 *   hub_rows / hub_count (device int32[N] / int32, both nullable): rows with more than hub_threshold in-edges.
 *   err_flag: device int32 (nullable), set to 1 if any endpoint is outside [0, N) (such edges are dropped).
 * Workspace: ss_csr_workspace_bytes(N, E) bytes (0 = unsupported size).  No per-edge global atomics: a
 * two-level counting sort (LDS histograms per edge slice -> bucket offsets -> per-bucket LDS sort). */
#define SS_MEGA_SLICE 1024      /* neighbours per slice of a mega row (one 64-neighbour chunk per wavefront of a hub workgroup) */
#define SS_MEGA_SLOT_BYTES 1280 /* scratch per slice: partial MinHash row (<= 256 x u32) + partial HLL row (256 B) */
#define SS_MEGA_DESC_WORDS 8   /* int32 words per entry of mega_rows: {row, first slice, slices, ticket (MinHash side), ticket (HLL side), 0, 0, 0} */
size_t ss_csr_workspace_bytes(int64_t N, int64_t E);
/* mega_rows / mega_count (nullable together): rows with more than max(hub_threshold, SS_MEGA_SLICE) in-edges are listed
 * there instead of in hub_rows: mega_rows must hold SS_MEGA_DESC_WORDS * (E / SS_MEGA_SLICE + 1) int32, mega_count 2 int32; the slices of
 * all mega rows number at most 3 * (E / SS_MEGA_SLICE + 1) -- the scratch ss_csr_graph.mega_scratch needs SS_MEGA_SLOT_BYTES for each. */
int ss_csr_build(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                 int64_t *n_self_loops_out, int32_t hub_threshold, int32_t *hub_rows, int32_t *hub_count,
                 int32_t *mega_rows, int32_t *mega_count,
                 int32_t *err_flag, void *workspace, size_t workspace_bytes, void *stream);
/* Failure of a build that was launched (every entry point that runs the builder: ss_csr_build, ss_csr_build_cached,
 * ss_group_links_by_source, ss_csr_group_ids).  Buckets too dense for one workgroup are worked off in shares by several workgroups of
 * the finish launch; the one cross-workgroup wait of that protocol is bounded (~2 s: it can only end late when the process is
 * descheduled for seconds).  A wait that gives up leaves the outputs INCOMPLETE and is reported, never silently:
 *   - bit 1 (SS_CSR_ERR_PROTOCOL) of *err_flag, when a flag was given (bit 0 = an endpoint outside [0, N), as before: test the bits);
 *   - ss_csr_protocol_faults(): 0 while no wait of this process (any device, any stream) has given up; afterwards a positive stamp
 *     that CHANGES with every further one.  Kept in pinned host memory and read WITHOUT synchronising -- compare before / after a
 *     synchronised build, or poll it as the host mirror does (ElphHashes.check_errors and every later call raise RuntimeError).
 *     -1: the word could not be allocated (no device). */
#define SS_CSR_ERR_BOUNDS 1
#define SS_CSR_ERR_PROTOCOL 2
int ss_csr_protocol_faults(void);
/* ss_csr_build preceded by a device-side content check: `fingerprint` (device buffer of SS_CSR_FINGERPRINT_BYTES, zeroed by the
 * caller before its first use and tied to THESE output buffers) holds two 64-bit sums over the edge list the outputs were last
 * built from; when the sums of (src, dst) agree every kernel of the build exits at once, otherwise the build runs and the sums are
 * replaced.  ELPH.forward (models/elph.py:186) concatenates the same self-looped edge_index into a fresh tensor every training
 * step: its CSR costs one 41 MB streaming pass instead of a rebuild.  No host synchronisation. */
#define SS_CSR_FINGERPRINT_BYTES 8448
int ss_csr_build_cached(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *rowptr, int32_t *col,
                        int64_t *n_self_loops_out, int32_t hub_threshold, int32_t *hub_rows, int32_t *hub_count,
                        int32_t *mega_rows, int32_t *mega_count, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                        void *fingerprint, void *stream);

/* One hop of sketch propagation over a CSR: out[i] = min (MinHash) / max (HLL) over the in-neighbours
 * of i, plus row i itself when i < n_self_loops (the implicit self loops of add_self_loops,
 * hashing.py:148); rows with no in-edge and no self loop are all-zero (PyG scatter default).
 * Replaces MinhashPropagation.forward / HllPropagation.forward (hashing.py:28-45) and, when
 * cards_out != NULL, the hll_count of hashing.py:163 (cards_out[i*cards_stride] = hll_count(out row)).
 * Either sketch may be NULL (both in and out). */
int ss_propagate(const ss_csr_graph *graph, const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                 const uint8_t *hll_in, uint8_t *hll_out, int32_t M,
                 float *cards_out, int64_t cards_stride, const ss_hll_params *prm, void *stream);

/* The MinHash half of ss_propagate for a LIST of destination rows: mh_out[r] = min over the in-neighbours of r (and r itself,
 * as above) for every r in rows[0 .. n_rows) -- ids may repeat, negative ids count from the end (torch indexing), ids outside
 * [-N, N) are ignored -- plus every hub row of the graph (the hub units always cover all of them); all other rows
 * of mh_out are left untouched.  For the caller whose next step reads only a few rows of the hop's table: ELPH's training
 * step (models/elph.py:209-212 followed by runners/train.py:204) propagates over the whole graph and then queries two rows
 * per link of ONE batch; the host mirror defers the last minhash_prop (hashing.py:28-35) and computes the batch's rows
 * through this entry point -- same values, 2 B instead of N rows. */
int ss_minhash_hop_rows(const ss_csr_graph *graph, const uint32_t *mh_in, uint32_t *mh_out, int32_t P, const int64_t *rows,
                        int64_t n_rows, void *stream);

/* Hop 1 straight from node ids: equivalent to ss_minhash_init + ss_hll_init + one ss_propagate
 * (hashing.py:118-137, 28-45 at k = 1, 163) but the hop-0 rows -- pure functions of the node id -- are
 * recomputed in registers instead of being written to and re-read from HBM.  a / b: device uint64[P].
 * Returns SS_ERR_UNSUPPORTED when (P, p) is outside the fused kernel's shape (p == 8, P % 64 == 0, P <= 256):
 * the caller then uses the three-call sequence. */
int ss_first_hop(const ss_csr_graph *graph, const uint64_t *a, const uint64_t *b, int32_t P,
                 uint32_t *mh_out, int32_t p, uint8_t *hll_out, float *cards_out, int64_t cards_stride,
                 const ss_hll_params *prm, void *stream);

/* Hops 1 and 2 of a build in one call: the hop-1 MinHash rows (from node ids), the hop-2 HLL rows + their cardinalities and --
 * when mh2_out is given (P == 128) -- the hop-2 MinHash rows; with cards1_out != NULL also the hop-1 HLL rows (`hll1` is then an
 * OUTPUT, cards1_out[i * cards_stride] their cardinalities), with cards1_out == NULL `hll1` is the COMPLETE hop-1 HLL table as
 * input.  Same results as ss_first_hop + ss_propagate (hashing.py:118-137, 28-45 at k = 1, 2, 163), but the VALU-bound MinHash first
 * hop and the memory-bound HLL table hop of hop 2 run interleaved inside one launch (csrc/ss_fused_hop.hip).  cards2_out:
 * nullable unless cards1_out is given.  Returns SS_ERR_UNSUPPORTED outside p == 8, P % 64 == 0, P <= 256: the caller then uses the
 * unfused calls. */
int ss_fused_hop_stage(const ss_csr_graph *graph, const uint64_t *a, const uint64_t *b, int32_t P, uint32_t *mh1_out,
                       uint32_t *mh2_out, int32_t p, uint8_t *hll1, float *cards1_out, uint8_t *hll2_out, float *cards2_out,
                       int64_t cards_stride, const ss_hll_params *prm, void *stream);

/* HLL++ cardinality of n register rows.  Replaces ElphHashes.hll_count (+ _linearcounting,
 * _estimate_bias, _refine_hll_count_estimate; hashing.py:194-232).  regs: device uint8[n, m];
 * out: device fp32, element i written to out[i*out_stride]. */
int ss_hll_count(const uint8_t *regs, int64_t n, const ss_hll_params *prm, float *out, int64_t out_stride,
                 void *stream);

/* The two estimator helpers the reference exposes on their own.  e: device fp32[n] raw estimates.
 *   refine == 0: out[i] = mean bias of the 6 nearest table entries   (_estimate_bias, hashing.py:197-204)
 *   refine != 0: out[i] = e[i] <= 5m ? e[i] - that bias : e[i]       (_refine_hll_count_estimate, :206-210) */
int ss_estimate_bias(const float *e, int64_t n, const ss_hll_params *prm, float *out, int32_t refine, void *stream);

/* Subgraph features of B node pairs.  Replaces ElphHashes._get_intersections + jaccard + _hll_merge +
 * get_subgraph_features for one chunk (hashing.py:167-189, 234-237, 247-256, 258-323).
 *   links: device int64[B,2]; negative ids wrap like torch indexing; ids outside [-N, N) set *err_flag
 *          (device int32, may be NULL) and produce NaN rows.
 *   mh / hll: HOST arrays of h device pointers, entry k-1 = hop-k table (k = 1..h).
 *   cards: device fp32, cards[i*cards_stride + k-1] = hop-k cardinality of node i.
 *   out: device fp32 [B, h(h+2)], feature order = LABEL_LOOKUP[h] (hashing.py:22-25).
 *   dbg_match / dbg_zero (device int32[B,h*h], nullable): MinHash match counts and union zero-register
 *   counts per (k1,k2) row-major; dbg_inter (device fp32[B,h*h], nullable): the intersections J*U. */
int ss_pair_features(const int64_t *links, int64_t B, int64_t N, int32_t h,
                     const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                     const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                     float *out, int32_t *dbg_match, int32_t *dbg_zero, float *dbg_inter, int32_t *err_flag,
                     void *stream);

/* The same features with BUDDY's degree-normalised copy appended (next row of the scope table: replaces
 * BUDDY._append_degree_normalised, models/elph.py:276-293, fed by HashDataset.degrees, datasets/elph.py:74).
 *   degrees: device fp32[N];  out: device fp32 [B, 2*h(h+2)]: columns [0, h(h+2)) as ss_pair_features, columns
 *   [h(h+2), 2h(h+2)) = feature / sqrt(degrees[u] * degrees[v]) with NaN and Inf (zero-degree nodes) replaced by 0. */
int ss_pair_features_normalised(const int64_t *links, int64_t B, int64_t N, int32_t h,
                                const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                                const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                                const float *degrees, float *out, int32_t *err_flag, void *stream);

/* The query exploiting LINK LOCALITY (BUDDY's precompute IS the query at ogbl-ppa / citation2 scale: datasets/elph.py:207-208
 * hands get_subgraph_features every link of a split; hashing.py:270-274, :180-183 read the rows of u and v once per pair, yet
 * the link sets repeat every source many times -- ogbl-citation2's evaluation set lists 1 000 negatives per source).
 *   ss_group_links_by_source: order[0 .. B) := a permutation of the pair indices in which all pairs with the same first node
 *     are consecutive (torch-style negative ids wrapped; ids out of range are keyed to node 0 -- nothing is dropped, the query
 *     reports them).  rowptr: int64[N + 1] scratch output (start of every node's group).  Workspace:
 *     ss_csr_workspace_bytes(N, B).  B < 2^31.
 *   ss_pair_features_grouped: ss_pair_features / ss_pair_features_normalised (degrees non-null) walking the pairs in the order
 *     given (order == NULL: as listed) and re-reading the rows of a first node only when it changes.  Row q of `out` is pair
 *     q; rows are bit-identical to the ungrouped entry points' (a pair's features depend on its own rows only). */
int ss_group_links_by_source(const int64_t *links, int64_t B, int64_t N, int32_t *order, int64_t *rowptr, void *workspace,
                             size_t workspace_bytes, void *stream);
/* The two permutations around a grouped query over a link set of gigabytes (the features of ogbl-citation2's 356 M links are
 * 21 GB): out_links[t] := links[order[t]] (int64 [n, 2]) before, out[order[t], :] := rows[t, :] (float [n, width]) after a
 * ss_pair_features_grouped(order = NULL) over the gathered chunk -- walking `order` inside the query would make every pair
 * read and write at random places of those arrays from inside its latency chain. */
/* (links and out_links must be 16-byte aligned -- a pair moves as one vector --: SS_ERR_INVALID_ARG otherwise.  `order` entries are
 * NOT validated by any of these calls: each must lie in [0, n) of the array it indexes, as ss_group_links_by_source produces them) */
int ss_gather_links(const int64_t *links, const int32_t *order, int64_t n, int64_t *out_links, void *stream);
int ss_scatter_feature_rows(const float *rows, const int32_t *order, int64_t n, int32_t width, float *out, void *stream);
int ss_pair_features_grouped(const int64_t *links, const int32_t *order, int64_t B, int64_t N, int32_t h,
                             const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                             const float *cards, int64_t cards_stride, const ss_hll_params *prm, uint32_t flags,
                             const float *degrees, float *out, int32_t *err_flag, void *stream);

/* Weighted common-neighbour scores of node pairs -- the other per-link precompute of HashDataset.__init__ (SURVEY 8(f)
 * row N4; reference datasets/elph.py:76-77,314 calling heuristics.py:51-70 RA; CN heuristics.py:10-27 and AA :30-48
 * are the same sum with another multiplier):
 *     out[q] = (float) sum_w A[u, w] * (A[v, w] * mult[w]),   (u, v) = links[q], fp64 inside like scipy.
 *   rowptr / col / val: device CSR of A (int64[N+1], int32[nnz] SORTED and duplicate-free inside a row -- what
 *   scipy.sparse.csr_matrix((w, (row, col))) holds after sum_duplicates/sort_indices --, double[nnz] or NULL = all 1);
 *   mult: device double[N] or NULL (= 1: common neighbours).  links: device int64[B, 2]; out: device fp32[B].
 *   err_flag (nullable): set to 1 when a link refers to a node outside [0, N) (its score is written as 0). */
int ss_common_neighbour_scores(const int64_t *rowptr, const int32_t *col, const double *val, const double *mult,
                               int64_t N, const int64_t *links, int64_t B, float *out, int32_t *err_flag, void *stream);

/* out = A * x for a row-grouped CSR with fp32 values -- the node-feature propagation of
 * HashDataset._generate_sign_features (reference datasets/elph.py:87-110: gcn_norm, then torch_sparse.spmm = multiply
 * and scatter-add in edge order).  Every output element is accumulated by one lane in CSR order, product and sum rounded
 * separately, so with a CSR made by a STABLE sort of the reference's edge list the result equals the sequential
 * scatter-add bit for bit.  rowptr: device int64[N+1]; col: int32[nnz]; val: fp32[nnz]; x, out: fp32 [N, F], F % 4 == 0. */
int ss_spmm_csr(const int64_t *rowptr, const int32_t *col, const float *val, int64_t N, const float *x, int32_t F,
                float *out, void *stream);

/* The entries of an id array grouped by id (the CSR builder with the entry index as payload): order[0 .. E) = entry indices with
 * equal ids consecutive, rowptr[N + 1] = where each id's group starts; unspecified order inside a group until ss_csr_sort_rows
 * makes it ascending = STABLE.  Ids outside [0, N) (after torch-style negative wrapping) raise err_flag and are keyed to node 0.
 * Workspace: ss_csr_workspace_bytes(N, E).  (reference datasets/elph.py:100-107: gcn_norm's degree sums and torch_sparse.spmm's
 * scatter-add accumulate in edge order -- sign.py groups the edge list by column and by row) */
int ss_csr_group_ids(const int64_t *ids, int64_t E, int64_t N, int32_t *order, int64_t *rowptr, int32_t *err_flag, void *workspace,
                     size_t workspace_bytes, void *stream);
/* every row of a CSR sorted ascending in place; workspace: ss_csr_sort_workspace_bytes(E) device bytes.  only_if (nullable): a
 * device word -- the rows are sorted only if it is non-zero when the launches run */
size_t ss_csr_sort_workspace_bytes(int64_t E);
int ss_csr_sort_rows(const int64_t *rowptr, int32_t *col, int64_t N, int64_t E, const int32_t *only_if, void *workspace, size_t workspace_bytes,
                     void *stream);
/* gcn_norm + torch_sparse.spmm of HashDataset._generate_sign_features (reference datasets/elph.py:100-107) without materialising
 * the normalised edge list [PyG semantics restated: add_remaining_self_loops(fill 1), deg = index_add over col, deg^-1/2 with
 * inf -> 0, norm = dinv[row] * w * dinv[col]].  ss_gcn_scan_edges: one pass over the edge list into `scan` (ss_gcn_scan_bytes(N)
 * device bytes; its first int32 word: some weight differs from 1) -- unit weights? existing self loops per node.  ss_gcn_degree:
 * dinv / loop_w [N] from the grouping by COLUMN (stable unless the weights are all 1: then the degrees are counts);
 * ss_sign_spmm: out = A_norm x from the stable grouping by ROW, every output element accumulated by one lane in edge order
 * (existing self loops skipped, the node's remaining loop last), products and sums rounded separately. */
size_t ss_gcn_scan_bytes(int64_t N);
int ss_gcn_scan_edges(const int64_t *row, const int64_t *col, const float *w, int64_t E, int64_t N, void *scan, void *stream);
int ss_gcn_degree(const int64_t *rowptr_c, const int32_t *order_c, const int64_t *row, const float *w, int64_t N, const void *scan,
                  float *dinv, float *loop_w, void *stream);
int ss_sign_spmm(const int64_t *rowptr_r, const int32_t *order_r, const int64_t *col, const float *w, const float *dinv,
                 const float *loop_w, const void *scan, int64_t N, const float *x, int32_t F, float *out, void *stream);

/* 128-bit content digest of a device buffer (bytes % 16 == 0, 16-byte aligned): out[0] = sum, out[1] = xor over its 16-byte chunks
 * of a 64-bit mix of (chunk, chunk index) -- order-independent to compute, position-dependent in value.  out: device uint64[2],
 * overwritten (the call clears it first).  One streaming pass at HBM rate.  For the multi-GPU builds: the reference has ONE
 * table (hashing.py:139-165); a build that leaves a replica on every rank -- above all the peer-write build, whose rows arrive
 * through other GPUs' stores -- compares the digests of all replicas after its first build (dist.verify_replicas). */
int ss_table_digest(const void *data, int64_t bytes, uint64_t *out, void *stream);

/* int64 <-> packed uint32 MinHash tables (the reference's tensors are int64, hashing.py:124). */
int ss_pack_minhash(const int64_t *in, uint32_t *out, int64_t count, void *stream);
int ss_unpack_minhash(const uint32_t *in, int64_t *out, int64_t count, void *stream);

/* Measurement-only entry points (launch-duration probes used by bench.py and tools/) are declared in
 * subgraph_sketch_debug.h; they are not part of the drop-in boundary. */

#ifdef __cplusplus
}
#endif
#endif /* SUBGRAPH_SKETCH_H */
