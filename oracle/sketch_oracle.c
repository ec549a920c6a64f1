/*
 * sketch_oracle.c -- CPU restatement of the subgraph-sketching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP engine in
 * subgraph-sketching_amd/csrc.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; the product path never does.
 *
 * It restates, function by function, /root/reference/src/hashing.py (cited per function).
 * Pinning: checked against golden vectors produced by importing the reference itself in
 * the build container (tests/golden/make_golden.py -> tests/golden/ *.npz; test_oracle_golden.py).
 *   - everything integer (hop-0 sketches, propagated tables, match / zero counts),
 *     the linear-counting branch, the raw-estimate branch with e > 5m and the feature
 *     algebra are pinned against the reference.
 *   - PARITY UNPINNED for the HLL++ bias-corrected branch *table values*: the tables come from
 *     the un-vendored, un-pinned third-party package `datasketch` (hashing.py:12,78-80;
 *     README.md:45), absent from this image.  The tables are therefore inputs of every function
 *     here; the golden vectors for that branch were produced by the reference code fed the
 *     regenerated tables (subgraph-sketching_amd/data/make_hllpp_tables.py) and are flagged
 *     `*_uses_tables`.
 *
 * Layouts: MinHash rows are uint32 (all reference values are < 2^32: hashing.py:59,122),
 * HLL rows are uint8 (reference int8 holding 0..64-p: hashing.py:75-76,137).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MERSENNE61 ((uint64_t)((1ULL << 61) - 1))

typedef struct {
    int32_t p;             /* hll precision; m = 1 << p                                   */
    int32_t n_tbl;         /* length of raw_est / bias                                    */
    float alpha_mm;        /* fp32(alpha * m * m)           hashing.py:228                */
    float threshold;       /* fp32(hll_threshold)           hashing.py:78,220,226         */
    const float *raw_est;  /* estimate_vector               hashing.py:80                 */
    const float *bias;     /* bias_vector                   hashing.py:79                 */
    const float *lc_table; /* optional [m+1]: lc_table[V] = m*log(m/V) as the host torch computes it;
                              NULL -> computed here with logf                               */
} so_hll_params;

/* pandas.util.hash_array on an int64 array == the splitmix64 finaliser applied to the u64 view
 * (hashing.py:121,128 call it on arange(1, n+1)). */
static inline uint64_t hash_u64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

void so_hash_nodes(int64_t first_node, int64_t n, uint64_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = hash_u64((uint64_t)(first_node + i + 1));
}

/* hashing.py:118-124  initialise_minhash.  a, b from _init_permutations (hashing.py:106-116),
 * generated on the host with numpy's legacy RandomState(1). */
void so_minhash_init(int64_t first_node, int64_t n, int32_t P, const uint64_t *a, const uint64_t *b, uint32_t *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t hv = hash_u64((uint64_t)(first_node + i + 1));
        for (int32_t j = 0; j < P; ++j) {
            const uint64_t t = a[j] * hv + b[j];             /* uint64 wrap-around, as numpy     */
            out[i * (int64_t)P + j] = (uint32_t)((t % MERSENNE61) & 0xFFFFFFFFULL);
        }
    }
}

/* hashing.py:126-137 initialise_hll, :91-104 _get_hll_rank, :83-89 _np_bit_length.
 * rank = (64-p) - bit_length(hv >> p) + 1.  Returns the number of nodes whose rank <= 0 (the
 * reference raises ValueError then, hashing.py:101-103). */
int64_t so_hll_init(int64_t first_node, int64_t n, int32_t p, uint8_t *out)
{
    const int64_t m = (int64_t)1 << p;
    int64_t bad = 0;
    memset(out, 0, (size_t)(n * m));
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t hv = hash_u64((uint64_t)(first_node + i + 1));
        const uint64_t idx = hv & (uint64_t)(m - 1);
        const uint64_t bits = hv >> p;
        const int bl = bits ? 64 - __builtin_clzll(bits) : 0;
        const int rank = (64 - p) - bl + 1;
        if (rank <= 0) { ++bad; continue; }
        out[i * m + (int64_t)idx] = (uint8_t)rank;
    }
    return bad;
}

/* hashing.py:28-45 MinhashPropagation / HllPropagation over the (self-looped) edge list:
 * out[i] = min / max over edges (j -> i) of x[j]; rows without an in-edge are 0.
 * Either sketch may be NULL.  Edge-list scatter, exactly the reference's dataflow. */
void so_propagate_edges(int64_t N, int64_t E, const int64_t *src, const int64_t *dst,
                        const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                        const uint8_t *hll_in, uint8_t *hll_out, int32_t M)
{
    uint8_t *seen = (uint8_t *)calloc((size_t)(N > 0 ? N : 1), 1);
    if (mh_out) memset(mh_out, 0, (size_t)N * P * sizeof(uint32_t));
    if (hll_out) memset(hll_out, 0, (size_t)N * M);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t j = src[e], i = dst[e];
        if (mh_out) {
            const uint32_t *x = mh_in + j * P;
            uint32_t *o = mh_out + i * P;
            if (!seen[i]) memcpy(o, x, (size_t)P * sizeof(uint32_t));
            else for (int32_t c = 0; c < P; ++c) o[c] = x[c] < o[c] ? x[c] : o[c];
        }
        if (hll_out) {
            const uint8_t *x = hll_in + j * M;
            uint8_t *o = hll_out + i * M;
            for (int32_t c = 0; c < M; ++c) o[c] = x[c] > o[c] ? x[c] : o[c];
        }
        seen[i] = 1;
    }
    free(seen);
}

/* hashing.py:194-195 */
static inline float linear_counting(const so_hll_params *prm, int64_t m, int64_t num_zero)
{
    if (prm->lc_table) return prm->lc_table[num_zero];
    return (float)m * logf((float)m / (float)num_zero);
}

/* hashing.py:197-204 _estimate_bias: mean of the bias entries at the 6 smallest fp32 squared
 * distances (e - raw_est[j])^2.  Ties resolved towards the lower index. */
static float estimate_bias(const so_hll_params *prm, float e)
{
    int best[6];
    float bd[6];
    int nb = 0;
    for (int j = 0; j < prm->n_tbl; ++j) {
        const float diff = e - prm->raw_est[j];
        const float d = diff * diff;
        int pos = nb;
        while (pos > 0 && d < bd[pos - 1]) --pos; /* strict: equal distances keep index order */
        if (pos >= 6) continue;
        const int last = nb < 6 ? nb : 5;
        for (int k = last; k > pos; --k) { bd[k] = bd[k - 1]; best[k] = best[k - 1]; }
        bd[pos] = d; best[pos] = j;
        if (nb < 6) ++nb;
    }
    float s = 0.0f;
    for (int k = 0; k < nb; ++k) s += prm->bias[best[k]];
    return s / (float)nb;
}

/* hashing.py:197-204 (refine == 0: the mean bias itself) and :206-210 (refine != 0: e <= 5m ? e - bias : e) on their own */
void so_estimate_bias(const float *e, int64_t n, const so_hll_params *prm, int32_t refine, float *out)
{
    const float five_m = 5.0f * (float)((int64_t)1 << prm->p);
    for (int64_t i = 0; i < n; ++i) {
        const float b = estimate_bias(prm, e[i]);
        out[i] = refine ? (e[i] <= five_m ? e[i] - b : e[i]) : b;
    }
}

/* hashing.py:212-232 hll_count for one register row (+ :206-210 _refine_hll_count_estimate).
 * branch (optional out): 0 = linear counting, 1 = raw estimate with bias correction (e <= 5m),
 * 2 = raw estimate unchanged (e > 5m). */
static float hll_count_row(const so_hll_params *prm, const uint8_t *regs, int64_t stride_bytes, int32_t *zeros_out,
                           int32_t *branch)
{
    const int64_t m = (int64_t)1 << prm->p;
    int64_t num_zero = 0;
    double s = 0.0; /* sum_j 2^-reg_j; every term is a power of two, the double sum is (near-)exact */
    for (int64_t j = 0; j < m; ++j) {
        const uint8_t r = regs[j * stride_bytes];
        num_zero += (r == 0);
        s += ldexp(1.0, -(int)r);
    }
    if (zeros_out) *zeros_out = (int32_t)num_zero;
    float retval = prm->threshold + 1.0f;                       /* hashing.py:220 */
    if (num_zero > 0) retval = linear_counting(prm, m, num_zero); /* :221-224      */
    if (!(retval > prm->threshold)) {                            /* :226          */
        if (branch) *branch = 0;
        return retval;
    }
    float e = prm->alpha_mm / (float)s;                          /* :228          */
    if (e <= 5.0f * (float)m) {                                  /* :207          */
        e = e - estimate_bias(prm, e);                           /* :208-209      */
        if (branch) *branch = 1;
    } else if (branch) *branch = 2;
    return e;
}

/* regs: [n, m] elements of elem_bytes (1 = uint8/int8, 8 = int64 little endian, low byte read) */
void so_hll_count(const void *regs, int32_t elem_bytes, int64_t n, const so_hll_params *prm, float *out,
                  int32_t *branch_out)
{
    const int64_t m = (int64_t)1 << prm->p;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        out[i] = hll_count_row(prm, (const uint8_t *)regs + i * m * elem_bytes, elem_bytes, NULL,
                               branch_out ? branch_out + i : NULL);
}

/* CSR-by-destination pull version of the propagation (same result as so_propagate_edges; used
 * for the multi-threaded CPU baseline and to cross-check the edge-scatter version).
 * n_self: rows i < n_self additionally receive their own row (implicit self loop). */
void so_propagate_csr(int64_t N, const int64_t *rowptr, const int32_t *col, int64_t n_self,
                      const uint32_t *mh_in, uint32_t *mh_out, int32_t P,
                      const uint8_t *hll_in, uint8_t *hll_out, int32_t M,
                      float *cards_out, int64_t cards_stride, const so_hll_params *prm)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        const int self = i < n_self;
        if (mh_out) {
            uint32_t *o = mh_out + i * P;
            if (e == b && !self) memset(o, 0, (size_t)P * 4);
            else {
                for (int32_t c = 0; c < P; ++c) o[c] = 0xFFFFFFFFu;
                if (self) for (int32_t c = 0; c < P; ++c) o[c] = mh_in[i * P + c];
                for (int64_t k = b; k < e; ++k) {
                    const uint32_t *x = mh_in + (int64_t)col[k] * P;
                    for (int32_t c = 0; c < P; ++c) o[c] = x[c] < o[c] ? x[c] : o[c];
                }
            }
        }
        if (hll_out) {
            uint8_t *o = hll_out + i * M;
            memset(o, 0, (size_t)M);
            if (self) memcpy(o, hll_in + i * M, (size_t)M);
            for (int64_t k = b; k < e; ++k) {
                const uint8_t *x = hll_in + (int64_t)col[k] * M;
                for (int32_t c = 0; c < M; ++c) o[c] = x[c] > o[c] ? x[c] : o[c];
            }
            if (cards_out) cards_out[i * cards_stride] = hll_count_row(prm, o, 1, NULL, NULL);
        }
    }
}

/* counting-sort CSR by destination; col gets the source ids.  rowptr [N+1], col [E]. */
void so_csr_build(int64_t N, int64_t E, const int64_t *src, const int64_t *dst, int64_t *rowptr, int32_t *col)
{
    memset(rowptr, 0, (size_t)(N + 1) * sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) rowptr[dst[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) rowptr[i + 1] += rowptr[i];
    int64_t *cur = (int64_t *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int64_t));
    memcpy(cur, rowptr, (size_t)N * sizeof(int64_t));
    for (int64_t e = 0; e < E; ++e) col[cur[dst[e]]++] = (int32_t)src[e];
    free(cur);
}

/* hashing.py:167-189 _get_intersections + :258-323 get_subgraph_features for B pairs.
 *   mh[k-1], hll[k-1]: hop-k tables (k = 1..h), cards [N, cards_stride] fp32.
 *   flags bit0 = use_zero_one, bit1 = floor_sf.
 *   out [B, h(h+2)] fp32.  Optional debug outputs (NULL to skip), all [B, h*h] row-major over
 *   (k1, k2): match counts, union zero counts, intersections J*U, estimator branch ids. */
static inline int64_t wrap_index(int64_t i, int64_t n) { return i < 0 ? i + n : i; }

void so_pair_features(const int64_t *links, int64_t B, int64_t N, int32_t h,
                      const uint32_t *const *mh, int32_t P, const uint8_t *const *hll,
                      const float *cards, int64_t cards_stride, const so_hll_params *prm, uint32_t flags,
                      float *out, int32_t *dbg_match, int32_t *dbg_zero, float *dbg_inter, int32_t *dbg_branch)
{
    const int64_t M = (int64_t)1 << prm->p;
    const int nf = h * (h + 2);
#pragma omp parallel
    {
        uint8_t *uni = (uint8_t *)malloc((size_t)M);
#pragma omp for schedule(static)
        for (int64_t q = 0; q < B; ++q) {
            const int64_t u = wrap_index(links[2 * q], N), v = wrap_index(links[2 * q + 1], N);
            float I[4][4] = {{0.0f}}; /* I[k1][k2], 1-based */
            for (int k1 = 1; k1 <= h; ++k1)
                for (int k2 = 1; k2 <= h; ++k2) {
                    const uint32_t *a = mh[k1 - 1] + u * P, *b = mh[k2 - 1] + v * P;
                    int32_t match = 0;
                    for (int32_t c = 0; c < P; ++c) match += (a[c] == b[c]);
                    const float jac = (float)match / (float)P;                 /* :256 */
                    const uint8_t *x = hll[k1 - 1] + u * M, *y = hll[k2 - 1] + v * M;
                    for (int64_t c = 0; c < M; ++c) uni[c] = x[c] > y[c] ? x[c] : y[c]; /* :237 */
                    int32_t zeros, br;
                    const float usz = hll_count_row(prm, uni, 1, &zeros, &br);  /* :186 */
                    I[k1][k2] = jac * usz;                                      /* :187 */
                    const int64_t d = q * h * h + (k1 - 1) * h + (k2 - 1);
                    if (dbg_match) dbg_match[d] = match;
                    if (dbg_zero) dbg_zero[d] = zeros;
                    if (dbg_inter) dbg_inter[d] = I[k1][k2];
                    if (dbg_branch) dbg_branch[d] = br;
                }
            const float *c1 = cards + u * cards_stride, *c2 = cards + v * cards_stride; /* :274 */
            float f[15];
            f[0] = I[1][1];
            if (h == 1) {                                   /* :277-279 */
                f[1] = c2[0] - f[0];
                f[2] = c1[0] - f[0];
            } else if (h == 2) {                            /* :280-288 */
                f[1] = I[2][1] - f[0];
                f[2] = I[1][2] - f[0];
                f[3] = I[2][2] - f[0] - f[1] - f[2];
                f[4] = c2[0] - (f[0] + f[1]);
                f[5] = c1[0] - f[0] - f[2];
                f[6] = c2[1] - ((((f[0] + f[4]) + f[1]) + f[2]) + f[3]);  /* torch.sum order over 5 strided floats, see note */
                f[7] = c1[1] - f[0] - (((f[0] + f[1]) + f[2]) + f[3]) - f[5];   /* f0 subtracted twice */
            } else {                                        /* :289-307 */
                f[1] = I[2][1] - f[0];
                f[2] = I[1][2] - f[0];
                f[3] = I[2][2] - f[0] - f[1] - f[2];
                f[4] = I[3][1] - f[0] - f[1];
                f[5] = I[1][3] - f[0] - f[2];
                const float s04 = ((f[0] + f[1]) + f[2]) + f[3];
                f[6] = I[3][2] - s04 - f[4];
                f[7] = I[2][3] - s04 - f[5];
                f[8] = I[3][3] - (((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]);
                f[9] = c2[0] - f[0] - f[1] - f[4];
                f[10] = c1[0] - f[0] - f[2] - f[5];
                const float s05 = (((f[0] + f[4]) + f[1]) + f[2]) + f[3];
                f[11] = c2[1] - s05 - f[6] - f[9];
                f[12] = c1[1] - s05 - f[7] - f[10];
                const float s09 = (((((((f[8] + f[0]) + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7];
                f[13] = c2[2] - s09 - f[9] - f[11];
                f[14] = c1[2] - s09 - f[10] - f[12];
            }
            if (!(flags & 1u)) {                            /* :310-318 */
                if (h == 2) { f[4] = 0.0f; f[5] = 0.0f; }
                else if (h == 3) { f[4] = 0.0f; f[5] = 0.0f; f[11] = 0.0f; f[12] = 0.0f; }
            }
            if (flags & 2u)                                 /* :319-320 */
                for (int k = 0; k < nf; ++k) if (f[k] < 0.0f) f[k] = 0.0f;
            for (int k = 0; k < nf; ++k) out[q * nf + k] = f[k];
        }
        free(uni);
    }
}

/* models/elph.py:276-293 BUDDY._append_degree_normalised (the first consumer of the feature rows):
 * out[q] = [x[q], x[q] / sqrt(deg[u] * deg[v])] with NaN and Inf replaced by 0.  x: [B, nf], out: [B, 2*nf]. */
void so_append_degree_normalised(const float *x, int64_t B, int32_t nf, const int64_t *links, int64_t N, const float *degrees,
                                 float *out)
{
    for (int64_t q = 0; q < B; ++q) {
        const int64_t u = wrap_index(links[2 * q], N), v = wrap_index(links[2 * q + 1], N);
        const float normaliser = sqrtf(degrees[u] * degrees[v]);
        for (int32_t k = 0; k < nf; ++k) {
            const float f = x[q * nf + k];
            float nrm = f / normaliser;
            if (isnan(nrm) || isinf(nrm)) nrm = 0.0f;
            out[q * 2 * nf + k] = f;
            out[q * 2 * nf + nf + k] = nrm;
        }
    }
}

/* heuristics.py:10-70 (CN / AA / RA; RA feeds HashDataset.RA, datasets/elph.py:76-77,314 -- SURVEY.md 8(f) N4):
 * score[q] = (float) sum_w A[u,w] * (A[v,w] * mult[w]) summed in fp64 in ascending column order, the order scipy's CSR
 * row sum visits the entries of A[src].multiply(A_[dst]).  CSR rows sorted and duplicate-free; val NULL = all ones,
 * mult NULL = all ones. */
void so_common_neighbour_scores(const int64_t *rowptr, const int32_t *col, const double *val, const double *mult, int64_t N,
                                const int64_t *links, int64_t B, float *out)
{
    for (int64_t q = 0; q < B; ++q) {
        const int64_t u = wrap_index(links[2 * q], N), v = wrap_index(links[2 * q + 1], N);
        int64_t i = rowptr[u], j = rowptr[v];
        const int64_t ie = rowptr[u + 1], je = rowptr[v + 1];
        double acc = 0.0;
        while (i < ie && j < je) {
            if (col[i] < col[j]) ++i;
            else if (col[i] > col[j]) ++j;
            else {
                const double a_src = val ? val[i] : 1.0, a_dst = val ? val[j] : 1.0;
                const double scaled = mult ? a_dst * mult[col[i]] : a_dst;
                acc += a_src * scaled;
                ++i;
                ++j;
            }
        }
        out[q] = (float)acc;
    }
}

/* torch_sparse.spmm as used by datasets/elph.py:87-110: out[row[e]] += val[e] * x[col[e]] for
 * e = 0..E-1 in edge order, fp32 product and fp32 add rounded separately (sequential CPU scatter_add). */
void so_spmm_coo(const int64_t *row, const int64_t *col, const float *val, int64_t E, int64_t N, const float *x, int32_t F, float *out)
{
    for (int64_t k = 0; k < N * F; ++k) out[k] = 0.0f;
    for (int64_t e = 0; e < E; ++e) {
        const float w = val[e];
        const float *src = x + col[e] * F;
        float *dst = out + row[e] * F;
        for (int32_t f = 0; f < F; ++f) {
            const float prod = src[f] * w;
            dst[f] += prod;
        }
    }
}
