"""Byte model of the hot path: the algorithmic bytes every kernel family moves per launch under the IMPLEMENTED schedule
(DESIGN.md section 3 states each formula), the bytes a step would move under SURVEY.md 8(d)'s per-hop definition, and
where a launch's working set lives.  Pure arithmetic -- used by bench.py for the `roofline` / `step_roofline` objects and
checked by tests/test_host_logic.py; nothing here touches a device.

Symbols: N nodes, E directed edges as given (edge_index columns), E' = E + N (the reference appends one self loop per node,
hashing.py:148; here they are implicit: `col` holds E entries, the walk visits E' rows), P MinHash permutations (u32 each),
M = 2^p HLL registers (u8 each), R = 4P + M bytes per sketch row, h hops, B pairs per query batch.
"""

HBM_PEAK_GBS = 8000.0           # MI355X spec (MI355X_MICROARCH.md); a float4 copy reaches 6.29 TB/s (79 %)
INFINITY_CACHE_BYTES = 256 << 20  # MALL / L3; FETCH_SIZE counts fabric requests INCLUDING the ones it serves


def csr_bytes(N, E):
    """ss_csr_build, ALGORITHMIC bytes: the edge list read once (16E), col (4E) and rowptr (8(N+1)) written once.  The level
    plans move more than that (one level 16E + 4E + 4E + 4E, two levels 16E + 8E + 8E + 4E + 4E + 4E); the
    roofline fraction of the build is taken on the algorithmic figure."""
    return 20 * E + 8 * (N + 1)


# random whole-row gathers of 512-byte rows against the size of the table they come from: tools/micro/gather_ceiling.hip on an
# MI355X (profiles/round4_gather_ceiling.txt; best of its in-flight / grid / rows-per-item settings), with the 4-byte id the probe
# reads for every gathered row COUNTED (the tracked table counts rows only: x 516 / 512) -- a row kernel's algorithmic bytes
# count its ids too.  Infinity-Cache resident below ~256 MB, HBM resident above.  A reference point for the access pattern, not a
# bound: a kernel also streams finished rows out, which travel faster than gathers.
GATHER_PROBE_GBS = [(30e6, 8730.0 * 516 / 512), (60e6, 8120.0 * 516 / 512), (120e6, 7840.0 * 516 / 512), (200e6, 7700.0 * 516 / 512),
                    (295e6, 7650.0 * 516 / 512), (600e6, 7550.0 * 516 / 512), (1500e6, 7090.0 * 516 / 512)]


def gather_probe_gbs(table_bytes):
    """measured rate (GB/s) of random row gathers from a table of this size: log-linear interpolation of GATHER_PROBE_GBS"""
    import math
    pts = GATHER_PROBE_GBS
    if table_bytes <= pts[0][0]:
        return pts[0][1]
    if table_bytes >= pts[-1][0]:
        return pts[-1][1]
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if x0 <= table_bytes <= x1:
            t = (math.log(table_bytes) - math.log(x0)) / (math.log(x1) - math.log(x0))
            return y0 + t * (y1 - y0)
    return pts[-1][1]


def graph_read_bytes(N, E):
    """what every propagation launch reads of the CSR: col (4E) + rowptr (8(N+1))"""
    return 4 * E + 8 * (N + 1)


def kernel_bytes(N, E, P=128, p=8, h=2, B=65536, hub_edges=0, hub_rows=0, hosted=True):
    """algorithmic bytes per LAUNCH of each kernel family of one step (build_hash_tables + one query batch).
    hub_edges / hub_rows: in-edges and count of the rows above the hub threshold.  The row wavefronts SKIP those rows; they are
    walked as hub units (csrc/ss_hub.hpp) by the leading workgroups of -- `hosted`, the default since round 4 -- the HLL first-hop
    launch (both hop-1 tables), the MinHash table-hop launch of hop 2 (both hop-2 tables: the fused kernel hosts none) and, for
    hops >= 3, the launch of their own sketch; their bytes are those launches' (`minhash_hop` is the MEAN over the h - 1 MinHash
    table-hop launches of a build).  hosted=False: launches of their own (`hub_first_hop` / `hub_table_hop`, one per hop, both
    sketches; rounds 1-3 and SS_HUB_LAUNCHES=1).  Crediting hub rows to a launch that skips them gave a roofline fraction above
    1 on skewed graphs."""
    M = 1 << p
    R = 4 * P + M
    Er, Nr = E - hub_edges, N - hub_rows      # what the row wavefronts walk / write
    Epr = Er + Nr                              # + one implicit self loop per written row
    graph = 4 * Er + 8 * (N + 1)               # col entries of the walked rows + rowptr
    hub_graph = 4 * hub_edges + 20 * hub_rows + 4 * hub_rows   # one pass over the hub rows: col + two rowptr words + list entry
    hub_tbl = hub_edges + 2 * hub_rows                         # table rows a table-hop pass reads (+ self loop) and writes
    out = {
        'csr_build': csr_bytes(N, E),
        # hop 1 from node ids: no table reads (hop-0 rows are recomputed in registers)
        'first_hop_hll': graph + Nr * M + 4 * Nr,                          # writes the HLL rows + cards[:, 0]
        'first_hop_minhash': graph + Nr * 4 * P,                           # writes the MinHash rows
        # table hops (k = 2..h): one input row per edge and self loop, one output row per node
        'hll_hop': (Epr + Nr) * M + graph + 4 * Nr,                        # + cards[:, k-1]
        'minhash_hop': (Epr + Nr) * 4 * P + graph,
        'pair_features': B * pair_bytes(P, p, h),
        # ss_fused_hop_stage's kernel: MinHash first hop + HLL table hop of hop 2 in one launch (the CSR is read once)
        'fused_first_hop_hll_hop': graph + Nr * 4 * P + (Epr + Nr) * M + 4 * Nr,
        # hub units as launches of their own: both sketches of the hub / mega rows of one hop
        'hub_first_hop': hub_graph + hub_rows * R,
        'hub_table_hop': hub_tbl * R + hub_graph,
    }
    if hosted and hub_rows:
        out['first_hop_hll'] += 2 * hub_graph + hub_rows * R + 4 * hub_rows          # a pass per sketch over the hub rows' ids
        hop2 = 2 * hub_graph + hub_tbl * R + 4 * hub_rows                            # MinHash launch of hop 2: both hop-2 tables
        later = hub_graph + hub_tbl * 4 * P                                          # hops >= 3: the MinHash units alone
        out['minhash_hop'] += (hop2 + (h - 2) * later) // max(h - 1, 1)
        out['hll_hop'] += hub_graph + hub_tbl * M + 4 * hub_rows                     # (hops >= 3; hop 2's HLL rows are the fused kernel's)
        out['hub_first_hop'] = out['hub_table_hop'] = 0
    return out


def hub_split(in_degree, hub_threshold):
    """(hub_edges, hub_rows) of a graph from its in-degree array (numpy) -- the arguments kernel_bytes takes"""
    hub = in_degree > hub_threshold
    return int(in_degree[hub].sum()), int(hub.sum())


def minhash_rows_bytes(N, E, n_rows, P=128):
    """ss_minhash_hop_rows over n_rows rows drawn uniformly: per row its (E/N + 1) neighbour rows (self loop included) and its own
    output row of 4P bytes, its col entries and two rowptr words, the int64 row id"""
    return int(n_rows * ((E / N + 2) * 4 * P + 4 * E / N + 16 + 8))


def pair_bytes(P=128, p=8, h=2):
    """per pair: 2h sketch rows + the two int64 ids + 2h cardinalities + h(h+2) fp32 features"""
    return 2 * h * (4 * P + (1 << p)) + 16 + 8 * h + 4 * h * (h + 2)


def pair_bytes_grouped(pairs, runs, P=128, p=8, h=2):
    """bytes of a query over `pairs` links walked grouped by their first node (ss_pair_features_grouped, hashing.GROUP_LINKS_MIN):
    the first node's h rows are fetched once per RUN of pairs that share it (`runs` = distinct first nodes of a grouped list),
    everything else per pair as in pair_bytes -- the bytes the implemented walk has to move; pairs * pair_bytes() is the
    SURVEY 8(d) definition, which charges 2h rows to every pair"""
    R = 4 * P + (1 << p)
    return pairs * (h * R + 16 + 8 * h + 4 * h * (h + 2)) + runs * h * R


def step_bytes_implemented(N, E, P=128, p=8, h=2, B=65536, hub_edges=0, hub_rows=0):
    """bytes of one step under the implemented schedule: CSR build, hop 1 from node ids, h - 1 table hops, one query batch
    (the hub units included: the same rows and edges, walked by other workgroups)"""
    k = kernel_bytes(N, E, P, p, h, B, hub_edges, hub_rows, hosted=False)
    return (k['csr_build'] + k['first_hop_hll'] + k['first_hop_minhash'] + k['hub_first_hop']
            + (h - 1) * (k['hll_hop'] + k['minhash_hop'] + k['hub_table_hop']) + k['pair_features'])


def step_bytes_survey(N, E, P=128, p=8, h=2, B=65536):
    """h table hops of (E'+N)R + 4E' + 8(N+1) + 4N bytes + the query; CSR construction and hop-0
    initialisation reported separately there.  Larger than the implemented schedule's bytes because hop 1 no longer reads
    a table -- a step faster than this figure / 8 TB/s is therefore not a measurement error."""
    R, Ep = 4 * P + (1 << p), E + N
    return h * ((Ep + N) * R + 4 * Ep + 8 * (N + 1) + 4 * N) + B * pair_bytes(P, p, h)


def unique_bytes(N, E, family, P=128, p=8):
    """distinct HBM bytes a launch touches (each input row counted once however many edges read it): what the HBM
    itself must deliver when the input table fits the Infinity Cache"""
    M = 1 << p
    row = {'minhash_hop': 4 * P, 'hll_hop': M}[family]
    return 2 * N * row + graph_read_bytes(N, E) + (4 * N if family == 'hll_hop' else 0)


def resident_label(table_bytes):
    """where a gathered table lives, from the share of it the 256 MiB Infinity Cache can hold: all of it -> 'infinity-cache' (its
    random row reads are served there, `achieved` is a fabric + cache rate), at least half -> 'mixed', less -> 'hbm'.  ONE rule for
    bench.py's line and for tools/roofline_table.py (round 5's tables called a 74-91 % cached table 'hbm': VERDICT r5 weak #9b)."""
    f = cache_resident_fraction(table_bytes)
    return 'infinity-cache' if f >= 1.0 else ('mixed' if f >= 0.5 else 'hbm')


def residency(N, family, P=128, p=8, h=2):
    """resident_label of the table(s) a kernel family gathers rows from"""
    return resident_label(gathered_table_bytes(N, family, P, p, h))


def cache_resident_fraction(table_bytes):
    """share of a gathered table set the 256 MiB Infinity Cache can hold: min(1, 256 MiB / bytes)"""
    return 1.0 if table_bytes <= 0 else min(1.0, INFINITY_CACHE_BYTES / float(table_bytes))


def gathered_table_bytes(N, family, P=128, p=8, h=2):
    """bytes of the table(s) a kernel family gathers rows from"""
    M = 1 << p
    return {'minhash_hop': N * 4 * P, 'hll_hop': N * M, 'fused_first_hop_hll_hop': N * M, 'pair_features': h * N * (4 * P + M)}[family]
