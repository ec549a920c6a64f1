"""Every switch of the engine that is not an argument of the reference's API: module attributes, read at call time (tests and
measurement scripts set them: `ssa.knobs.DEFER_TABLE_HOP = False`), most initialised from an SS_* environment variable.
INTEGRATION.md section 5 lists them with defaults and scope; `subgraph_sketching_amd.hashing.<NAME>` reads through to here."""
import os

KERNEL_TIMER = None  # bench.py installs an object with record(name, stream) / span(name, start, end)

# Link sets of at least this many pairs are walked GROUPED BY THEIR FIRST NODE (ss_group_links_by_source + ss_pair_features_grouped):
# BUDDY's precompute hands get_subgraph_features every link of a split (reference datasets/elph.py:207-208) and every source
# occurs many times -- 120 times on average in ogbl-citation2's 356 M links -- so the rows of u are read once per GROUP instead
# of once per pair (they meet in the L1 / L2, or stay in registers).  The whole set is grouped at once (not chunk by chunk: a chunk
# of 11 M links holds a source 3.8 times, the set 120 times); `batch_size` then only bounds the pairs per launch.  Rows are
# bit-identical and in the caller's order.  The grouping costs ~35 ps per link (0.15 ms for 4 M links); SS_GROUP_LINKS_MIN=0 disables.
GROUP_LINKS_MIN = int(os.environ.get('SS_GROUP_LINKS_MIN', str(1 << 20)))
# from this many links on, the grouped query does not walk the order itself: every chunk's links are gathered first and its rows
# scattered afterwards by two streaming kernels (ss_gather_links / ss_scatter_feature_rows): random 16-byte reads and 60-byte
# writes over arrays of gigabytes from inside the query's latency chain cost it more than half its rate (csrc/ss_pairs.hip)
GROUP_GATHER_MIN = int(os.environ.get('SS_GROUP_GATHER_MIN', str(1 << 24)))



LAZY_MINHASH = True  # minhash_prop returns its int64 result as a LazyMinhash (materialised on first outside use)
# ELPH.forward (reference models/elph.py:209-212) calls hll_prop then minhash_prop per hop.  With this on, the hop-1 minhash_prop
# (input: an unmodified hop-0 tensor) only RECORDS its work; the hop-2 hll_prop that follows on the same edge_index computes the
# hop-1 MinHash rows together with its own HLL rows in one launch (ss_fused_hop_stage: the VALU-bound first hop under the
# memory-bound table hop).  Anything else that needs the table first (the next minhash_prop, get_subgraph_features, any torch
# operator on the tensor) triggers the ordinary first-hop launch.  Same results either way.
DEFER_FIRST_HOP = os.environ.get('SS_FUSED_STAGE', '1') != '0'
# Deferred table hop: `minhash_prop` on any other input only RECORDS the hop as well.  The next consumer decides how much of it is
# computed: another `minhash_prop` (or any torch operator, torch.save ...) needs the whole table; `get_subgraph_features` reads two
# rows per link, and ELPH's training step (models/elph.py:209-212, runners/train.py:204) queries ONE batch after every full-graph
# propagation -- the rows of that batch are computed through ss_minhash_hop_rows (2 B rows instead of N; same values) and the
# table stays owed for everybody else.  Batches on the same table whose rows add up to more than N make it complete instead.
DEFER_TABLE_HOP = os.environ.get('SS_DEFER_TABLE_HOP', '1') != '0'



# Rows with more in-edges than the hub threshold are propagated by a 16-wave workgroup instead of one wavefront (MinHash) /
# one 16-lane group (HLL).  None = adaptive: a single wavefront walking d neighbour rows takes ~0.35 us * d, which must
# stay well below the whole hop (~E * 40 ps): d <= E / 16384, clamped to [128, 1024].  Measured (power-law endpoints,
# alpha 0.5): collab size 0.754 -> 0.674 ms per step with 144 instead of 512; ppa size flat between 512 and 2048 and 11 %
# slower at 128 (too many rows on the cooperative path).  SS_HUB_THRESHOLD / this constant force a value.
HUB_THRESHOLD = int(os.environ['SS_HUB_THRESHOLD']) if 'SS_HUB_THRESHOLD' in os.environ else None



# ELPH.forward builds a fresh self-looped edge_index every step (reference models/elph.py:186): the CSR cache below is keyed on the
# tensor OBJECT, so that step rebuilt an identical CSR every time.  With this on, a cache miss on a tensor of the cached shape goes
# through ss_csr_build_cached: one streaming pass over the edges + a device-side comparison; the build only runs if the edges differ.
REUSE_CSR_BY_CONTENT = os.environ.get('SS_REUSE_CSR', '1') != '0'



# largest hop-1 HLL table (bytes) ss_fused_hop_stage is used for.  The first version of the stage lost on tables that do not fit
# the 256 MiB Infinity Cache (citation2-like: 4.40 against 4.09 ms for the two launches) and was capped there; with LDS landings
# and batched tail walks it wins there too (3.70 against 3.92 ms), so there is no cap any more.  SS_FUSED_STAGE_MAX_MB:
# measurement hook
FUSED_STAGE_MAX_TABLE_BYTES = int(os.environ.get('SS_FUSED_STAGE_MAX_MB', str(1 << 30))) << 20
