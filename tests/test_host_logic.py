"""Host-side logic of the ElphHashes mirror that needs no GPU: constants, constructor contract, parameter
derivation, containers, sharding arithmetic -- and that compute entry points refuse to run without a HIP device."""
import os
import pickle
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import load_golden

no_gpu = not torch.cuda.is_available()


def _args(h=2, p=8, P=128, floor_sf=False, use_zero_one=True):
    return Namespace(max_hash_hops=h, hll_p=p, minhash_num_perm=P, floor_sf=floor_sf, use_zero_one=use_zero_one)


@pytest.fixture(scope='module')
def ssa():
    import subgraph_sketching_amd as m
    return m


def test_label_lookup(ssa):
    for key, val in ssa.LABEL_LOOKUP.items():  # reference test_hashing.py:196-198
        assert len(val) == key * (key + 2)
    assert ssa.LABEL_LOOKUP[2][7] == (2, 0) and ssa.LABEL_LOOKUP[3][4] == (3, 1) and ssa.LABEL_LOOKUP[3][12] == (2, 0)


def test_constructor_contract(ssa):
    eh = ssa.ElphHashes(_args())
    assert eh.max_hops == 2 and eh.num_perm == 128 and eh.p == 8 and eh.m == 256 and eh.hll_size == 256
    assert eh.max_rank == 56 and int(eh._max_minhash) == 2 ** 32 - 1 and int(eh._mersenne_prime) == 2 ** 61 - 1
    assert eh.minhash_seed == 1 and eh.label_lookup is ssa.LABEL_LOOKUP[2] and eh.hll_threshold == 220
    assert abs(eh.alpha - 0.7182725932495458) < 1e-15
    assert eh.bias_vector.dtype == torch.float32 and eh.estimate_vector.shape == eh.bias_vector.shape
    assert callable(eh.hll_prop) and callable(eh.minhash_prop)
    for bad in (0, 4):
        with pytest.raises(AssertionError):
            ssa.ElphHashes(_args(h=bad))
    eh3 = ssa.ElphHashes(_args(h=3, p=4, P=8))
    assert eh3.m == 16 and eh3.max_rank == 60 and eh3.hll_threshold == 10 and eh3.alpha == 0.673


def test_bit_length_and_rank(ssa):
    eh = ssa.ElphHashes(_args())
    arr = np.arange(1000)
    for bl, elem in zip(eh._np_bit_length(arr), arr):  # reference test_hashing.py:331-336
        assert int(elem).bit_length() == bl
    big = np.array([2 ** 53 - 1, 2 ** 53, 2 ** 53 + 1, 2 ** 55 + 1, 2 ** 56 - 1], dtype=np.uint64)
    assert list(eh._np_bit_length(big)) == [int(x).bit_length() for x in big]
    assert list(eh._get_hll_rank(np.array([0, 1, 2 ** 55], dtype=np.uint64))) == [57, 56, 1]
    with pytest.raises(ValueError):
        eh._get_hll_rank(np.array([2 ** 56], dtype=np.uint64))


def test_permutations_match_reference(ssa):
    g = load_golden('g1_g2_init.npz')
    for P in (8, 128):
        ab = ssa.ElphHashes(_args(P=P))._init_permutations(P)
        assert ab.dtype == np.uint64 and ab.shape == (2, P)
        assert np.array_equal(ab[0], g[f'perm_a_P{P}']) and np.array_equal(ab[1], g[f'perm_b_P{P}'])


def test_linear_counting_table_and_threshold(ssa):
    lc = ssa.hashing.linear_counting_table(256).numpy()
    assert lc.shape == (257,) and lc[256] == 0.0 and np.all(np.diff(lc[1:]) < 0)
    assert abs(lc[110] - 216.24244689941406) < 1e-4 and abs(lc[109] - 218.58035278320312) < 1e-4
    ok = lc[1:] <= np.float32(220)
    assert int(np.argmax(ok)) + 1 == 109  # p=8: linear counting iff V >= 109 zero registers


def test_hll_tables_provider(ssa):
    for p in (4, 8, 16):
        t = ssa.hll_tables.load(p, prefer='regenerated')
        assert t.provenance == 'regenerated' and t.max_rank == 64 - p
        assert 6 <= len(t.raw_estimate) <= 512 and len(t.raw_estimate) == len(t.bias)
        assert np.all(np.diff(t.raw_estimate) >= 0)
        m = 1 << p
        assert abs(t.raw_estimate[0] - t.alpha * m) < 1e-6 * m    # zero items: raw estimate = alpha*m
        assert abs(t.bias[0] - t.raw_estimate[0]) < 1e-9
        assert 4.9 * m < t.raw_estimate[-1] < 5.4 * m
    with pytest.raises(ValueError):
        ssa.hll_tables.load(3)
    assert ssa.hll_tables.load(8).provenance in ('datasketch', 'regenerated')


def test_shard_bounds(ssa):
    for L in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [ssa.dist.shard_bounds(L, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == L
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) <= (L + world - 1) // world


def test_engine_pickles_without_device_state(ssa):
    eh = ssa.ElphHashes(_args())
    eh._dev_params['cuda:0'] = object()  # would not pickle
    clone = pickle.loads(pickle.dumps(eh))
    assert clone.max_hops == 2 and clone._dev_params == {} and callable(clone.hll_prop)


def test_thin_torch_helpers(ssa):
    eh = ssa.ElphHashes(_args())
    a = torch.tensor([[1, 2, 3, 4]] * 2)
    assert torch.equal(eh._hll_merge(a, a + 1), a + 1)
    with pytest.raises(ValueError):
        eh._hll_merge(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises(ValueError):
        eh.jaccard(torch.zeros(2, 4), torch.zeros(2, 5))
    eh.num_perm = 4
    assert torch.allclose(eh.jaccard(a, torch.tensor([[1, 2, 0, 0], [1, 2, 3, 4]])), torch.tensor([0.5, 1.0]))
    root, nb = torch.tensor([3, 1]), torch.tensor([[2, 5], [4, 0]])
    assert torch.equal(eh.hll_neighbour_merge(root, nb), torch.tensor([4, 5]))
    assert torch.equal(eh.minhash_neighbour_merge(root, nb), torch.tensor([2, 0]))


@pytest.mark.skipif(not no_gpu, reason='only meaningful on a GPU-less host')
def test_no_cpu_fallback(ssa):
    """the product path must fail loudly when there is no HIP device -- never compute on the CPU"""
    eh = ssa.ElphHashes(_args())
    ei = torch.tensor([[0, 1], [1, 0]])
    for call in (lambda: eh.initialise_minhash(4), lambda: eh.initialise_hll(4), lambda: eh.build_hash_tables(2, ei),
                 lambda: eh.hll_count(torch.zeros(2, 256, dtype=torch.int8)),
                 lambda: eh.hll_prop(torch.zeros(2, 256, dtype=torch.int8), ei),
                 lambda: eh.minhash_prop(torch.zeros(2, 128, dtype=torch.int64), ei),
                 lambda: eh.get_subgraph_features(torch.tensor([[0, 1]]), {1: {}, 2: {}}, torch.zeros(2, 2))):
        with pytest.raises(RuntimeError, match='HIP device'):
            call()


def test_unsupported_sizes_raise(ssa):
    with pytest.raises(NotImplementedError):
        ssa.hashing._check_sizes(6, 8)
    with pytest.raises(NotImplementedError):
        ssa.hashing._check_sizes(128, 17)


def test_datasketch_tables_are_used_when_importable(ssa, monkeypatch):
    """with `datasketch` importable its objects are used verbatim (provenance 'datasketch'); simulated here with a
    stand-in module carrying recognisable values"""
    import sys
    import types
    ds = types.ModuleType('datasketch')
    const = types.ModuleType('datasketch.hyperloglog_const')
    const._thresholds = [10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000]
    const._raw_estimate = [[float(100 + 7 * i) for i in range(80 + k)] for k in range(15)]
    const._bias = [[float(i % 5) for i in range(80 + k)] for k in range(15)]

    class HyperLogLogPlusPlus(object):
        def __init__(self, p=8):
            self.alpha, self.max_rank, self.reg = 0.5, 64 - p, np.zeros(1 << p, dtype=np.int8)

    ds.HyperLogLogPlusPlus, ds.hyperloglog_const = HyperLogLogPlusPlus, const
    monkeypatch.setitem(sys.modules, 'datasketch', ds)
    monkeypatch.setitem(sys.modules, 'datasketch.hyperloglog_const', const)
    t = ssa.hll_tables.load(8)
    assert t.provenance == 'datasketch' and t.alpha == 0.5 and t.threshold == 220.0 and len(t.raw_estimate) == 84
    eh = ssa.ElphHashes(_args())
    assert eh.hll_tables.provenance == 'datasketch' and eh.alpha == 0.5 and eh.estimate_vector.shape == (84,)
    assert ssa.hll_tables.load(8, prefer='regenerated').provenance == 'regenerated'


def test_adaptive_hub_threshold():
    """E / 16384 clamped to [128, 1024]: the shapes of BASELINE.json's configs"""
    from subgraph_sketching_amd.hashing import default_hub_threshold
    assert default_hub_threshold(0) == 128 and default_hub_threshold(10138) == 128            # Cora
    assert default_hub_threshold(2358104 + 235868) == 158 and default_hub_threshold(2358104) == 143   # collab with / without loops
    assert default_hub_threshold(42463862) == 1024 and default_hub_threshold(60775990) == 1024   # ppa, citation2
    assert all(128 <= default_hub_threshold(e) <= 1024 for e in (1, 10 ** 5, 10 ** 7, 10 ** 10))


def test_abi_constants_match_the_header():
    import re
    from conftest import REPO
    import os
    from subgraph_sketching_amd import _native
    text = open(os.path.join(REPO, 'include', 'subgraph_sketch.h')).read()
    assert int(re.search(r'#define SS_MEGA_SLICE (\d+)', text).group(1)) == _native.MEGA_SLICE
    assert int(re.search(r'#define SS_MEGA_SLOT_BYTES (\d+)', text).group(1)) == _native.MEGA_SLOT_BYTES
    assert int(re.search(r'#define SS_CSR_FINGERPRINT_BYTES (\d+)', text).group(1)) == _native.CSR_FINGERPRINT_BYTES
    assert int(re.search(r'#define SS_MAX_MIRRORS (\d+)', text).group(1)) == _native.MAX_MIRRORS
    api = open(os.path.join(REPO, 'subgraph-sketching_amd', 'csrc', 'ss_api.hip')).read()
    assert int(re.search(r'ss_version\(void\) \{ return (\d+);', api).group(1)) == _native.ABI_VERSION
    fields = re.search(r'typedef struct ss_csr_graph \{(.*?)\} ss_csr_graph;', text, re.S).group(1)
    names = re.findall(r'(\w+)(?:\[\d+\])?;', re.sub(r'/\*.*?\*/', '', fields, flags=re.S))   # (array members: name[7];)
    assert names == [f[0] for f in _native.CsrGraphStruct._fields_]   # the ctypes mirror lists the same fields in the same order


def test_regenerated_tables_warn_and_are_identified(ssa, caplog, monkeypatch):
    """the silent fall-back to simulated HLL++ tables now logs a warning (once per precision), and every table
    has an identity (provenance + digest of the numbers) that travels with the cardinalities it produced"""
    import logging
    ht = ssa.hll_tables
    monkeypatch.setattr(ht, '_warned', set())
    monkeypatch.setattr(ht, 'EXPORTED', '/nonexistent/hllpp_tables_datasketch.npz')
    try:
        import datasketch  # noqa: F401
        pytest.skip('datasketch is importable here: the fall-back is not taken')
    except ImportError:
        pass
    with caplog.at_level(logging.WARNING, logger=ht.logger.name):
        t = ht.load(8)
        ht.load(8)
    assert t.provenance == 'regenerated'
    assert sum('REGENERATED' in r.getMessage() for r in caplog.records) == 1
    tid = ht.table_id(t)
    assert tid.startswith('regenerated:') and tid == ht.table_id(ht.load(8, prefer='regenerated'))
    other = t._replace(bias=t.bias + 1e-3)
    assert ht.table_id(other) != tid and ht.table_id(t._replace(provenance='datasketch')) != tid
    # ... but the same NUMBERS under another provenance label are the same tables for every compatibility check:
    # a cache built next to the datasketch package loads next to its shipped export
    assert ht.same_tables(ht.table_id(t._replace(provenance='datasketch')), ht.table_id(t._replace(provenance='datasketch-export')))
    assert ht.same_tables(tid, tid) and not ht.same_tables(ht.table_id(other), tid) and not ht.same_tables(None, tid)
    eh = ssa.ElphHashes(_args())
    assert eh.tables_id == tid
    eh.hll_tables = other
    assert eh.tables_id == ht.table_id(other)  # recomputed when the tables are replaced


def test_exported_datasketch_tables_are_preferred_over_regenerated(ssa, tmp_path, monkeypatch):
    """data/hllpp_tables_datasketch.npz (written by tools/export_datasketch_fixture.py where datasketch is installed) is
    used when the package itself is absent; the exporter's file format round-trips through hll_tables.load"""
    import importlib.util
    from conftest import REPO
    import os
    ht = ssa.hll_tables
    try:
        import datasketch  # noqa: F401
        pytest.skip('datasketch is importable here')
    except ImportError:
        pass
    spec = importlib.util.spec_from_file_location('export_ds', os.path.join(REPO, 'tools', 'export_datasketch_fixture.py'))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    fake = {p: ht.load(p, prefer='regenerated')._replace(alpha=0.5 + p / 100.0) for p in (4, 8)}
    path = tmp_path / 'hllpp_tables_datasketch.npz'
    tool.write_tables(str(path), fake)
    monkeypatch.setattr(ht, 'EXPORTED', str(path))
    t = ht.load(8)
    assert t.provenance == 'datasketch-export' and t.alpha == 0.58 and np.array_equal(t.bias, fake[8].bias)
    assert ht.load(6).provenance == 'regenerated'  # a precision the export does not hold
    with pytest.raises(ImportError):
        monkeypatch.setattr(ht, 'EXPORTED', '/nonexistent.npz')
        ht.load(8, prefer='datasketch')


def test_byte_model_of_the_step(ssa):
    """subgraph_sketching_amd/roofline.py: the figures bench.py's roofline objects are built from"""
    rf = ssa.roofline
    n, e = 235868, 2358104                      # the bench graph (BASELINE configs[1])
    k = rf.kernel_bytes(n, e, 128, 8, 2, 65536)
    ep = e + n
    assert k['minhash_hop'] == (ep + n) * 512 + 4 * e + 8 * (n + 1)
    assert k['hll_hop'] == (ep + n) * 256 + 4 * e + 8 * (n + 1) + 4 * n
    assert k['pair_features'] == 65536 * 3136 and rf.pair_bytes(128, 8, 3) == 4708 and rf.pair_bytes(128, 8, 1) == 1572
    # grouped walk: the first node's h rows once per run -- one run per pair is the per-pair definition again
    assert rf.pair_bytes_grouped(1000, 1000, 128, 8, 2) == 1000 * 3136
    assert rf.pair_bytes_grouped(1000, 10, 128, 8, 2) == 1000 * (3136 - 2 * 768) + 10 * 2 * 768
    assert abs(k['minhash_hop'] - 1.4611e9) < 5e6                   # 1.461 GB per launch
    survey = rf.step_bytes_survey(n, e, 128, 8, 2, 65536)
    assert abs(survey - (2 * 2.187e9 + 0.2055e9)) < 2e7             # 4.58 GB per step by SURVEY 8(d)
    impl = rf.step_bytes_implemented(n, e, 128, 8, 2, 65536)
    assert impl < survey and abs(impl - 2.66e9) < 0.1e9             # hop 1 reads no table, the CSR build at its algorithmic 20E + 8N: ~2.66 GB per step
    assert rf.residency(n, 'minhash_hop') == 'infinity-cache' and rf.residency(2927963, 'minhash_hop') == 'hbm'
    assert rf.residency(576289, 'minhash_hop') == 'mixed' and rf.residency(576289, 'hll_hop') == 'infinity-cache'  # (295 MB: 87 % of it fits the cache)
    assert rf.residency(2927963, 'pair_features', h=3) == 'hbm' and rf.resident_label(300 << 20) == 'mixed'
    assert rf.unique_bytes(n, e, 'minhash_hop') == 2 * n * 512 + 4 * e + 8 * (n + 1)
    assert rf.csr_bytes(n, e) == 20 * e + 8 * (n + 1)               # algorithmic: edge list read once, col + rowptr written once
    ids = 516 / 512  # (the probe's id bytes counted)
    assert rf.gather_probe_gbs(120e6) == 7840.0 * ids and 7090.0 * ids < rf.gather_probe_gbs(1e9) < 7550.0 * ids and rf.gather_probe_gbs(1e7) == 8730.0 * ids
    # ss_minhash_hop_rows over all N rows moves what the full table hop moves (+ an 8-byte row id and a second rowptr word per
    # listed row), over one ELPH batch 1.7 % of it
    assert abs(rf.minhash_rows_bytes(n, e, n) - (k['minhash_hop'] + 16 * n)) < 1e5
    assert abs(rf.minhash_rows_bytes(n, e, 4096) / k['minhash_hop'] - 4096 / n) < 1e-3
    # skewed graphs: the rows above the hub threshold are walked by the hub passes -- their bytes are NOT the row kernels'
    # (a fraction of 1.305 came from crediting them to propagate_kernel); nothing is lost or counted twice
    import numpy as np
    deg = np.full(n, 10, dtype=np.int64)
    deg[:50] = 20000
    eh_, nh_ = rf.hub_split(deg, 144)
    assert (eh_, nh_) == (50 * 20000, 50)
    e2 = int(deg.sum())
    ks = rf.kernel_bytes(n, e2, 128, 8, 2, 65536, eh_, nh_, hosted=False)  # hub units as launches of their own (rounds 1-3)
    k0 = rf.kernel_bytes(n, e2, 128, 8, 2, 65536)
    assert ks['minhash_hop'] == (e2 - eh_ + 2 * (n - nh_)) * 512 + 4 * (e2 - eh_) + 8 * (n + 1) < k0['minhash_hop']
    assert k0['hub_table_hop'] == 0 and k0['hub_first_hop'] == 0
    assert ks['hub_table_hop'] == (eh_ + 2 * nh_) * 768 + 4 * eh_ + 24 * nh_
    moved = (k0['minhash_hop'] + k0['hll_hop']) - (ks['minhash_hop'] + ks['hll_hop'])
    # (one hub pass serves both sketches: it reads the rows' col entries once where the two row kernels read them twice; per row
    # it also reads its list entry and both rowptr words)
    assert abs(moved - 4 * eh_ - ks['hub_table_hop']) <= 32 * nh_
    assert abs(rf.step_bytes_implemented(n, e2, 128, 8, 2, 65536, eh_, nh_) + 8 * eh_ - rf.step_bytes_implemented(n, e2, 128, 8, 2, 65536)) <= 64 * nh_
    # hub units hosted by the row launches (round 4, the default): no hub family, the hosting launches carry the hub rows' bytes --
    # the HLL first-hop launch both hop-1 tables, the MinHash table hop both hop-2 tables; the fused kernel hosts none
    kh = rf.kernel_bytes(n, e2, 128, 8, 2, 65536, eh_, nh_)
    assert kh['hub_table_hop'] == 0 and kh['hub_first_hop'] == 0
    assert kh['fused_first_hop_hll_hop'] == ks['fused_first_hop_hll_hop'] and kh['first_hop_minhash'] == ks['first_hop_minhash']
    assert kh['minhash_hop'] - ks['minhash_hop'] == ks['hub_table_hop'] + 4 * eh_ + 24 * nh_ + 4 * nh_  # + a second pass over the ids, cards
    assert kh['first_hop_hll'] - ks['first_hop_hll'] == ks['hub_first_hop'] + 4 * eh_ + 24 * nh_ + 4 * nh_
    k3, s3 = rf.kernel_bytes(n, e2, 128, 8, 3, 65536, eh_, nh_), rf.kernel_bytes(n, e2, 128, 8, 3, 65536, eh_, nh_, hosted=False)
    later = 4 * eh_ + 24 * nh_ + (eh_ + 2 * nh_) * 512          # hops >= 3: the MinHash launch hosts the MinHash units alone
    assert k3['minhash_hop'] - s3['minhash_hop'] == ((kh['minhash_hop'] - ks['minhash_hop']) + later) // 2  # mean over the two launches
    assert k3['hll_hop'] - s3['hll_hop'] == 4 * eh_ + 24 * nh_ + (eh_ + 2 * nh_) * 256 + 4 * nh_
    # residency as a fraction: the label is a threshold, the number is what to read
    assert rf.cache_resident_fraction(rf.gathered_table_bytes(n, 'minhash_hop')) == 1.0
    assert abs(rf.cache_resident_fraction(rf.gathered_table_bytes(576289, 'minhash_hop')) - 0.9097) < 1e-3
    assert abs(rf.cache_resident_fraction(rf.gathered_table_bytes(2927963, 'pair_features', h=3)) - 0.0398) < 1e-3


def test_batch_plan_bookkeeping(ssa):
    """weak: every rank its own batch; strong: one global batch cut into contiguous slices that tile it"""
    BatchPlan = ssa.dist.BatchPlan
    for batch in (65536, 131072, 261424, 7, 1):
        for world in (1, 2, 4, 8):
            plans = [BatchPlan('strong', world, r, batch) for r in range(world)]
            assert plans[0].lo == 0 and plans[-1].hi == batch and all(a.hi == b.lo for a, b in zip(plans, plans[1:]))
            assert all(p.pairs_per_step == batch and p.links_seed == 2 for p in plans)
            assert all(p.local_pairs <= p.rows_per_rank for p in plans) and world * plans[0].rows_per_rank >= batch
            weak = [BatchPlan('weak', world, r, batch) for r in range(world)]
            assert all(p.pairs_per_step == world * batch and p.local_pairs == batch for p in weak)
            assert len({p.links_seed for p in weak}) == world
    with pytest.raises(ValueError):
        BatchPlan('medium', 2, 0, 8)


def _load_shim(monkeypatch, tmp_path, env=None):
    """integration/src/hashing.py copied to a foreign location (as it would sit in a reference checkout) and imported"""
    import importlib.util
    import shutil
    from conftest import REPO
    import os
    dst = tmp_path / 'src'
    dst.mkdir(exist_ok=True)
    shutil.copy(os.path.join(REPO, 'integration', 'src', 'hashing.py'), dst / 'hashing.py')
    monkeypatch.setenv('SUBGRAPH_SKETCH_AMD_ROOT', REPO)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    spec = importlib.util.spec_from_file_location('shim_src_hashing', str(dst / 'hashing.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_integration_shim_exports_the_reference_surface(ssa, monkeypatch, tmp_path):
    """the shipped drop-in src/hashing.py (VERDICT r1 missing #5 / SURVEY section 7 layout): same names as the reference module,
    bound to the engine; the constructor contract of reference hashing.py:53-81 holds through it"""
    shim = _load_shim(monkeypatch, tmp_path)
    assert shim.LABEL_LOOKUP is ssa.LABEL_LOOKUP and shim.MinhashPropagation is ssa.MinhashPropagation
    assert shim.HllPropagation is ssa.HllPropagation and issubclass(shim.ElphHashes, ssa.ElphHashes)
    eh = shim.ElphHashes(_args(h=3))
    assert eh.max_hops == 3 and eh.m == 256 and eh.max_rank == 56 and eh.num_perm == 128 and eh.label_lookup is ssa.LABEL_LOOKUP[3]
    with pytest.raises(AssertionError):
        shim.ElphHashes(_args(h=4))
    lazy = _load_shim(monkeypatch, tmp_path, {'SS_LAZY_FEATURES': '1', 'SS_LAZY_MIN_LINKS': '10'})
    assert lazy._LAZY and lazy._LAZY_MIN == 10


def test_reference_modules_import_the_shim(ssa, monkeypatch, tmp_path):
    """in the build container only (needs /root/reference): the reference's OWN src/models/elph.py and src/datasets/elph.py are
    imported with `src.hashing` resolved to the shipped shim (PyG / ogb / torch_sparse stood in for by empty modules, exactly
    as tests/golden/make_golden.py does); ELPH(args) and BUDDY construct, and their `elph_hashes` is the MI355X engine"""
    import os
    import sys
    import types
    if not os.path.exists('/root/reference/src/models/elph.py'):
        pytest.skip('the reference checkout is not present on this machine (GPU box): nothing to import')
    shim = _load_shim(monkeypatch, tmp_path)

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return type(name, (torch.nn.Module,), {'__init__': lambda self, *a, **k: torch.nn.Module.__init__(self)})
    import torch
    for name in ('torch_geometric', 'torch_geometric.nn', 'torch_geometric.nn.conv', 'torch_geometric.nn.conv.gcn_conv',
                 'torch_geometric.utils', 'torch_geometric.loader', 'torch_geometric.data', 'torch_geometric.nn.dense',
                 'torch_geometric.nn.dense.linear', 'torch_geometric.nn.models', 'torch_geometric.nn.inits', 'torch_geometric.typing',
                 'torch_geometric.transforms', 'torch_geometric.datasets', 'torch_sparse', 'torch_scatter', 'ogb',
                 'ogb.linkproppred', 'wandb', 'fast_pagerank'):
        monkeypatch.setitem(sys.modules, name, _Any(name))
    pkg = types.ModuleType('src')
    pkg.__path__ = ['/root/reference/src']
    monkeypatch.setitem(sys.modules, 'src', pkg)
    monkeypatch.setitem(sys.modules, 'src.hashing', shim)            # what copying the shim over src/hashing.py does
    for name in [m for m in sys.modules if m.startswith('src.') and m != 'src.hashing']:
        monkeypatch.delitem(sys.modules, name)
    import importlib
    ref_models = importlib.import_module('src.models.elph')
    from argparse import Namespace
    a = Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True, use_feature=True,
                  feature_prop='gcn', propagate_embeddings=False, sign_k=0, label_dropout=0.0, feature_dropout=0.0,
                  hidden_channels=16, sign_dropout=0.5, add_normed_features=False, use_RA=False, use_struct_feature=True,
                  num_negs=1)
    model = ref_models.ELPH(a, num_features=8)
    assert isinstance(model.elph_hashes, ssa.ElphHashes) and model.elph_hashes.__class__ is shim.ElphHashes
    assert model.init_hashes is None and callable(model.elph_hashes.hll_prop) and callable(model.elph_hashes.minhash_prop)
    ref_data = importlib.import_module('src.datasets.elph')
    assert ref_data.ElphHashes is shim.ElphHashes                       # HashDataset.__init__ (datasets/elph.py:31) will build ours


def test_every_python_file_compiles():
    """tools/ and probes are not imported by any other test: at least their syntax is checked on this interpreter"""
    import glob
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [f for pat in ('*.py', 'tools/*.py', 'tests/*.py', 'oracle/*.py', 'integration/src/*.py', 'subgraph-sketching_amd/*.py')
             for f in glob.glob(os.path.join(root, pat))]
    assert len(files) > 40
    for f in files:
        py_compile.compile(f, doraise=True)


def test_bias_table_sensitivity_bound(ssa):
    """DESIGN 4 states how far the third-party hole can move a result: when the HLL++ bias table moves by at most s in every entry,
    a bias-branch cardinality moves by at most s (it subtracts the MEAN of six table entries) and a feature entry by about as much
    (measured amplification 1.0 - 1.14x with re-simulated tables, profiles/round5_table_sensitivity.txt; asserted <= 2x here).
    Recomputed on the CPU oracle with a table perturbed by seeded noise of amplitude s = 1.9 -- the largest difference seen between
    two independent simulations of the table at the shipped trial count -- on the G8 shape at h = 3, a third of whose
    cardinalities are on the bias branch.  Linear-counting rows must not move at all."""
    import table_sensitivity as ts  # (tests/ is on the path: conftest lives there)
    base = ssa.hll_tables.load(8, prefer='regenerated')
    s = 1.9
    rng = np.random.RandomState(5)
    moved = base._replace(bias=base.bias + rng.uniform(-s, s, size=base.bias.shape), provenance='perturbed')
    n, h = 3000, 3
    ei = ts.uniform_graph(n, 12000, 8)
    links = np.random.RandomState(9).randint(0, n, size=(4000, 2)).astype(np.int64)
    cards0, feats0 = ts.run_oracle(ei, n, h, links, base)
    cards1, feats1 = ts.run_oracle(ei, n, h, links, moved)
    lc_values = ssa.hashing.linear_counting_table(256).numpy()       # M ln(M / V): a cardinality on the linear-counting branch IS one of these
    on_lc = np.isin(cards0, lc_values) & (cards0 <= base.threshold)
    assert on_lc.any() and (~on_lc).mean() > 0.25
    assert np.array_equal(cards0[on_lc], cards1[on_lc])              # the linear-counting branch reads no table
    d_cards, d_feat = np.abs(cards1 - cards0).max(), np.abs(feats1 - feats0).max()
    assert 0 < d_cards <= s * (1 + 1e-5), d_cards                      # the mean of six entries moves by at most what the entries do
    assert d_cards / cards0[~on_lc].min() < 0.01                       # < 1 % of any bias-branch cardinality
    assert 0 < d_feat <= 2 * s, d_feat                                 # a feature: differences of Jaccard-weighted union estimates and cards


def test_peer_shard_slab_layout_and_pool_release(ssa):
    """dist.PeerShard keeps ONE allocation per shard: _SlabLayout carves the 2 h + 1 tables out of it (256-byte aligned, the shapes
    and dtypes the kernels expect, no overlap), and a dropped shard's slab goes back to the per-shape pool only if no tensor handed
    out by its builds is still alive -- otherwise it is set aside for good (a later shard must not overwrite a table somebody kept)"""
    D = ssa.dist
    L = D._SlabLayout(1001, 3, 128, 256)
    slab = torch.zeros(L.bytes, dtype=torch.uint8)
    mh, hll, cards = L.views(slab)
    assert [tuple(t.shape) for t in mh] == [(1001, 128)] * 3 and all(t.dtype == torch.int32 and t.is_contiguous() for t in mh)
    assert [tuple(t.shape) for t in hll] == [(1001, 256)] * 3 and all(t.dtype == torch.uint8 for t in hll)
    assert tuple(cards.shape) == (1001, 3) and cards.dtype == torch.float32
    ptrs = sorted((t.data_ptr() - slab.data_ptr(), t.numel() * t.element_size()) for t in mh + hll + [cards])
    assert all(off % 256 == 0 for off, _ in ptrs) and all(a + n <= b for (a, n), (b, _) in zip(ptrs, ptrs[1:])) and ptrs[-1][0] + ptrs[-1][1] <= L.bytes
    for k, t in enumerate(mh + hll):
        t.fill_(k + 1)
    cards.fill_(0.5)
    mh2, hll2, cards2 = L.views(slab)  # (what a peer carves out of the mapped slab)
    assert all(int(t[7, 5]) == k + 1 for k, t in enumerate(mh2 + hll2)) and float(cards2[1000, 2]) == 0.5
    # above PEER_SLAB_MAX_BYTES the tables are packed into several slabs in table order (a 6.8 GB slab -- citation2 size -- could not be
    # opened through IPC): same views, every slab within the limit unless a single table exceeds it
    assert len(L.slab_bytes) == 1 and L.place[0] == (0, 0)
    M = D._SlabLayout(1001, 3, 128, 256, max_slab_bytes=700000)
    assert len(M.slab_bytes) == 5 and M.bytes == L.bytes and all(b <= 700000 for b in M.slab_bytes)
    parts = [torch.zeros(b, dtype=torch.uint8) for b in M.slab_bytes]
    mh3, hll3, cards3 = M.views(parts)
    assert [tuple(t.shape) for t in mh3 + hll3] == [(1001, 128)] * 3 + [(1001, 256)] * 3 and tuple(cards3.shape) == (1001, 3)
    for k, t in enumerate(mh3 + hll3 + [cards3]):
        t.fill_(k + 1)
    assert all(bool((t == k + 1).all()) for k, t in enumerate(mh3 + hll3 + [cards3]))  # no table overlaps another
    assert len(D._SlabLayout(2927976, 3, 128, 256).slab_bytes) == 5 and max(D._SlabLayout(2927976, 3, 128, 256).slab_bytes) <= 2 << 30
    with pytest.raises(ValueError):
        M.views(parts[:2])
    del mh3, hll3, cards3, parts
    # release: pooled when nothing else refers to the slab's storage, set aside when a view survives the shard
    key = ('test', L.bytes, 'cpu')
    D._POOL.pop(key, None)
    before = len(D._QUARANTINE)
    del mh, hll, cards, mh2, hll2, cards2, t

    class Shard(object):  # what PeerShard does with its slab: own views as attributes, a finaliser that knows how many users that makes
        def __init__(self, slab_, ident):
            import weakref
            self.slab = slab_
            self.mh, self.hll, self.cards = L.views(slab_)
            weakref.finalize(self, D._release_entry, key, {'id': ident, 'slab': slab_, 'peers': {}}, D._storage_users(slab_))

    shard = Shard(slab, 1)
    table = [shard.mh[0], shard.cards]  # what a build hands out
    del table, shard                    # (the finaliser runs while the shard's own views are still alive: they must not count)
    assert len(D._POOL[key]) == 1 and len(D._QUARANTINE) == before
    slab_b = torch.zeros(L.bytes, dtype=torch.uint8)
    shard = Shard(slab_b, 2)
    kept = shard.cards[:10]  # the caller kept a piece of `cards` of a build through the shard it dropped
    del shard
    assert len(D._POOL[key]) == 1 and len(D._QUARANTINE) == before + 1
    stats = D.pool_stats()  # what a long-lived process watches: the pool never hands memory back to the allocator
    assert stats['pooled_slabs'] >= 1 and stats['pooled_bytes'] >= L.bytes and stats['set_aside_slabs'] >= 1 and stats['set_aside_bytes'] >= L.bytes
    del kept
    D._POOL.pop(key, None)
    D._QUARANTINE.pop()


def test_import_of_external_fixtures_dry_run(ssa, tmp_path):
    """tools/import_external_fixtures.sh + the exporters are the only way the third-party holes (datasketch's tables,
    real PyG) ever close -- so the path is exercised: the committed REGENERATED tables are written in the exporter's own format
    (tools/export_datasketch_fixture.write_tables), the import script validates them in --dry-run (nothing is installed), regenerates
    the golden vectors from the reference with those tables (where /root/reference exists) and must report that NO array moved --
    same numbers in, same vectors out; a truncated file must be refused"""
    import importlib.util
    import subprocess
    from conftest import REPO
    spec = importlib.util.spec_from_file_location('export_datasketch_fixture', os.path.join(REPO, 'tools', 'export_datasketch_fixture.py'))
    exporter = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(exporter)
    tables = {p: ssa.hll_tables.load(p, prefer='regenerated') for p in range(4, 17)}
    path = tmp_path / 'hllpp_tables_datasketch.npz'
    exporter.write_tables(str(path), tables)
    back = ssa.hll_tables._from_export(8, str(path))  # format round trip: what the engine would load from the installed file
    assert back.provenance == 'datasketch-export' and ssa.hll_tables.same_tables(ssa.hll_tables.table_id(back), ssa.hll_tables.table_id(tables[8]))
    installed = [os.path.join(REPO, 'subgraph-sketching_amd', 'data', 'hllpp_tables_datasketch.npz'), os.path.join(REPO, 'tests', 'golden', 'g13_pyg_sign.npz')]
    before = [os.path.exists(f) for f in installed]
    done = subprocess.run(['bash', os.path.join(REPO, 'tools', 'import_external_fixtures.sh'), '--dry-run', str(tmp_path)], capture_output=True, text=True,
                          timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    assert [os.path.exists(f) for f in installed] == before, 'a dry run must not install anything'
    assert 'dry run: would install' in done.stdout and 'datasketch tables for p in [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]' in done.stdout
    if os.path.isdir('/root/reference'):
        assert 'arrays that moved: 0' in done.stdout, done.stdout[-3000:]
    else:
        assert 'cannot be regenerated on this machine' in done.stdout
    # a file that is not the exporter's is refused before anything is touched
    np.savez(tmp_path / 'hllpp_tables_datasketch.npz', p_list=np.asarray([8]), raw_p8=np.zeros(3))
    bad = subprocess.run(['bash', os.path.join(REPO, 'tools', 'import_external_fixtures.sh'), '--dry-run', str(tmp_path)], capture_output=True, text=True,
                         timeout=120)
    assert bad.returncode != 0 and 'missing' in bad.stderr


def test_linear_counting_table_has_an_independent_restatement(ssa):
    """VERDICT r5 weak #1b: the oracle takes its linear-counting values from the table the PRODUCT computes (conftest.oracle_params), so
    "LC rows bit-exact vs the oracle" holds by construction.  The table itself is pinned here without the product's code: m * ln(m / V)
    (reference hashing.py:194-195: an int64 zero count, so m / V and the logarithm are float32) restated with numpy in float32 -- the
    quotient rounded to float32 FIRST, as torch does: near V = m that rounding moves the result by thousands of ulps of the float64
    value -- agrees with the shipped torch evaluation to 2 ulp for every V (the golden vectors pin the same values against the
    reference's own outputs, LC_RTOL = 3e-7)"""
    for p in (4, 8, 12):
        m = 1 << p
        got = ssa.hashing.linear_counting_table(m).numpy()
        ratio = np.float32(m) / np.arange(1, m + 1).astype(np.float32)
        want = np.float32(m) * np.log(ratio, dtype=np.float32)
        assert want.dtype == np.float32 and got.shape == (m + 1,) and np.isinf(got[0])
        ulp = np.spacing(np.maximum(np.abs(want), np.float32(1e-30)))
        assert np.all(np.abs(got[1:] - want) <= 2 * ulp), p
        assert got[m] == 0.0 and np.all(np.diff(got[1:]) < 0)  # V = m: empty sketch -> 0; strictly decreasing in V
