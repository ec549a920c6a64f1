"""Pins the CPU oracle (oracle/sketch_oracle.c) against golden vectors produced by importing the reference
itself (tests/golden/make_golden.py).  Runs without a GPU.

Contract: everything integer is bit-exact; floats are bit-exact on this image (the oracle follows the
reference's fp32 operation order, including torch.sum's accumulation order) and are otherwise required to be
within RTOL/ATOL -- torch's own low-order bits depend on the host's vector ISA."""
import numpy as np
import pytest

from conftest import load_golden, oracle_params
from oracle import oracle

RTOL, ATOL = 1e-5, 1e-4  # stated fp32 tolerance for cardinalities / features (values up to ~1e3)


def _prm(tables, p):
    return oracle_params(tables[p])


def test_permutation_parameters():
    g = load_golden('g1_g2_init.npz')
    for P in (8, 128):
        a, b = oracle.init_permutations(P)
        assert np.array_equal(a, g[f'perm_a_P{P}']) and np.array_equal(b, g[f'perm_b_P{P}'])
    # survey known answers
    a, b = oracle.init_permutations(128)
    assert [int(x) for x in a[:3]] == [775169054918279404, 2109959069025162, 401325382989534145]
    assert [int(x) for x in b[:3]] == [1758426461858698312, 965365488286768773, 1703346441743126657]


def test_hop0_sketches_bit_exact():
    g = load_golden('g1_g2_init.npz')
    assert np.array_equal(oracle.minhash_init(64, 128), g['init_mh_P128'])
    assert np.array_equal(oracle.minhash_init(64, 8), g['init_mh_P8'])
    assert np.array_equal(oracle.minhash_init(16, 128, first_node=100000 - 16), g['init_mh_P128_tail'])
    for p in (4, 8, 16):
        assert np.array_equal(oracle.hll_init(64, p), g[f'init_hll_p{p}'])
    tail = oracle.hll_init(64, 8, first_node=100000 - 64)
    assert np.array_equal(np.argmax(tail, axis=1), g['init_hll_p8_tail_idx'])
    assert np.array_equal(np.max(tail, axis=1), g['init_hll_p8_tail_val'])
    assert np.all(np.count_nonzero(tail, axis=1) == 1)


def test_hash_matches_pandas():
    pd_util = pytest.importorskip('pandas.util')
    n = 5000
    assert np.array_equal(oracle.hash_nodes(n), pd_util.hash_array(np.arange(1, n + 1)))


def test_bit_length_formula_agrees_with_reference_float_formula():
    """reference _np_bit_length = ceil(log2(bits + 1)) in float64 (hashing.py:89); the oracle / HIP kernels use
    an integer clz.  They agree on every node id of the largest BASELINE config (citation2: 2.93 M nodes)."""
    n = 2_927_963
    hv = oracle.hash_nodes(n)
    for p in (4, 8, 16):
        bits = hv >> np.uint64(p)
        float_formula = np.ceil(np.log2(bits + 1)).astype(int)
        rank = (64 - p) - float_formula + 1
        idx = (hv & np.uint64((1 << p) - 1)).astype(np.int64)
        sample = slice(n - 4096, n)
        regs = oracle.hll_init(4096, p, first_node=n - 4096)
        assert np.array_equal(regs[np.arange(4096), idx[sample]], rank[sample].astype(np.uint8))
        assert rank.min() >= 1 and rank.max() <= 64 - p + 1


@pytest.mark.parametrize('h', [1, 2, 3])
def test_ba40_tables_cards_features(regenerated_tables, h):
    g = load_golden('g3_g4_ba40.npz')
    prm = _prm(regenerated_tables, 8)
    tb, cards = oracle.build_hash_tables(int(g['num_nodes']), g['edge_index'], 3, 128, prm)
    for k in range(4):
        assert np.array_equal(tb[k]['minhash'], g[f't_mh_{k}']), f'minhash hop {k}'
        assert np.array_equal(tb[k]['hll'], g[f't_hll_{k}']), f'hll hop {k}'
    assert np.array_equal(cards, g['cards'])
    for zo in (0, 1):
        for fl in (0, 1):
            f, dbg = oracle.pair_features(g['links'], tb, cards[:, :h].copy(), h, prm, bool(zo), bool(fl), debug=True)
            assert np.array_equal(dbg['match'], g[f'match_h{h}'])
            assert np.array_equal(dbg['zeros'], g[f'zeros_h{h}'])
            assert np.array_equal(dbg['inter'].reshape(len(f), -1), g[f'inter_h{h}'])
            ref = g[f'feat_h{h}_zo{zo}_fl{fl}']
            assert not g[f'feat_h{h}_zo{zo}_fl{fl}_uses_tables'].any()  # whole graph on the linear-counting branch
            np.testing.assert_allclose(f, ref, rtol=RTOL, atol=ATOL)
            assert np.array_equal(f, ref), 'not bit-exact against the reference on the golden host'
    f = oracle.pair_features(g['links'], tb, cards[:, :2].copy(), 2, prm)
    assert np.array_equal(f, g['feat_h2_batched7'])
    assert np.array_equal(f[:1], g['feat_h2_1d'])


@pytest.mark.parametrize('p,P', [(4, 8), (16, 128), (6, 64)])
def test_other_parameterisations(regenerated_tables, p, P):
    g = load_golden('g3b_params.npz')
    prm = _prm(regenerated_tables, p)
    tb, cards = oracle.build_hash_tables(int(g['num_nodes']), g['edge_index'], 2, P, prm)
    key = f'p{p}P{P}'
    for k in range(3):
        assert np.array_equal(tb[k]['minhash'], g[f'{key}_mh_{k}'])
        assert np.array_equal(tb[k]['hll'], g[f'{key}_hll_{k}'])
    pinned = ~g[f'{key}_cards_uses_tables']
    assert np.array_equal(cards[pinned], g[f'{key}_cards'][pinned])          # table-independent: bit-exact
    np.testing.assert_allclose(cards, g[f'{key}_cards'], rtol=RTOL, atol=ATOL)  # same tables in: within fp32 tol
    f = oracle.pair_features(g['links'], tb, cards, 2, prm)
    np.testing.assert_allclose(f, g[f'{key}_feat'], rtol=RTOL, atol=ATOL)


def test_hll_count_known_answers(regenerated_tables):
    g = load_golden('g5_hll_count.npz')
    prm = _prm(regenerated_tables, 8)
    out, br = oracle.hll_count(g['regs'], prm, return_branch=True)
    lc = br == 0
    assert np.array_equal(out[lc], g['count'][lc]), 'linear-counting branch must be bit-exact'
    # raw-estimate branches: the reference sums 2^-reg in fp32 in torch's SIMD order, the oracle sums exactly
    np.testing.assert_allclose(out, g['count'], rtol=RTOL, atol=ATOL)
    assert np.max(np.abs(out - g['count']) / np.maximum(g['count'], 1)) < 5e-7
    assert np.array_equal(br == 1, g['count_uses_tables']), 'branch selection must match the reference'
    assert set(np.unique(br)) == {0, 1, 2}, 'fixture must reach all three estimator branches'
    # survey known answers: independent of the bias tables
    assert out[0] == 0.0
    assert out[1] == np.float32(1.0019733905792236)
    assert out[2] == np.float32(216.24244689941406) and out[3] == np.float32(218.58035278320312)
    assert br[4] == 1  # V = 108: lc = 220.94 > 220 -> estimator branch
    assert np.array_equal(oracle.hll_count(g['regs'][1], prm), g['count_1d'])          # 1-D input -> shape [1]
    np.testing.assert_allclose(oracle.hll_count(g['regs'].astype(np.int64), prm), g['count_int64'], rtol=RTOL, atol=ATOL)
    assert np.array_equal(oracle.hll_count(g['regs'].astype(np.int64), prm), out)      # int64 registers accepted
    # logf fallback (no host lc table) stays within tolerance
    np.testing.assert_allclose(oracle.hll_count(g['regs'], oracle_params(regenerated_tables[8], with_lc=False)),
                               g['count'], rtol=RTOL, atol=ATOL)


def test_edge_cases_isolated_nodes_self_loops(regenerated_tables):
    """trailing isolated nodes get no self loop -> all-zero hop>=1 rows (hashing.py:148); one-way edges"""
    g = load_golden('g7_edge_cases.npz')
    prm = _prm(regenerated_tables, 8)
    n = int(g['num_nodes'])
    tb, cards = oracle.build_hash_tables(n, g['edge_index'], 2, 128, prm)
    for k in range(3):
        assert np.array_equal(tb[k]['minhash'], g[f't_mh_{k}'])
        assert np.array_equal(tb[k]['hll'], g[f't_hll_{k}'])
    assert not tb[1]['minhash'][9:].any() and not tb[2]['hll'][9:].any()
    assert np.array_equal(cards, g['cards'])
    assert np.array_equal(oracle.pair_features(g['links'], tb, cards, 2, prm), g['feat'])
    # CSR pull with implicit self loops == edge scatter with explicit ones
    rowptr, col = oracle.csr_build(n, g['edge_index'])
    n_self = int(g['edge_index'].max()) + 1
    mh, hll, c = oracle.propagate_csr(n, rowptr, col, n_self, tb[0]['minhash'], tb[0]['hll'], prm)
    assert np.array_equal(mh, tb[1]['minhash']) and np.array_equal(hll, tb[1]['hll']) and np.array_equal(c, cards[:, 0])


@pytest.mark.parametrize('h', [2, 3])
def test_uniform3000_reaches_bias_branch(regenerated_tables, h):
    import hashlib
    g = load_golden('g8_uniform3000.npz')
    n, e_und, seed = [int(x) for x in g['graph']]
    rng = np.random.RandomState(seed)
    e = rng.randint(0, n, size=(2, e_und)).astype(np.int64)
    ei = np.concatenate([e, e[::-1]], axis=1)
    prm = _prm(regenerated_tables, 8)
    tb, cards = oracle.build_hash_tables(n, ei, 3, 128, prm)
    for k in range(4):
        assert hashlib.sha256(tb[k]['hll'].tobytes()).hexdigest() == str(g[f'sha_hll_{k}'])
        assert hashlib.sha256(tb[k]['minhash'].tobytes()).hexdigest() == str(g[f'sha_mh_{k}'])
    pinned = ~g['cards_uses_tables']
    assert 0.05 < g['cards_uses_tables'].mean() < 0.95
    assert np.array_equal(cards[pinned], g['cards'][pinned])
    np.testing.assert_allclose(cards, g['cards'], rtol=RTOL, atol=ATOL)
    f, dbg = oracle.pair_features(g['links'], tb, cards[:, :h].copy(), h, prm, debug=True)
    assert np.array_equal(dbg['match'], g[f'match_h{h}']) and np.array_equal(dbg['zeros'], g[f'zeros_h{h}'])
    ref = g[f'feat_h{h}']
    # features are differences of numbers up to ~3000: absolute tolerance scaled to the operands
    np.testing.assert_allclose(f, ref, rtol=RTOL, atol=1e-5 * float(np.abs(g['cards']).max()) * 4)


def test_reference_style_torch_path_matches_oracle(regenerated_tables):
    """the second CPU baseline of bench.py (stock torch ops, reference dataflow) computes the same things as the oracle"""
    import torch
    from oracle import torch_refstyle as tr
    t = regenerated_tables[8]
    prm = _prm(regenerated_tables, 8)
    n = 3000
    rng = np.random.RandomState(1)
    e = rng.randint(0, n, size=(2, 12000)).astype(np.int64)
    ei = np.concatenate([e, e[::-1]], axis=1)
    otab, ocards = oracle.build_hash_tables(n, ei, 2, 128, prm)
    raw, bias = torch.tensor(t.raw_estimate, dtype=torch.float), torch.tensor(t.bias, dtype=torch.float)
    tabs, cards = tr.build_tables(n, torch.from_numpy(ei), 2, torch.from_numpy(otab[0]['minhash'].astype(np.int64)),
                                  torch.from_numpy(otab[0]['hll'].view(np.int8)), 8, t.alpha, t.threshold, raw, bias)
    for k in (1, 2):
        assert np.array_equal(tabs[k]['minhash'].numpy().astype(np.uint32), otab[k]['minhash'])
        assert np.array_equal(tabs[k]['hll'].numpy().view(np.uint8), otab[k]['hll'])
    np.testing.assert_allclose(cards.numpy(), ocards, rtol=RTOL, atol=ATOL)
    links = rng.randint(0, n, size=(500, 2)).astype(np.int64)
    _, dbg = oracle.pair_features(links, otab, ocards, 2, prm, debug=True)
    inter = tr.pair_intersections(torch.from_numpy(links), tabs, 2, 128, 8, t.alpha, t.threshold, raw, bias)
    for k1 in (1, 2):
        for k2 in (1, 2):
            np.testing.assert_allclose(inter[(k1, k2)].numpy(), dbg['inter'][:, k1 - 1, k2 - 1], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize('h', [1, 2, 3])
def test_degree_normalised_features_vs_reference(h):
    """BUDDY._append_degree_normalised (models/elph.py:276-293) as computed by the reference's own method"""
    g = load_golden('g9_degree_normalised.npz')
    base = load_golden('g3_g4_ba40.npz')[f'feat_h{h}_zo1_fl0']
    got = oracle.append_degree_normalised(base, g['links'], g['degrees'])
    assert got.shape == (len(base), 2 * base.shape[1])
    assert np.array_equal(got, g[f'normed_h{h}']), 'sqrt / divide / NaN-Inf rule must be bit-exact'
    zero_rows = (g['degrees'][g['links'][:, 0]] == 0) | (g['degrees'][g['links'][:, 1]] == 0)
    assert zero_rows.any() and not got[zero_rows][:, base.shape[1]:].any()


def _g10_matrices():
    import scipy.sparse as ssp
    g = load_golden('g10_heuristics.npz')
    n = int(g['num_nodes'])
    coo = (g['src'], g['dst'])
    return g, {'': ssp.csr_matrix((g['w'], coo), shape=(n, n)),
               '_float_weights': ssp.csr_matrix((g['w'].astype(np.float64) * 0.37, coo), shape=(n, n)),
               '_unit_weights': ssp.csr_matrix((np.ones(g['src'].size, dtype=int), coo), shape=(n, n))}


def test_common_neighbour_heuristics_vs_reference():
    """CN / AA / RA of the reference's src/heuristics.py (G10): the oracle sums in fp64 in column order like scipy, so the
    float32 results are bit-identical"""
    g, mats = _g10_matrices()
    for kind in ('CN', 'AA', 'RA'):
        got = oracle.common_neighbour_scores(mats[''], g['links'], kind)
        assert got.dtype == np.float32 and np.array_equal(got, g[kind]), kind
    for suffix in ('_float_weights', '_unit_weights'):
        assert np.array_equal(oracle.common_neighbour_scores(mats[suffix], g['links'], 'RA'), g['RA' + suffix]), suffix


def test_sign_feature_propagation_restatement():
    """gcn_norm + spmm (reference datasets/elph.py:87-110) have no golden vectors: PyG and torch_sparse are absent from this
    image, so this row is PARITY UNPINNED.  What can be checked: the restated semantics against an independent dense fp64
    evaluation of D^-1/2 (A + remaining self loops) D^-1/2 x."""
    rng = np.random.RandomState(3)
    n = 120
    e = rng.randint(0, n, size=(2, 900))
    e[:, :7] = np.array([[5, 5, 9, 9, 11, 40, 40], [5, 5, 9, 9, 11, 41, 41]])  # self loops (one duplicated) and a duplicate edge
    w = rng.randint(1, 5, size=900).astype(np.float32)
    ei, wn = oracle.gcn_norm(e, w, n)
    x = rng.randn(n, 12).astype(np.float32)
    got = oracle.spmm(ei, wn, n, x)
    A = np.zeros((n, n))
    keep = e[0] != e[1]
    np.add.at(A, (e[0][keep], e[1][keep]), w[keep].astype(np.float64))
    loop = np.ones(n)
    loop[e[0][~keep]] = w[~keep]  # last listed weight of an existing self loop
    A[np.arange(n), np.arange(n)] += loop
    deg = A.sum(axis=0)
    dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    want = (dinv[:, None] * A * dinv[None, :]) @ x.astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)


def test_sign_features_vs_the_references_own_function():
    """G15 = outputs of the reference's OWN HashDataset._generate_sign_features (datasets/elph.py:87-110; made by
    tests/golden/make_golden.py --only-g15) for sign_k in {0, 2}, unit and non-integer weights, F = 12 / 64, on a graph with
    existing self loops, duplicate edges, isolated nodes and a hub -- under the RESTATED gcn_norm / torch_sparse.spmm (PyG and
    torch_sparse are not in this image: "PyG semantics restated", as for A6).  Pins the reference-owned part -- the sign_k loop
    that re-multiplies data.x, the concatenation -- and the edge-order accumulation of both sums: bit-exact."""
    g = load_golden('g15_sign_features.npz')
    n = int(g['num_nodes'])
    for F in (12, 64):
        for wname in ('unit', 'float'):
            for k in (0, 2):
                got = oracle.generate_sign_features(g[f'x_F{F}'], g['edge_index'], g[f'w_{wname}'], k)
                want = g[f'sign_k{k}_F{F}_{wname}']
                assert got.shape == want.shape == (n, F * (1 if k == 0 else k + 1))
                assert np.array_equal(got, want), (F, wname, k, float(np.abs(got - want).max()))


def test_estimate_bias_entry_point_matches_the_hll_count_branch(regenerated_tables):
    """so_estimate_bias (the stand-alone _estimate_bias / _refine_hll_count_estimate) against golden G5's values"""
    prm = oracle_params(regenerated_tables[8])
    t = regenerated_tables[8]
    e = np.linspace(float(t.raw_estimate.min()), 1400.0, 200).astype(np.float32)
    bias = oracle.estimate_bias(e, prm)
    raw32, bias32 = t.raw_estimate.astype(np.float32), t.bias.astype(np.float32)
    for x, b in zip(e[::17], bias[::17]):  # independent numpy restatement: stable argsort of fp32 squared distances
        near = np.argsort((x - raw32) ** 2, kind='stable')[:6]
        want = np.float32(0)
        for j in near:
            want = np.float32(want + bias32[j])
        assert b == np.float32(want / np.float32(6))
    refined = oracle.estimate_bias(e, prm, refine=True)
    small = e <= np.float32(1280.0)
    assert np.array_equal(refined[small], (e - bias)[small]) and np.array_equal(refined[~small], e[~small])


def test_datasketch_tables_fixture():
    """VERDICT r1 missing #4: real HLL++ tables.  tests/golden/g11_datasketch_tables.npz and
    subgraph-sketching_amd/data/hllpp_tables_datasketch.npz are produced by tools/export_datasketch_fixture.py on a
    machine that has `datasketch`; until someone commits them this test SKIPS (loudly) and the bias-corrected branch stays
    'parity unpinned'.  With them: the shipped export hashes to the recorded digests, hll_tables.load() serves it, and the
    oracle's bias correction reproduces the float64 known answers computed from datasketch's own numbers."""
    import os
    import hashlib
    from conftest import GOLDEN
    import subgraph_sketching_amd as ssa
    path = os.path.join(GOLDEN, 'g11_datasketch_tables.npz')
    if not (os.path.exists(path) and os.path.exists(ssa.hll_tables.EXPORTED)):
        pytest.skip('PARITY UNPINNED: no datasketch export committed (run tools/export_datasketch_fixture.py where datasketch '
                    'is installed and commit its two output files)')
    g = dict(np.load(path, allow_pickle=False))
    for p in (4, 6, 8, 12, 16):
        t = ssa.hll_tables._from_export(p)
        assert t is not None and t.provenance == 'datasketch-export'
        assert hashlib.sha256(np.asarray(t.raw_estimate, dtype=np.float64).tobytes()).hexdigest() == str(g[f'sha_raw_p{p}'])
        assert hashlib.sha256(np.asarray(t.bias, dtype=np.float64).tobytes()).hexdigest() == str(g[f'sha_bias_p{p}'])
        assert t.alpha == float(g[f'alpha_p{p}']) and t.threshold == float(g[f'threshold_p{p}'])
        assert ssa.hll_tables.table_id(t) == str(g[f'table_id_p{p}'])
        prm = oracle_params(t, with_lc=False)
        got = oracle.estimate_bias(g[f'estimate_p{p}'].astype(np.float32), prm, refine=True)
        np.testing.assert_allclose(got, g[f'corrected_p{p}'], rtol=2e-6, atol=2e-4 * (1 << p) / 256)


def test_reference_elph_forward_fixture_vs_oracle(regenerated_tables):
    """golden G12 (outputs of the reference's own ELPH.forward + its training-loop query) against the oracle"""
    g = load_golden('g12_elph_forward.npz')
    prm = _prm(regenerated_tables, 8)
    for tag in ('ba', 'uni'):
        n, h = int(g[f'{tag}_num_nodes']), int(g[f'{tag}_hops'])
        tables, cards = oracle.build_hash_tables(n, g[f'{tag}_edge_index'], h, 128, prm)
        if tag == 'ba':
            for k in range(h + 1):
                assert np.array_equal(tables[k]['minhash'], g[f'ba_t_mh_{k}']) and np.array_equal(tables[k]['hll'], g[f'ba_t_hll_{k}'])
        np.testing.assert_allclose(cards, g[f'{tag}_cards'], rtol=RTOL, atol=ATOL)
        feats = oracle.pair_features(g[f'{tag}_links'], tables, cards, h, prm)
        np.testing.assert_allclose(feats, g[f'{tag}_feat'], rtol=RTOL, atol=ATOL * 10)


def _pyg_fixture():
    import os
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, 'g13_pyg_sign.npz')
    if not os.path.exists(path):
        pytest.skip('PARITY UNPINNED: no PyG / torch_sparse fixture committed (run tools/export_pyg_fixture.py where those packages '
                    'are installed and commit tests/golden/g13_pyg_sign.npz)')
    return dict(np.load(path, allow_pickle=False))


def test_pyg_sign_fixture():
    """VERDICT r1 missing #3: gcn_norm / spmm against outputs of the REAL torch_geometric / torch_sparse (fixture made by
    tools/export_pyg_fixture.py).  Edge order of gcn_norm's output is part of the contract (spmm accumulates in that order)."""
    g = _pyg_fixture()
    n = int(g['num_nodes'])
    for tag in ('int', 'frac'):
        ei, w = oracle.gcn_norm(g['edge_index'], g[f'w_{tag}'], n)
        assert np.array_equal(ei, g[f'norm_edge_index_{tag}']), 'gcn_norm: edge order / self-loop placement differs from PyG'
        np.testing.assert_allclose(w, g[f'norm_weight_{tag}'], rtol=2e-6, atol=0)
        np.testing.assert_allclose(oracle.spmm(ei, w, n, g['x']), g[f'spmm_{tag}'], rtol=1e-5, atol=1e-6)


def _buddy_precompute_with(compute, g, tag, fl, zo):
    """HashDataset.__init__'s sketch side re-enacted (reference datasets/elph.py:175-222): build, query every link, then the
    POST-HOC floor and knock-out on the returned tensor -- the ElphHashes the dataset owns is constructed from the SAME args, so
    the query has already applied both; the second application must be idempotent"""
    h = int(g[f'{tag}_hops'])
    sf = compute(int(g[f'{tag}_num_nodes']), g[f'{tag}_edge_index'], g[f'{tag}_links'], h, bool(fl), bool(zo))
    sf = np.array(sf, copy=True)
    if fl:
        sf[sf < 0] = 0
    if not zo:
        if h > 1:
            sf[:, [4, 5]] = 0
        if h == 3:
            sf[:, [11, 12]] = 0
    return sf


def test_buddy_precompute_vs_the_references_hash_dataset(regenerated_tables):
    """G14 = outputs of the reference's OWN HashDataset (BUDDY's feature precompute, the call BASELINE's north_star names; made
    by tests/golden/make_golden.py --only-g14 importing src/datasets/elph.py): subgraph_features for floor_sf x use_zero_one,
    degrees, RA, cache file names.  The oracle through the same sequence of calls."""
    import scipy.sparse as ssp
    g = load_golden('g14_hash_dataset.npz')
    prm = oracle_params(regenerated_tables[8])

    def compute(n, ei, links, h, fl, zo):
        tables, cards = oracle.build_hash_tables(n, ei, h, 128, prm)
        return oracle.pair_features(links, tables, cards, h, prm, use_zero_one=zo, floor_sf=fl)
    for tag in ('ba', 'uni'):
        n, ei, h = int(g[f'{tag}_num_nodes']), g[f'{tag}_edge_index'], int(g[f'{tag}_hops'])
        links = g[f'{tag}_links']
        assert np.array_equal(links, np.concatenate([g[f'{tag}_pos'], g[f'{tag}_neg']]))      # datasets/elph.py:51
        assert list(g[f'{tag}_labels']) == [1] * len(g[f'{tag}_pos']) + [0] * len(g[f'{tag}_neg'])
        A = ssp.csr_matrix((np.ones(ei.shape[1], dtype=int), (ei[0], ei[1])), shape=(n, n))    # :62,68-71
        assert np.array_equal(np.asarray(A.sum(axis=0, dtype=float), dtype=np.float32).ravel(), g[f'{tag}_degrees'])   # :74
        assert np.array_equal(oracle.common_neighbour_scores(A, links, 'RA'), g[f'{tag}_RA'])  # :76-77
        hop = '' if h == 2 else f'{h}hop_'
        for fl in (0, 1):
            for zo in (0, 1):
                key = f'{tag}_fl{fl}_zo{zo}'
                assert list(g[f'{key}_files']) == [f'elph_train_{hop}cardcache.pt', f'elph_train_{hop}hashcache.pt',
                                                   f'elph_train_{hop}subgraph_featurecache.pt']               # :154-173,187-188
                want, uses = g[f'{key}_subgraph_features'], g[f'{key}_uses_tables']
                got = _buddy_precompute_with(compute, g, tag, fl, zo)
                assert got.shape == want.shape == (len(links), h * (h + 2)) and got.dtype == np.float32
                assert np.array_equal(got[~uses], want[~uses]), f'{key}: entries independent of the bias tables must be bit-exact'
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4, err_msg=key)
                # the reference saves its cache BEFORE the post-hoc edits (:211-213 then :214-222): identical here because the
                # query already floored / knocked out (the dataset's ElphHashes is built from the same args)
                assert np.array_equal(g[f'{key}_cached_features'], want)
                if fl:
                    assert (want >= 0).all()
                if not zo:
                    assert not want[:, [4, 5]].any()
        assert g[f'{tag}_fl0_zo1_uses_tables'].mean() < 0.9   # a good part of the fixture pins values whatever the tables
