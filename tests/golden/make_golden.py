#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  Nothing here travels to the GPU
box except the produced `.npz` data files.  The reference file is imported unmodified; the
two third-party packages it needs that are absent from the image are replaced by minimal
`sys.modules` stand-ins restating their documented semantics:

  torch_geometric.nn.MessagePassing(aggr='max').propagate(edge_index, x)
      out[i] = max over edges (j -> i) of x[j]; rows without an in-edge stay 0
  torch_geometric.utils.add_self_loops(edge_index, num_nodes=None)
      appends (i, i) for i < N, N = num_nodes or max(edge_index)+1 (0 for an empty edge_index)
  torch_geometric.loader.DataLoader = torch.utils.data.DataLoader
  datasketch.HyperLogLogPlusPlus(p) -> .alpha, .max_rank, .reg, .hashfunc
  datasketch.hyperloglog_const._thresholds/_bias/_raw_estimate
      THE REAL TABLES ARE NOT AVAILABLE OFFLINE.  The stand-in serves the regenerated
      tables of subgraph-sketching_amd/data/hllpp_tables_regenerated.npz.  Every golden
      output that depended on them is flagged in a `*_uses_tables` mask obtained by
      re-running the reference with NaN bias tables; only the unflagged entries pin
      parity with the reference independent of table provenance.

Usage:  python tests/golden/make_golden.py      (writes tests/golden/*.npz)
"""
import hashlib
import importlib.util
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/src/hashing.py'
TABLES = os.path.join(REPO, 'subgraph-sketching_amd', 'data', 'hllpp_tables_regenerated.npz')
# tools/import_external_fixtures.sh: the same vectors with datasketch's REAL tables (SS_GOLDEN_TABLES = the file written by
# tools/export_datasketch_fixture.py) into another directory (SS_GOLDEN_OUT), to be compared with the committed ones
EXPORTED_TABLES = os.environ.get('SS_GOLDEN_TABLES')
OUT = os.environ.get('SS_GOLDEN_OUT', HERE)


def _alpha(p):
    m = 1 << p
    return {4: 0.673, 5: 0.697, 6: 0.709}.get(p, 0.7213 / (1.0 + 1.079 / m))


def _load_tables():
    """{'p_min', 'p_max', 'thresholds' [15], 'raw_p<p>', 'bias_p<p>', 'alpha' {p: value} or None}: the regenerated tables this
    repository ships, or datasketch's own as exported by tools/export_datasketch_fixture.py (SS_GOLDEN_TABLES)"""
    if not EXPORTED_TABLES:
        tz = np.load(TABLES)
        return {**{k: tz[k] for k in tz.files}, 'alpha': None}
    ez = np.load(EXPORTED_TABLES)
    ps = [int(p) for p in ez['p_list']]
    out = {'p_min': min(ps), 'p_max': max(ps), 'alpha': {p: float(ez[f'alpha_p{p}']) for p in ps},
           'thresholds': np.asarray([float(ez[f'threshold_p{p}']) if p in ps else 0.0 for p in range(4, 19)])}
    for p in ps:
        out[f'raw_p{p}'], out[f'bias_p{p}'] = ez[f'raw_p{p}'], ez[f'bias_p{p}']
    return out


def install_shims(nan_bias=False):
    tz = _load_tables()
    pmin, pmax = int(tz['p_min']), int(tz['p_max'])

    class MessagePassing(torch.nn.Module):
        def __init__(self, aggr='max'):
            super().__init__()
            assert aggr == 'max'

        def propagate(self, edge_index, x):
            src, dst = edge_index[0], edge_index[1]
            out = torch.zeros_like(x)
            if src.numel() == 0:
                return out
            idx = dst.unsqueeze(1).expand(-1, x.size(1))
            return out.scatter_reduce(0, idx, x[src], 'amax', include_self=False)

    def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
        if num_nodes is None:
            num_nodes = int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
        loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device).repeat(2, 1)
        return torch.cat([edge_index, loop], dim=1), None

    tg = types.ModuleType('torch_geometric')
    tg_nn = types.ModuleType('torch_geometric.nn')
    tg_nn.MessagePassing = MessagePassing
    tg_utils = types.ModuleType('torch_geometric.utils')
    tg_utils.add_self_loops = add_self_loops
    tg_loader = types.ModuleType('torch_geometric.loader')
    tg_loader.DataLoader = torch.utils.data.DataLoader
    tg.nn, tg.utils, tg.loader = tg_nn, tg_utils, tg_loader

    class HyperLogLogPlusPlus:
        def __init__(self, p=8):
            self.p = p
            self.m = 1 << p
            self.alpha = tz['alpha'][p] if tz['alpha'] else _alpha(p)
            self.max_rank = 64 - p
            self.reg = np.zeros(self.m, dtype=np.int8)
            self.hashfunc = None

    const = types.ModuleType('datasketch.hyperloglog_const')
    const._thresholds = [float(x) for x in tz['thresholds']]
    raw, bias = [], []
    for p in range(4, 19):
        if pmin <= p <= pmax:
            raw.append([float(x) for x in tz[f'raw_p{p}']])
            b = tz[f'bias_p{p}']
            bias.append([float('nan')] * len(b) if nan_bias else [float(x) for x in b])
        else:
            raw.append([0.0] * 200)
            bias.append([float('nan')] * 200)
    const._raw_estimate, const._bias = raw, bias
    ds = types.ModuleType('datasketch')
    ds.HyperLogLogPlusPlus = HyperLogLogPlusPlus
    ds.hyperloglog_const = const
    sys.modules.update({'torch_geometric': tg, 'torch_geometric.nn': tg_nn, 'torch_geometric.utils': tg_utils,
                        'torch_geometric.loader': tg_loader, 'datasketch': ds,
                        'datasketch.hyperloglog_const': const})
    return add_self_loops


def load_reference(nan_bias=False):
    asl = install_shims(nan_bias)
    spec = importlib.util.spec_from_file_location('ref_hashing_nan' if nan_bias else 'ref_hashing', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import tqdm as _tqdm  # silence progress bars
    mod.tqdm = lambda it, **kw: it
    return mod, asl


def load_reference_buddy():
    """the reference's BUDDY class (src/models/elph.py); its module imports PyG layers that are absent here and unused by
    the one static-like method we call, so they are stood in for by empty classes"""
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return type(name, (object,), {})
    for name in ('torch_geometric.nn.conv', 'torch_geometric.nn.conv.gcn_conv', 'torch_geometric.nn.dense',
                 'torch_geometric.nn.dense.linear', 'torch_geometric.nn.models', 'torch_geometric.nn.inits', 'torch_sparse',
                 'torch_scatter', 'torch_geometric.data', 'torch_geometric.typing'):
        sys.modules.setdefault(name, _Any(name))
    tgnn = sys.modules['torch_geometric.nn']
    for name in ('GCNConv', 'SAGEConv', 'global_mean_pool', 'global_sort_pool', 'global_add_pool', 'global_max_pool', 'GINConv',
                 'MLP', 'TransformerConv', 'GATConv'):
        if not hasattr(tgnn, name):
            setattr(tgnn, name, type(name, (object,), {}))
    if '/root/reference' not in sys.path:
        sys.path.insert(0, '/root/reference')
    import src.models.elph as ref_models
    return ref_models.BUDDY


def load_reference_heuristics():
    """the reference's src/heuristics.py (CN / AA / RA); needs only the DataLoader shim, scipy and tqdm"""
    install_shims()
    if '/root/reference' not in sys.path:
        sys.path.insert(0, '/root/reference')
    import src.heuristics as ref_heuristics
    return ref_heuristics


def make_g10():
    """G10: CN / AA / RA of the reference on a small weighted directed multigraph (duplicates are summed by csr_matrix,
    datasets/elph.py:68-71), with isolated nodes, a hub, zero column sums and column sums of 1 (log -> 0 -> inf -> 0)"""
    import scipy.sparse as ssp
    H = load_reference_heuristics()
    rng = np.random.RandomState(5)
    n = 90
    src = np.concatenate([rng.randint(0, 80, size=700), np.full(60, 7), rng.randint(0, 80, size=60), [81, 82]])
    dst = np.concatenate([rng.randint(0, 80, size=700), rng.randint(0, 80, size=60), np.full(60, 7), [83, 83]])
    w = rng.randint(1, 4, size=src.size).astype(np.int64)
    A = ssp.csr_matrix((w, (src, dst)), shape=(n, n))       # int64 weights, as the reference builds it
    links = np.concatenate([rng.randint(0, n, size=(300, 2)), [[7, 7], [7, 3], [3, 7], [85, 86], [81, 82], [82, 81], [0, 0]]]).astype(np.int64)
    g = {'src': src, 'dst': dst, 'w': w, 'num_nodes': np.asarray(n), 'links': links}
    lk = torch.from_numpy(links)
    # the reference indexes scipy matrices with torch tensors (A[src], heuristics.py:21); scipy >= 1.13 rejects those while
    # probing `idx.dtype.kind`.  Environment shim: hand scipy the same indices as numpy arrays.
    import scipy.sparse._index as _spi
    _orig_validate = _spi.IndexMixin._validate_indices

    def _validate_with_tensors(self, key, *a, **kw):
        conv = (lambda k: k.numpy() if isinstance(k, torch.Tensor) else k)
        key = tuple(conv(k) for k in key) if isinstance(key, tuple) else conv(key)
        return _orig_validate(self, key, *a, **kw)
    _spi.IndexMixin._validate_indices = _validate_with_tensors
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for name in ('CN', 'AA', 'RA'):
            g[name] = getattr(H, name)(A, lk, batch_size=128)[0].numpy()
        Af = ssp.csr_matrix((w.astype(np.float64) * 0.37, (src, dst)), shape=(n, n))   # float edge weights
        g['RA_float_weights'] = H.RA(Af, lk, batch_size=1000)[0].numpy()
        Au = ssp.csr_matrix((np.ones(src.size, dtype=int), (src, dst)), shape=(n, n))  # edge_weight = ones (datasets/elph.py:62)
        g['RA_unit_weights'] = H.RA(Au, lk, batch_size=1000)[0].numpy()
    assert g['RA'].dtype == np.float32 and np.isfinite(g['RA']).all() and (g['CN'] > 0).sum() > 100
    np.savez_compressed(os.path.join(OUT, 'g10_heuristics.npz'), **g)
    print('G10 written')


def load_reference_elph():
    """the reference's ELPH module class (src/models/elph.py:98-218) running on the reference's own src/hashing.py.
    PyG is absent: `GCNConv` is stood in for by a small dense-normalised graph convolution (its output `x` is NOT part of
    the fixture -- only the sketch side of ELPH.forward is), everything else the module imports but the forward pass never
    touches by empty classes (load_reference_buddy)."""
    install_shims()
    load_reference_buddy()  # installs the stand-ins for unused imports and puts /root/reference on sys.path

    class GCNConv(torch.nn.Module):
        def __init__(self, in_channels, out_channels, **kw):
            super().__init__()
            self.lin = torch.nn.Linear(in_channels, out_channels)

        def forward(self, x, edge_index):
            src, dst = edge_index
            deg = torch.zeros(x.size(0)).index_add_(0, dst, torch.ones(dst.numel())).clamp(min=1)
            out = torch.zeros(x.size(0), self.lin.out_features).index_add_(0, dst, self.lin(x)[src])
            return out / deg[:, None]
    import src.models.elph as ref_models
    ref_models.GCNConv = GCNConv
    ref_models.add_self_loops = sys.modules['torch_geometric.utils'].add_self_loops
    return ref_models.ELPH


def make_g12():
    """G12: the sketch outputs of the reference's own ELPH.forward (models/elph.py:180-218) -- node_hashings_table and cards
    -- and of the query its training loop issues right after (runners/train.py:198-204), on two graphs:
      'ba'  : the 40-node Barabasi-Albert graph of G3, max_hash_hops = 3, full tables
      'uni' : the 3000-node uniform graph of G8, max_hash_hops = 2, sha256 of the tables + cards + features of 512 links
    VERDICT r1 missing #5: the reference CALLER, not a re-expressed call sequence, produced these."""
    ELPH = load_reference_elph()
    refnan, _ = load_reference(True)
    load_reference(False)
    g = {}
    for tag, (n, ei, h) in {'ba': (40, ba_graph(40, 5, seed=7), 3), 'uni': (3000, uniform_graph(3000, 12000, seed=1), 2)}.items():
        a = Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True, use_feature=True,
                      feature_prop='gcn', propagate_embeddings=False, sign_k=0, label_dropout=0.0, feature_dropout=0.0,
                      hidden_channels=16, sign_dropout=0.5)
        torch.manual_seed(0)
        model = ELPH(a, num_features=8)
        model.eval()
        x = torch.randn(n, 8)
        edge_index = torch.from_numpy(ei)
        with torch.no_grad():
            _, table, cards = model(x, edge_index)
            _, table2, cards2 = model(x, edge_index)  # second forward: hop-0 sketches are cached on the module (:189-192)
        assert all(torch.equal(table[k]['minhash'], table2[k]['minhash']) and torch.equal(table[k]['hll'], table2[k]['hll'])
                   for k in table) and torch.equal(cards, cards2)
        links = torch.from_numpy(np.random.RandomState(12).randint(0, n, size=(512 if tag == 'uni' else 96, 2)).astype(np.int64))
        feats = model.elph_hashes.get_subgraph_features(links, table, cards)          # train.py:204
        g[f'{tag}_edge_index'], g[f'{tag}_num_nodes'], g[f'{tag}_hops'] = ei, np.asarray(n), np.asarray(h)
        g[f'{tag}_links'], g[f'{tag}_feat'], g[f'{tag}_cards'] = links.numpy(), feats.numpy(), cards.numpy()
        # which outputs depended on the (regenerated) bias tables: same forward with NaN tables
        eh_nan = refnan.ElphHashes(a)
        tn = {k: table[k] for k in table}
        cn = torch.zeros_like(cards)
        for k in range(1, h + 1):
            cn[:, k - 1] = eh_nan.hll_count(table[k]['hll'])
        g[f'{tag}_cards_uses_tables'] = torch.isnan(cn).numpy()
        g[f'{tag}_feat_uses_tables'] = torch.isnan(eh_nan.get_subgraph_features(links, tn, cn)).numpy()
        if tag == 'ba':
            table_arrays(table, 'ba_t', g)
        else:
            for k in table:
                g[f'uni_sha_hll_{k}'] = np.asarray(sha(table[k]['hll'].numpy().astype(np.uint8)))
                g[f'uni_sha_mh_{k}'] = np.asarray(sha(table[k]['minhash'].numpy().astype(np.uint32)))
        assert table[1]['minhash'].dtype == torch.int64 and table[1]['hll'].dtype == torch.int8 and cards.dtype == torch.float32
    np.savez_compressed(os.path.join(OUT, 'g12_elph_forward.npz'), **g)
    print('G12 written')


def load_reference_hash_dataset(nan_bias=False):
    """the reference's own src/datasets/elph.py (HashDataset, BUDDY's feature precompute) on the reference's own src/hashing.py
    and src/heuristics.py.  Stand-ins for what the image lacks and the precompute does not depend on: torch_geometric.data.Dataset
    (an empty base class: the reference only uses its constructor and len/get protocol), to_undirected / coalesce (unused: the
    fixture graphs are undirected and use_coalesce is False), gcn_norm + torch_sparse.spmm (the SIGN node features `x`, NOT part of
    this fixture -- that row stays pinned by G13's exporter -- are produced by a plain normalised product)."""
    install_shims(nan_bias)
    load_reference_buddy()  # stand-ins for the unused PyG imports; puts /root/reference on sys.path

    class Dataset(object):
        def __init__(self, root=None, *a, **kw):
            self.root = root

    tgd = types.ModuleType('torch_geometric.data')
    tgd.Dataset = Dataset
    sys.modules['torch_geometric.data'] = tgd
    sys.modules['torch_geometric.utils'].to_undirected = lambda ei, ew=None, *a, **kw: (_ for _ in ()).throw(NotImplementedError())
    ts = types.ModuleType('torch_sparse')
    ts.coalesce = lambda *a, **kw: (_ for _ in ()).throw(NotImplementedError())

    def spmm(index, value, m, n, matrix):
        out = torch.zeros((m, matrix.size(1)), dtype=matrix.dtype)
        return out.index_add_(0, index[0], value[:, None] * matrix[index[1]])
    ts.spmm = spmm
    sys.modules['torch_sparse'] = ts
    gc = types.ModuleType('torch_geometric.nn.conv.gcn_conv')

    def gcn_norm(edge_index, edge_weight, num_nodes):
        # torch_geometric.nn.conv.gcn_conv.gcn_norm with its defaults, RESTATED (the package is not in this image):
        # add_remaining_self_loops(fill_value=1) -- existing self loops leave the list, every node gets one loop behind all other
        # edges carrying its existing loop's weight (scatter assignment: the last one in edge order) or 1 --, deg = scatter_add
        # of the weights over `col`, deg^-1/2 with inf -> 0, norm = dis[row] * w * dis[col]
        row, col = edge_index[0], edge_index[1]
        keep = row != col
        loop_w = torch.ones(num_nodes)
        loop_w[row[~keep]] = edge_weight[~keep]
        loops = torch.arange(num_nodes).repeat(2, 1)
        ei = torch.cat([edge_index[:, keep], loops], dim=1)
        ew = torch.cat([edge_weight[keep], loop_w])
        deg = torch.zeros(num_nodes).index_add_(0, ei[1], ew)
        dis = deg.pow(-0.5)
        dis[torch.isinf(dis)] = 0
        return ei, dis[ei[0]] * ew * dis[ei[1]]
    gc.gcn_norm = gcn_norm
    sys.modules['torch_geometric.nn.conv.gcn_conv'] = gc
    for name in ('src.hashing', 'src.datasets.elph', 'src.datasets', 'src.heuristics'):
        sys.modules.pop(name, None)  # a fresh import against the stand-ins installed above (nan / real bias tables)
    import src.datasets.elph as ref_ds
    ref_ds.tqdm = lambda it, **kw: it
    return ref_ds


class _Data(object):
    """what HashDataset reads of a PyG Data object: attributes + `'edge_weight' in data`"""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, key):
        return key in self.__dict__


def make_g14():
    """G14: the outputs of the reference's OWN HashDataset (datasets/elph.py:26-85, 175-222: BUDDY's feature precompute, which
    BASELINE's north_star names) -- subgraph_features after its post-hoc floor / knock-out, degrees, RA, the cache file names --
    on the BA40 graph (h = 3) and the uniform 3000-node graph (h = 2), for floor_sf x use_zero_one; `*_uses_tables` from the same
    run with NaN bias tables.  VERDICT r2 missing #3: the reference itself was never run for this call."""
    import tempfile
    import warnings
    import scipy.sparse._index as _spi
    _orig_validate = _spi.IndexMixin._validate_indices

    def _validate_with_tensors(self, key, *a, **kw):  # (scipy >= 1.13 rejects torch tensors as indices: same values as numpy)
        conv = (lambda k: k.numpy() if isinstance(k, torch.Tensor) else k)
        key = tuple(conv(k) for k in key) if isinstance(key, tuple) else conv(key)
        return _orig_validate(self, key, *a, **kw)
    _spi.IndexMixin._validate_indices = _validate_with_tensors
    g = {}
    graphs = {'ba': (40, ba_graph(40, 5, seed=7), 3, 60), 'uni': (3000, uniform_graph(3000, 12000, seed=1), 2, 700)}
    for tag, (n, ei, h, n_pos) in graphs.items():
        rng = np.random.RandomState(14)
        pos = torch.from_numpy(ei[:, rng.permutation(ei.shape[1])[:n_pos]].T.copy())
        neg = torch.from_numpy(rng.randint(0, n, size=(n_pos, 2)).astype(np.int64))
        g[f'{tag}_edge_index'], g[f'{tag}_num_nodes'], g[f'{tag}_hops'] = ei, np.asarray(n), np.asarray(h)
        g[f'{tag}_pos'], g[f'{tag}_neg'] = pos.numpy(), neg.numpy()
        for fl in (0, 1):
            for zo in (0, 1):
                out = {}
                for nan_bias in (False, True):
                    ref_ds = load_reference_hash_dataset(nan_bias)
                    with tempfile.TemporaryDirectory() as tmp:
                        root = os.path.join(tmp, 'elph_')
                        a = Namespace(model='BUDDY', max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=bool(fl), use_zero_one=bool(zo),
                                      load_features=False, load_hashes=True, cache_subgraph_features=True, use_feature=True, use_RA=True,
                                      sign_k=0, num_negs=1, subgraph_feature_batch_size=11000000, dataset_name='synthetic', year=0,
                                      use_struct_feature=True)
                        data = _Data(edge_index=torch.from_numpy(ei), num_nodes=n, x=torch.ones((n, 4)))
                        with warnings.catch_warnings():
                            warnings.simplefilter('ignore')
                            ds = ref_ds.HashDataset(root, 'train', data, pos, neg, a)
                        files = sorted(os.listdir(tmp))
                        cached = torch.load(os.path.join(tmp, [f for f in files if f.endswith('subgraph_featurecache.pt')][0]))
                    out[nan_bias] = (ds, files, cached)
                ds, files, cached = out[False]
                key = f'{tag}_fl{fl}_zo{zo}'
                assert ds.subgraph_features.dtype == torch.float32 and ds.subgraph_features.shape == (2 * n_pos, h * (h + 2))
                g[f'{key}_subgraph_features'] = ds.subgraph_features.numpy()
                g[f'{key}_uses_tables'] = torch.isnan(out[True][0].subgraph_features).numpy()
                # what the reference wrote to its feature cache: the tensor BEFORE the post-hoc floor / knock-out? No: torch.save
                # stores the same tensor object HashDataset then edits in place only AFTER saving (datasets/elph.py:211-222)
                g[f'{key}_cached_features'] = cached.numpy()
                g[f'{key}_files'] = np.asarray(files)
                if fl == 0 and zo == 1:
                    g[f'{tag}_links'] = ds.links.numpy()
                    g[f'{tag}_degrees'] = ds.degrees.numpy()
                    g[f'{tag}_RA'] = ds.RA.numpy()
                    g[f'{tag}_labels'] = np.asarray(ds.labels)
    _spi.IndexMixin._validate_indices = _orig_validate
    np.savez_compressed(os.path.join(OUT, 'g14_hash_dataset.npz'), **g)
    print('G14 written:', {k: v.shape for k, v in g.items() if k.endswith('subgraph_features')}, g['ba_fl0_zo1_files'])


def make_g15():
    """G15: the reference's OWN HashDataset._generate_sign_features (datasets/elph.py:87-110) for sign_k in {0, 2} under the
    restated gcn_norm / spmm of load_reference_hash_dataset -- flagged "PyG semantics restated", exactly how A6 is pinned: it
    pins the reference-owned part (the sign_k loop that re-multiplies data.x, the concatenation, the float() of the weights)
    and the edge-order accumulation; the restatement itself waits for tools/export_pyg_fixture.py (G13).  Graphs: existing self
    loops (one node with two of different weight: the last one counts), duplicate edges, isolated nodes, a hub row; unit and
    non-integer weights (so that the ORDER of every fp32 sum matters); F = 12 (not a multiple of 4) and 64."""
    ref_ds = load_reference_hash_dataset(False)
    g = {}
    rng = np.random.RandomState(15)
    n, e = 300, 3000
    ei = rng.randint(0, n - 5, size=(2, e)).astype(np.int64)          # the last five nodes are isolated
    ei[:, :6] = np.array([[5, 5, 9, 11, 40, 40], [5, 5, 9, 11, 41, 41]])  # self loops (node 5 twice) and a duplicate edge
    ei[1, 100:400] = 17                                                 # a hub column / row after the flip below
    ei = np.concatenate([ei, ei[::-1]], axis=1)
    g['edge_index'], g['num_nodes'] = ei, np.asarray(n)
    weights = {'unit': np.ones(ei.shape[1], dtype=np.float32), 'float': (rng.random_sample(ei.shape[1]) * 3 + 0.1).astype(np.float32)}
    for F in (12, 64):
        x = rng.randn(n, F).astype(np.float32)
        g[f'x_F{F}'] = x
        for wname, w in weights.items():
            g[f'w_{wname}'] = w
            for k in (0, 2):
                data = _Data(x=torch.from_numpy(x), num_nodes=n)
                out = ref_ds.HashDataset._generate_sign_features(None, data, torch.from_numpy(ei), torch.from_numpy(w), k)
                assert out.dtype == torch.float32 and out.shape == (n, F * (1 if k == 0 else k + 1))
                g[f'sign_k{k}_F{F}_{wname}'] = out.numpy()
    g['note'] = np.asarray('outputs of /root/reference/src/datasets/elph.py HashDataset._generate_sign_features; gcn_norm / torch_sparse.spmm '
                           'are the restatements of tests/golden/make_golden.py (PyG and torch_sparse are not in this image)')
    np.savez_compressed(os.path.join(OUT, 'g15_sign_features.npz'), **g)
    print('G15 written:', sorted(k for k in g if k.startswith('sign_')))


def args(h=2, p=8, P=128, floor_sf=False, use_zero_one=True):
    return Namespace(max_hash_hops=h, hll_p=p, minhash_num_perm=P, floor_sf=floor_sf, use_zero_one=use_zero_one)


def ba_graph(n, mdeg, seed):
    """undirected Barabasi-Albert edge list (both directions), via networkx"""
    import networkx as nx
    g = nx.barabasi_albert_graph(n, mdeg, seed=seed)
    e = np.array(list(g.edges()), dtype=np.int64).T
    return np.concatenate([e, e[::-1]], axis=1)


def uniform_graph(n, e_und, seed):
    rng = np.random.RandomState(seed)
    e = rng.randint(0, n, size=(2, e_und)).astype(np.int64)
    return np.concatenate([e, e[::-1]], axis=1)


def table_arrays(tables, prefix, out):
    for k, d in tables.items():
        out[f'{prefix}_hll_{k}'] = d['hll'].numpy().astype(np.uint8)
        mh = d['minhash'].numpy()
        assert mh.min() >= 0 and mh.max() < (1 << 32)
        out[f'{prefix}_mh_{k}'] = mh.astype(np.uint32)


def counts(ref_eh, links, tables, h):
    """integer match counts / zero counts per (pair, k1, k2), from the reference's own tensors"""
    mc = np.zeros((links.shape[0], h, h), dtype=np.int32)
    zc = np.zeros((links.shape[0], h, h), dtype=np.int32)
    for k1 in range(1, h + 1):
        for k2 in range(1, h + 1):
            a = tables[k1]['minhash'][links[:, 0]]
            b = tables[k2]['minhash'][links[:, 1]]
            mc[:, k1 - 1, k2 - 1] = torch.count_nonzero(a == b, dim=-1).numpy()
            u = ref_eh._hll_merge(tables[k1]['hll'][links[:, 0]], tables[k2]['hll'][links[:, 1]])
            zc[:, k1 - 1, k2 - 1] = (u.shape[1] - torch.count_nonzero(u, dim=1)).numpy()
    return mc, zc


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def main():
    ref, add_self_loops = load_reference(False)
    refnan, _ = load_reference(True)
    ref, add_self_loops = load_reference(False)  # leave the real-table shim installed last
    torch.manual_seed(0)

    # ---- G1: permutation parameters ------------------------------------------------------
    g = {}
    for P in (8, 128):
        ab = ref.ElphHashes(args(P=P))._init_permutations(P)
        g[f'perm_a_P{P}'], g[f'perm_b_P{P}'] = ab[0], ab[1]
    # ---- G2: hop-0 sketches -----------------------------------------------------------------
    for p in (4, 8, 16):
        eh = ref.ElphHashes(args(p=p))
        g[f'init_hll_p{p}'] = eh.initialise_hll(64).numpy().astype(np.uint8)
    g['init_mh_P128'] = ref.ElphHashes(args()).initialise_minhash(64).numpy().astype(np.uint32)
    g['init_mh_P8'] = ref.ElphHashes(args(P=8)).initialise_minhash(64).numpy().astype(np.uint32)
    g['init_mh_P128_tail'] = ref.ElphHashes(args()).initialise_minhash(100000).numpy()[-16:].astype(np.uint32)
    g['init_hll_p8_tail_idx'] = np.argmax(ref.ElphHashes(args()).initialise_hll(100000).numpy()[-64:], axis=1)
    g['init_hll_p8_tail_val'] = np.max(ref.ElphHashes(args()).initialise_hll(100000).numpy()[-64:], axis=1)
    np.savez_compressed(os.path.join(OUT, 'g1_g2_init.npz'), **g)

    # ---- G3/G4: small BA graph, full tables + features -------------------------------------
    g = {}
    n = 40
    ei = ba_graph(n, 5, seed=7)
    g['edge_index'] = ei
    g['num_nodes'] = np.asarray(n)
    eh3 = ref.ElphHashes(args(h=3))
    tables, cards = eh3.build_hash_tables(n, torch.from_numpy(ei))
    table_arrays(tables, 't', g)
    g['cards'] = cards.numpy()
    links = torch.randint(0, n, (128, 2), generator=torch.Generator().manual_seed(3))
    links[0] = torch.tensor([0, 1])
    links[1] = torch.tensor([5, 5])
    g['links'] = links.numpy()
    for h in (1, 2, 3):
        sub = {k: tables[k] for k in range(h + 1)}
        mc, zc = counts(eh3, links, tables, h)
        g[f'match_h{h}'], g[f'zeros_h{h}'] = mc, zc
        for zo in (0, 1):
            for fl in (0, 1):
                a = args(h=h, floor_sf=bool(fl), use_zero_one=bool(zo))
                f = ref.ElphHashes(a).get_subgraph_features(links, sub, cards[:, :h])
                fn = refnan.ElphHashes(a).get_subgraph_features(links, sub, cards[:, :h])
                assert f.dtype == torch.float32
                g[f'feat_h{h}_zo{zo}_fl{fl}'] = f.numpy()
                g[f'feat_h{h}_zo{zo}_fl{fl}_uses_tables'] = torch.isnan(fn).numpy()
        inter = ref.ElphHashes(args(h=h))._get_intersections(links, sub)
        g[f'inter_h{h}'] = np.stack([inter[(k1, k2)].numpy() for k1 in range(1, h + 1) for k2 in range(1, h + 1)], 1)
    # batched == unbatched, 1-D link
    a = args(h=2)
    g['feat_h2_batched7'] = ref.ElphHashes(a).get_subgraph_features(links, {k: tables[k] for k in range(3)},
                                                                   cards[:, :2], batch_size=7).numpy()
    g['feat_h2_1d'] = ref.ElphHashes(a).get_subgraph_features(links[0], {k: tables[k] for k in range(3)},
                                                              cards[:, :2]).numpy()
    np.savez_compressed(os.path.join(OUT, 'g3_g4_ba40.npz'), **g)
    # the reference's own on-disk cache format (datasets/elph.py:200-204: torch.save of the hashes dict and of cards),
    # written by the reference's objects themselves: the --load_hashes compatibility fixture
    torch.save({k: tables[k] for k in range(3)}, os.path.join(OUT, 'ref_ba40_hashcache.pt'))
    torch.save(cards[:, :2].clone(), os.path.join(OUT, 'ref_ba40_cardcache.pt'))

    # ---- G9: BUDDY's degree-normalised copy (models/elph.py:276-293), computed by the reference's own method -----------
    buddy = load_reference_buddy()
    deg = np.bincount(ei[1], minlength=n).astype(np.float32)   # HashDataset.degrees = A.sum(axis=0) (datasets/elph.py:74)
    deg[[3, 17]] = 0.0                                         # zero-degree nodes exercise the NaN / Inf -> 0 rule
    g9 = {'degrees': deg, 'links': links.numpy()}
    for h in (1, 2, 3):
        sf = torch.from_numpy(g[f'feat_h{h}_zo1_fl0'])
        d = torch.from_numpy(deg)
        g9[f'normed_h{h}'] = buddy._append_degree_normalised(None, sf, d[links[:, 0]], d[links[:, 1]]).numpy()
    np.savez_compressed(os.path.join(OUT, 'g9_degree_normalised.npz'), **g9)

    # ---- G3b: other (p, P) parameterisations on the same graph ------------------------------
    g = {'edge_index': ei, 'num_nodes': np.asarray(n), 'links': links.numpy()}
    for (p, P) in ((4, 8), (16, 128), (6, 64)):
        e = ref.ElphHashes(args(h=2, p=p, P=P))
        en = refnan.ElphHashes(args(h=2, p=p, P=P))
        tb, cd = e.build_hash_tables(n, torch.from_numpy(ei))
        _, cdn = en.build_hash_tables(n, torch.from_numpy(ei))
        table_arrays(tb, f'p{p}P{P}', g)
        g[f'p{p}P{P}_cards'] = cd.numpy()
        g[f'p{p}P{P}_cards_uses_tables'] = torch.isnan(cdn).numpy()
        g[f'p{p}P{P}_feat'] = e.get_subgraph_features(links, tb, cd).numpy()
        g[f'p{p}P{P}_feat_uses_tables'] = torch.isnan(en.get_subgraph_features(links, tb, cdn)).numpy()
    np.savez_compressed(os.path.join(OUT, 'g3b_params.npz'), **g)

    # ---- G5: hll_count known answers ------------------------------------------------------------
    g = {}
    eh = ref.ElphHashes(args())
    ehn = refnan.ElphHashes(args())
    rng = np.random.RandomState(11)
    rows = [np.zeros(256, np.int8)]
    r = np.zeros(256, np.int8); r[17] = 3; rows.append(r)
    for nz in (110, 109, 108, 107, 64, 1):
        r = rng.randint(1, 12, size=256).astype(np.int8); r[rng.permutation(256)[:nz]] = 0; rows.append(r)
    for lo, hi in ((1, 3), (1, 6), (2, 8), (4, 10), (6, 14), (1, 57)):
        rows.append(rng.randint(lo, hi, size=256).astype(np.int8))
    # geometric-looking rows at many fill levels (what real unions look like)
    for load in (0.3, 0.7, 1.0, 1.5, 2.0, 3.0, 4.0, 4.9, 5.0, 5.1, 6.0, 8.0, 20.0, 100.0):
        k = rng.multinomial(int(load * 256), np.full(256, 1 / 256.0))
        u = rng.random_sample(256)
        x = 1.0 - np.power(u, 1.0 / np.maximum(k, 1))
        rr = np.clip(np.ceil(-np.log2(np.maximum(x, 2.0 ** -60))), 1, 56)
        rows.append(np.where(k > 0, rr, 0).astype(np.int8))
    regs = torch.from_numpy(np.stack(rows))
    g['regs'] = regs.numpy().astype(np.uint8)
    g['count'] = eh.hll_count(regs).numpy()
    g['count_uses_tables'] = torch.isnan(ehn.hll_count(regs)).numpy()
    g['count_1d'] = eh.hll_count(regs[1]).numpy()
    g['count_int64'] = eh.hll_count(regs.long()).numpy()
    np.savez_compressed(os.path.join(OUT, 'g5_hll_count.npz'), **g)

    # ---- G7: edge cases ----------------------------------------------------------------------------
    g = {}
    n = 12  # nodes 9..11 are trailing isolated nodes: no self-loop is added for them (hashing.py:148)
    ei7 = np.array([[0, 1, 1, 2, 3, 4, 8, 0, 5, 5], [1, 0, 2, 1, 4, 3, 0, 8, 5, 6]], dtype=np.int64)  # incl. self loop 5-5, one-way 5->6
    g['edge_index'] = ei7
    g['num_nodes'] = np.asarray(n)
    e = ref.ElphHashes(args(h=2))
    tb, cd = e.build_hash_tables(n, torch.from_numpy(ei7))
    table_arrays(tb, 't', g)
    g['cards'] = cd.numpy()
    lk = torch.tensor([[0, 1], [9, 10], [11, 0], [5, 6], [6, 5], [7, 7]])
    g['links'] = lk.numpy()
    g['feat'] = e.get_subgraph_features(lk, tb, cd).numpy()
    np.savez_compressed(os.path.join(OUT, 'g7_edge_cases.npz'), **g)

    # ---- G8: medium graph reaching every estimator branch -----------------------------------------
    g = {}
    n, e_und = 3000, 12000
    ei8 = uniform_graph(n, e_und, seed=1)
    g['graph'] = np.asarray([n, e_und, 1])
    e = ref.ElphHashes(args(h=3))
    en = refnan.ElphHashes(args(h=3))
    tb, cd = e.build_hash_tables(n, torch.from_numpy(ei8))
    _, cdn = en.build_hash_tables(n, torch.from_numpy(ei8))
    g['cards'] = cd.numpy()
    g['cards_uses_tables'] = torch.isnan(cdn).numpy()
    for k in range(4):
        g[f'sha_hll_{k}'] = np.asarray(sha(tb[k]['hll'].numpy().astype(np.uint8)))
        g[f'sha_mh_{k}'] = np.asarray(sha(tb[k]['minhash'].numpy().astype(np.uint32)))
    lk = torch.from_numpy(np.random.RandomState(2).randint(0, n, size=(512, 2)).astype(np.int64))
    lk[:128] = torch.from_numpy(ei8[:, :128].T.copy())  # positive pairs (high Jaccard)
    g['links'] = lk.numpy()
    for h in (2, 3):
        sub = {k: tb[k] for k in range(h + 1)}
        mc, zc = counts(e, lk, tb, h)
        g[f'match_h{h}'], g[f'zeros_h{h}'] = mc, zc
        a = args(h=h)
        g[f'feat_h{h}'] = ref.ElphHashes(a).get_subgraph_features(lk, sub, cd[:, :h]).numpy()
        g[f'feat_h{h}_uses_tables'] = torch.isnan(
            refnan.ElphHashes(a).get_subgraph_features(lk, sub, cdn[:, :h])).numpy()
    np.savez_compressed(os.path.join(OUT, 'g8_uniform3000.npz'), **g)

    # ---- G6: digests of a collab-scale build --------------------------------------------------------
    if '--big' in sys.argv:
        g = {}
        n, e_und = 235868, 1285465
        ei6 = uniform_graph(n, e_und, seed=1)
        e = ref.ElphHashes(args(h=2))
        tb, cd = e.build_hash_tables(n, torch.from_numpy(ei6))
        g['graph'] = np.asarray([n, e_und, 1])
        for k in range(3):
            g[f'sha_hll_{k}'] = np.asarray(sha(tb[k]['hll'].numpy().astype(np.uint8)))
            g[f'sha_mh_{k}'] = np.asarray(sha(tb[k]['minhash'].numpy().astype(np.uint32)))
        _, cdn = refnan.ElphHashes(args(h=2)).build_hash_tables(n, torch.from_numpy(ei6))
        keep = ~torch.isnan(cdn)
        g['cards_sha_table_independent'] = np.asarray(sha(torch.where(keep, cd, torch.zeros_like(cd)).numpy()))
        g['cards_sample'] = cd[:4096].numpy()
        g['cards_sample_uses_tables'] = torch.isnan(cdn[:4096]).numpy()
        np.savez_compressed(os.path.join(OUT, 'g6_collab_scale_digests.npz'), **g)
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    if '--only-g10' in sys.argv:
        make_g10()
    elif '--only-g12' in sys.argv:
        make_g12()
    elif '--only-g14' in sys.argv:
        make_g14()
    elif '--only-g15' in sys.argv:
        make_g15()
    else:
        main()
        make_g10()
        make_g12()
        make_g14()
        make_g15()
