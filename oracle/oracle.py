"""numpy/ctypes front end of the C oracle (oracle/sketch_oracle.c).

TEST INFRASTRUCTURE: the checker for the HIP engine, never the thing shipped or measured (except as the
explicitly labelled `cpu_baseline` of bench.py).  Each function names the reference code it restates
(/root/reference/src/hashing.py:line).  Parity pinning and the "parity unpinned" caveat for the HLL++
bias tables are described in the header of sketch_oracle.c.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int32, c_int64, c_uint32, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'liboracle.so')


class _Params(ctypes.Structure):
    _fields_ = [('p', c_int32), ('n_tbl', c_int32), ('alpha_mm', c_float), ('threshold', c_float),
                ('raw_est', c_void_p), ('bias', c_void_p), ('lc_table', c_void_p)]


def build(force=False):
    src = os.path.join(_HERE, 'sketch_oracle.c')
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.so_hll_init.restype = c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else c_void_p(0)


def hll_alpha(p):
    m = 1 << p
    return {4: 0.673, 5: 0.697, 6: 0.709}.get(p, 0.7213 / (1.0 + 1.079 / m))


class HllParams(object):
    """estimator constants in the reference's own (unsorted) table order.
    lc_table (optional, fp32 [m+1]): linear-counting values as the host torch computes them."""

    def __init__(self, p, threshold, raw_estimate, bias, alpha=None, lc_table=None):
        self.p, self.m = p, 1 << p
        self.raw = np.ascontiguousarray(raw_estimate, dtype=np.float32)
        self.bias = np.ascontiguousarray(bias, dtype=np.float32)
        self.lc = np.ascontiguousarray(lc_table, dtype=np.float32) if lc_table is not None else None
        alpha = hll_alpha(p) if alpha is None else alpha
        self.struct = _Params(p=p, n_tbl=len(self.raw), alpha_mm=float(np.float32(alpha * self.m ** 2)),
                              threshold=float(np.float32(threshold)), raw_est=self.raw.ctypes.data,
                              bias=self.bias.ctypes.data, lc_table=self.lc.ctypes.data if self.lc is not None else None)


def init_permutations(num_perm, seed=1):
    """reference hashing.py:106-116 (numpy legacy RandomState, draws interleaved a0,b0,a1,b1,...)"""
    gen = np.random.RandomState(seed)
    prime = np.uint64((1 << 61) - 1)
    ab = np.array([(gen.randint(1, prime, dtype=np.uint64), gen.randint(0, prime, dtype=np.uint64))
                   for _ in range(num_perm)], dtype=np.uint64).T
    return np.ascontiguousarray(ab[0]), np.ascontiguousarray(ab[1])


def hash_nodes(n, first_node=0):
    out = np.empty(n, dtype=np.uint64)
    lib().so_hash_nodes(c_int64(first_node), c_int64(n), _p(out))
    return out


def minhash_init(n, num_perm, first_node=0):
    a, b = init_permutations(num_perm)
    out = np.empty((n, num_perm), dtype=np.uint32)
    lib().so_minhash_init(c_int64(first_node), c_int64(n), c_int32(num_perm), _p(a), _p(b), _p(out))
    return out


def hll_init(n, p, first_node=0):
    out = np.empty((n, 1 << p), dtype=np.uint8)
    bad = lib().so_hll_init(c_int64(first_node), c_int64(n), c_int32(p), _p(out))
    if bad:
        raise ValueError("Hash value overflow, maximum size is %d bits" % (64 - p))
    return out


def add_self_loops(edge_index, num_nodes=None):
    """torch_geometric.utils.add_self_loops as used at hashing.py:148 (N inferred from max id when omitted)"""
    edge_index = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.size else 0
    loops = np.arange(num_nodes, dtype=np.int64)
    return np.concatenate([edge_index, np.stack([loops, loops])], axis=1)


def propagate(num_nodes, edge_index, mh=None, hll=None):
    """one hop over the given (already self-looped) edge list: hashing.py:28-45"""
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    mh_out = np.empty_like(mh) if mh is not None else None
    hll_out = np.empty_like(hll) if hll is not None else None
    lib().so_propagate_edges(c_int64(num_nodes), c_int64(src.size), _p(src), _p(dst), _p(mh), _p(mh_out),
                             c_int32(mh.shape[1] if mh is not None else 0), _p(hll), _p(hll_out),
                             c_int32(hll.shape[1] if hll is not None else 0))
    return mh_out, hll_out


def csr_build(num_nodes, edge_index):
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    src, dst = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    rowptr = np.empty(num_nodes + 1, dtype=np.int64)
    col = np.empty(max(src.size, 1), dtype=np.int32)
    lib().so_csr_build(c_int64(num_nodes), c_int64(src.size), _p(src), _p(dst), _p(rowptr), _p(col))
    return rowptr, col


def propagate_csr(num_nodes, rowptr, col, n_self, mh=None, hll=None, params=None):
    """CSR pull version with implicit self loops and fused cardinalities (multi-threaded)"""
    mh_out = np.empty_like(mh) if mh is not None else None
    hll_out = np.empty_like(hll) if hll is not None else None
    cards = np.empty(num_nodes, dtype=np.float32) if (params is not None and hll is not None) else None
    lib().so_propagate_csr(c_int64(num_nodes), _p(rowptr), _p(col), c_int64(n_self), _p(mh), _p(mh_out),
                           c_int32(mh.shape[1] if mh is not None else 0), _p(hll), _p(hll_out),
                           c_int32(hll.shape[1] if hll is not None else 0), _p(cards), c_int64(1),
                           ctypes.byref(params.struct) if params is not None else None)
    return mh_out, hll_out, cards


def hll_count(regs, params, return_branch=False):
    """hashing.py:212-232; regs [n, m] (or [m]) of uint8/int8 or int64"""
    regs = np.asarray(regs)
    if regs.ndim == 1:
        regs = regs[None]
    if regs.dtype.itemsize not in (1, 8):
        regs = regs.astype(np.int64)
    regs = np.ascontiguousarray(regs)
    out = np.empty(regs.shape[0], dtype=np.float32)
    br = np.empty(regs.shape[0], dtype=np.int32)
    lib().so_hll_count(_p(regs), c_int32(regs.dtype.itemsize), c_int64(regs.shape[0]), ctypes.byref(params.struct), _p(out),
                       _p(br))
    return (out, br) if return_branch else out


def estimate_bias(e, params, refine=False):
    """hashing.py:197-204 (_estimate_bias) / :206-210 (_refine_hll_count_estimate, refine=True) for raw estimates e [n]"""
    e = np.ascontiguousarray(e, dtype=np.float32)
    out = np.empty_like(e)
    lib().so_estimate_bias(_p(e), c_int64(e.size), ctypes.byref(params.struct), c_int32(int(refine)), _p(out))
    return out


def build_hash_tables(num_nodes, edge_index, max_hops, num_perm, params):
    """hashing.py:139-165.  returns ({k: {'hll': uint8 [N,m], 'minhash': uint32 [N,P]}}, cards fp32 [N, max_hops])"""
    ei = add_self_loops(edge_index)
    tables = {0: {'minhash': minhash_init(num_nodes, num_perm), 'hll': hll_init(num_nodes, params.p)}}
    cards = np.zeros((num_nodes, max_hops), dtype=np.float32)
    for k in range(1, max_hops + 1):
        mh, hll = propagate(num_nodes, ei, tables[k - 1]['minhash'], tables[k - 1]['hll'])
        tables[k] = {'minhash': mh, 'hll': hll}
        cards[:, k - 1] = hll_count(hll, params)
    return tables, cards


def pair_features(links, tables, cards, max_hops, params, use_zero_one=True, floor_sf=False, debug=False):
    """hashing.py:167-189 + 258-323 for all links at once.  tables as returned by build_hash_tables."""
    links = np.ascontiguousarray(np.asarray(links, dtype=np.int64).reshape(-1, 2))
    B, h = links.shape[0], max_hops
    mh = [np.ascontiguousarray(tables[k]['minhash'], dtype=np.uint32) for k in range(1, h + 1)]
    hl = [np.ascontiguousarray(tables[k]['hll']).view(np.uint8) for k in range(1, h + 1)]
    N, P = mh[0].shape
    cards = np.ascontiguousarray(cards, dtype=np.float32)
    out = np.empty((B, h * (h + 2)), dtype=np.float32)
    dbg = {}
    if debug:
        dbg = {'match': np.empty((B, h, h), np.int32), 'zeros': np.empty((B, h, h), np.int32),
               'inter': np.empty((B, h, h), np.float32), 'branch': np.empty((B, h, h), np.int32)}
    mh_ptrs = (c_void_p * h)(*[a.ctypes.data for a in mh])
    hl_ptrs = (c_void_p * h)(*[a.ctypes.data for a in hl])
    flags = (1 if use_zero_one else 0) | (2 if floor_sf else 0)
    lib().so_pair_features(_p(links), c_int64(B), c_int64(N), c_int32(h), mh_ptrs, c_int32(P), hl_ptrs, _p(cards),
                           c_int64(cards.shape[1]), ctypes.byref(params.struct), c_uint32(flags), _p(out),
                           _p(dbg.get('match')), _p(dbg.get('zeros')), _p(dbg.get('inter')), _p(dbg.get('branch')))
    return (out, dbg) if debug else out


def append_degree_normalised(x, links, degrees):
    """models/elph.py:276-293 on top of pair_features output: [B, nf] -> [B, 2*nf]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    links = np.ascontiguousarray(np.asarray(links, dtype=np.int64).reshape(-1, 2))
    degrees = np.ascontiguousarray(degrees, dtype=np.float32)
    out = np.empty((x.shape[0], 2 * x.shape[1]), dtype=np.float32)
    lib().so_append_degree_normalised(_p(x), c_int64(x.shape[0]), c_int32(x.shape[1]), _p(links), c_int64(degrees.shape[0]),
                                      _p(degrees), _p(out))
    return out


def common_neighbour_scores(A, links, kind):
    """heuristics.py:10-70: kind in {'CN', 'AA', 'RA'}; A scipy sparse (any format), links [L, 2] -> float32 [L]"""
    A = A.tocsr().copy()
    A.sum_duplicates()
    rowptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    col = np.ascontiguousarray(A.indices, dtype=np.int32) if A.nnz else np.zeros(1, dtype=np.int32)
    val = np.ascontiguousarray(A.data, dtype=np.float64) if A.nnz else np.zeros(1, dtype=np.float64)
    mult = None
    if kind != 'CN':
        colsum = np.asarray(A.sum(axis=0)).ravel()
        with np.errstate(divide='ignore', invalid='ignore'):
            mult = 1 / (np.log(colsum) if kind == 'AA' else colsum)
        mult = np.asarray(mult, dtype=np.float64)
        mult[np.isinf(mult)] = 0
        mult = np.ascontiguousarray(mult)
    links = np.ascontiguousarray(np.asarray(links, dtype=np.int64).reshape(-1, 2))
    out = np.empty(links.shape[0], dtype=np.float32)
    lib().so_common_neighbour_scores(_p(rowptr), _p(col), _p(val), _p(mult), c_int64(A.shape[0]), _p(links), c_int64(links.shape[0]),
                                     _p(out))
    return out


def gcn_norm(edge_index, edge_weight, num_nodes):
    """PyG gcn_norm defaults restated in numpy (PyG is absent here: unpinned): remaining self loops with weight 1 (existing
    self loops keep theirs), symmetric normalisation by the weighted in-degree over edge_index[1]"""
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    w = np.asarray(edge_weight, dtype=np.float32)
    keep = ei[0] != ei[1]
    loop_w = np.ones(num_nodes, dtype=np.float32)
    loop_w[ei[0][~keep]] = w[~keep]
    loops = np.arange(num_nodes, dtype=np.int64)
    ei2 = np.concatenate([ei[:, keep], np.stack([loops, loops])], axis=1)
    w2 = np.concatenate([w[keep], loop_w]).astype(np.float32)
    deg = np.zeros(num_nodes, dtype=np.float32)
    np.add.at(deg, ei2[1], w2)
    with np.errstate(divide='ignore', invalid='ignore'):
        dinv = (np.float32(1.0) / np.sqrt(deg)).astype(np.float32)  # torch evaluates pow(-0.5) as the reciprocal of the square root
    dinv[np.isinf(dinv)] = 0
    return ei2, (dinv[ei2[0]] * w2 * dinv[ei2[1]]).astype(np.float32)


def generate_sign_features(x, edge_index, edge_weight, sign_k):
    """HashDataset._generate_sign_features (reference datasets/elph.py:87-110): gcn_norm, then spmm of data.x -- for sign_k > 0
    the SAME product sign_k times behind x itself (the loop multiplies data.x every time, :105-107)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    ei, wn = gcn_norm(edge_index, np.asarray(edge_weight, dtype=np.float32), n)
    ax = spmm(ei, wn, n, x)
    return ax if sign_k == 0 else np.concatenate([x] + [ax] * sign_k, axis=1)


def spmm(index, value, n, x):
    """torch_sparse.spmm: sequential fp32 scatter-add of value * x[index[1]] into rows index[0]"""
    row = np.ascontiguousarray(index[0], dtype=np.int64)
    col = np.ascontiguousarray(index[1], dtype=np.int64)
    val = np.ascontiguousarray(value, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty((n, x.shape[1]), dtype=np.float32)
    lib().so_spmm_coo(_p(row), _p(col), _p(val), c_int64(row.size), c_int64(n), _p(x), c_int32(x.shape[1]), _p(out))
    return out
