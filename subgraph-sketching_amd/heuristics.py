"""Host mirror of the reference's per-link neighbourhood heuristics (src/heuristics.py:10-70: CN, AA, RA) -- SURVEY 8(f)
row N4: `RA` is the other per-link precompute of HashDataset.__init__ (datasets/elph.py:76-77, negatives at :314).

Same call surface as the reference: `RA(A, edge_index, batch_size)` with A a scipy sparse adjacency matrix and
`edge_index` an int tensor of links [L, 2]; returns `(float32 scores [L], edge_index)`.  The three scores are one kernel
(ss_common_neighbour_scores) with a different column multiplier; the multiplier itself is computed on the host with the
reference's own numpy expression, so its fp64 values are identical.  No CPU fallback: the kernels need a HIP device.
"""
import logging
import weakref

import numpy as np
import torch

from . import _native
from .hashing import _compute_device, _error_flag, _ptr, _stream, _take_error

logger = logging.getLogger(__name__)


class DeviceAdjacency(object):
    """scipy sparse matrix -> device CSR with sorted, duplicate-free rows (the canonical form scipy itself computes with)
    plus fp64 values; built once per matrix object and reused by CN / AA / RA"""

    def __init__(self, A, device):
        A = A.tocsr()
        if not A.has_canonical_format:
            A = A.copy()
            A.sum_duplicates()  # also sorts the column ids of every row
        if A.shape[0] != A.shape[1]:
            raise ValueError('adjacency matrix must be square')
        self.num_nodes = A.shape[0]
        self.device = device
        self.rowptr = torch.from_numpy(A.indptr.astype(np.int64)).to(device)
        self.col = torch.from_numpy(A.indices.astype(np.int32) if A.nnz else np.zeros(1, dtype=np.int32)).to(device)
        unit = A.nnz == 0 or bool(np.all(A.data == 1))
        self.val = None if unit else torch.from_numpy(A.data.astype(np.float64)).to(device)
        # column sums exactly as the reference forms them (heuristics.py:38,59): np.matrix of the matrix's dtype
        self.colsum = np.asarray(A.sum(axis=0)).ravel()

    def multiplier(self, kind):
        if kind == 'CN':
            return None
        with np.errstate(divide='ignore', invalid='ignore'):
            mult = 1 / (np.log(self.colsum) if kind == 'AA' else self.colsum)
        mult = np.asarray(mult, dtype=np.float64)
        mult[np.isinf(mult)] = 0
        return torch.from_numpy(mult).to(self.device)


_ADJ_CACHE = {}


def _adjacency(A, device):
    if isinstance(A, DeviceAdjacency):
        return A
    key = (id(A), str(device))
    hit = _ADJ_CACHE.get(key)
    if hit is not None and hit[0]() is A and hit[2] == (A.nnz, A.shape):
        return hit[1]
    adj = DeviceAdjacency(A, device)
    try:
        _ADJ_CACHE[key] = (weakref.ref(A, lambda _r, k=key: _ADJ_CACHE.pop(k, None)), adj, (A.nnz, A.shape))
    except TypeError:  # not weak-referenceable: do not cache
        pass
    return adj


def _scores(kind, A, edge_index, batch_size):
    links = torch.as_tensor(edge_index)
    if links.dim() != 2 or links.size(1) != 2:
        raise ValueError('edge_index must be a tensor of links with shape [num_links, 2]')
    home = links.device
    device = _compute_device(links)
    adj = _adjacency(A, device)
    mult = adj.multiplier(kind)
    lk = links.to(device=device, dtype=torch.int64).contiguous()
    L = lk.size(0)
    out = torch.empty(L, dtype=torch.float32, device=device)
    err = _error_flag(device)
    lib = _native.lib()
    step = max(int(batch_size), 1)
    for lo in range(0, L, step):  # the reference's DataLoader chunks (heuristics.py:18,41,62); results do not depend on it
        hi = min(lo + step, L)
        _native.check(lib.ss_common_neighbour_scores(_ptr(adj.rowptr), _ptr(adj.col), _ptr(adj.val), _ptr(mult), adj.num_nodes,
                                                     _ptr(lk[lo:hi]), hi - lo, _ptr(out[lo:hi]), _ptr(err), _stream(device)),
                      'ss_common_neighbour_scores')
    if _take_error(device):
        raise IndexError(f'edge_index refers to nodes outside [0, {adj.num_nodes})')
    return out.to(home), edge_index


def CN(A, edge_index, batch_size=100000):
    """common neighbours (reference heuristics.py:10-27)"""
    scores, edge_index = _scores('CN', A, edge_index, batch_size)
    logger.info(f'evaluated Common Neighbours for {len(scores)} edges')
    return scores, edge_index


def AA(A, edge_index, batch_size=100000):
    """Adamic Adar (reference heuristics.py:30-48)"""
    scores, edge_index = _scores('AA', A, edge_index, batch_size)
    logger.info(f'evaluated Adamic Adar for {len(scores)} edges')
    return scores, edge_index


def RA(A, edge_index, batch_size=100000):
    """resource allocation (reference heuristics.py:51-70)"""
    scores, edge_index = _scores('RA', A, edge_index, batch_size)
    logger.info(f'evaluated Resource Allocation for {len(scores)} edges')
    return scores, edge_index
