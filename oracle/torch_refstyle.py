"""Reference-STYLE CPU path in stock torch ops -- the second CPU baseline of bench.py.

TEST / BASELINE INFRASTRUCTURE (never on the product path).  The C oracle (sketch_oracle.c) is a tight OpenMP port and
therefore much faster than what a user of the reference actually runs.  This module restates the reference's
*dataflow* with the same stock torch operators it dispatches to on a CPU, so the cost it measures is representative:

  * propagation (hashing.py:28-45 through PyG MessagePassing(aggr='max')): one materialised message x[src] per edge of the
    self-looped edge list, then a scatter-amax by destination; MinHash travels as int64 and is negated around the max;
  * hll_count (hashing.py:212-232): count_nonzero, log, 2.0 ** (-regs) in fp32, row sums, and the 6-nearest bias
    lookup through a full argsort of the [n, table] squared-distance matrix (hashing.py:203);
  * pair features (hashing.py:167-189, 258-323): for every (k1, k2) four row gathers, ==, count_nonzero, maximum.

Checked against the C oracle in tests/test_oracle_golden.py (integers bit-exact, floats within tolerance).
"""
import numpy as np
import torch


def init_sketches(num_nodes, num_perm, p, seed=1):
    """hop-0 sketches the way the reference makes them on the host (hashing.py:106-137): pandas' hash_array of the ids 1..N,
    one broadcast numpy expression over [N, P] uint64 for the permuted hashes (with the int64 ones-matrix and the minimum
    the reference applies), numpy scatter for the HLL registers, then torch.tensor copies.  Part of the reference's build
    time (3.9 s of 13.3 s at ogbl-collab size on 8 cores), so the reference-style baseline times it too."""
    from pandas.util import hash_array
    prime, max_hash = np.uint64((1 << 61) - 1), np.uint64((1 << 32) - 1)
    gen = np.random.RandomState(seed)
    ab = np.array([(gen.randint(1, prime, dtype=np.uint64), gen.randint(0, prime, dtype=np.uint64)) for _ in range(num_perm)],
                  dtype=np.uint64).T
    ceiling = np.ones((num_nodes, num_perm), dtype=np.int64) * max_hash
    hv = hash_array(np.arange(1, num_nodes + 1))
    permuted = np.bitwise_and((ab[0] * np.expand_dims(hv, 1) + ab[1]) % prime, max_hash)
    mh0 = torch.tensor(np.minimum(permuted, ceiling), dtype=torch.int64)
    m = 1 << p
    regs = np.zeros((num_nodes, m), dtype=np.int8)
    hv = hash_array(np.arange(1, num_nodes + 1))
    bits = hv >> np.uint64(p)
    rank = (64 - p) - np.ceil(np.log2(bits.astype(np.float64) + 1.0)).astype(np.int64) + 1   # reference :83-104 (float bit length)
    rows = np.arange(num_nodes)
    idx = (hv & np.uint64(m - 1)).astype(np.int64)
    regs[rows, idx] = np.maximum(regs[rows, idx], rank)
    return mh0, torch.tensor(regs, dtype=torch.int8)


def scatter_max_propagate(x, src, dst):
    """out[i] = max over edges (j -> i) of x[j]; rows without an in-edge are 0"""
    out = torch.zeros_like(x)
    if src.numel() == 0:
        return out
    return out.scatter_reduce(0, dst.unsqueeze(1).expand(-1, x.size(1)), x[src], 'amax', include_self=False)


def cardinality(regs, p, alpha, threshold, raw_estimate, bias):
    m = 1 << p
    out = torch.full((regs.shape[0],), float(threshold) + 1.0)
    zeros = m - torch.count_nonzero(regs, dim=1)
    has_zero = zeros > 0
    out[has_zero] = m * torch.log(m / zeros[has_zero])
    use_raw = out > threshold
    e = (alpha * m ** 2) / torch.sum(2.0 ** (-regs[use_raw]), dim=1)
    small = e <= 5 * m  # only these estimates go through the bias lookup (hashing.py:206-210)
    nearest = torch.argsort((e[small].unsqueeze(-1) - raw_estimate) ** 2)[:, :6]
    e[small] = e[small] - torch.mean(bias[nearest], dim=1)
    out[use_raw] = e
    return out


def build_tables(num_nodes, edge_index, max_hops, mh0, hll0, p, alpha, threshold, raw_estimate, bias, hops_to_run=None):
    """mh0 int64 [N,P], hll0 int8 [N,m] (hop-0 sketches).  Returns ({k: {...}}, cards) like the reference."""
    n_loops = int(edge_index.max()) + 1 if edge_index.numel() else 0
    loops = torch.arange(n_loops, dtype=edge_index.dtype)
    src = torch.cat([edge_index[0], loops])
    dst = torch.cat([edge_index[1], loops])
    tables = {0: {'minhash': mh0, 'hll': hll0}}
    cards = torch.zeros((num_nodes, max_hops))
    for k in range(1, (hops_to_run or max_hops) + 1):
        hll = scatter_max_propagate(tables[k - 1]['hll'], src, dst)
        mh = -scatter_max_propagate(-tables[k - 1]['minhash'], src, dst)
        tables[k] = {'minhash': mh, 'hll': hll}
        cards[:, k - 1] = cardinality(hll, p, alpha, threshold, raw_estimate, bias)
    return tables, cards


def pair_intersections(links, tables, max_hops, num_perm, p, alpha, threshold, raw_estimate, bias):
    inter = {}
    for k1 in range(1, max_hops + 1):
        for k2 in range(1, max_hops + 1):
            a_h, a_m = tables[k1]['hll'][links[:, 0]], tables[k1]['minhash'][links[:, 0]]
            b_h, b_m = tables[k2]['hll'][links[:, 1]], tables[k2]['minhash'][links[:, 1]]
            jac = torch.count_nonzero(a_m == b_m, dim=-1) / num_perm
            inter[(k1, k2)] = jac * cardinality(torch.maximum(a_h, b_h), p, alpha, threshold, raw_estimate, bias)
    return inter
