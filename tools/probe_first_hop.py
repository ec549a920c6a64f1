"""probe: first-hop kernel variants (both / mh only / hll only) and per-sketch propagate, single stream"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import byref
import numpy as np, torch
import bench
import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _ptr, _stream, build_csr, _propagate
from argparse import Namespace
dev = torch.device('cuda:0')
n = bench.N_NODES
ei = torch.from_numpy(bench.synthetic_graph()).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
csr = build_csr(ei, n, dev); csr.use_inferred_self_loops = True
prm = eh._params(dev)
mh = torch.empty((n, 128), dtype=torch.int32, device=dev); hll = torch.empty((n, 256), dtype=torch.uint8, device=dev)
mh2 = torch.empty_like(mh); hll2 = torch.empty_like(hll)
cards = torch.empty((n, 2), device=dev)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print('first_hop both', timeit(lambda: eh._first_hop(csr, dev, mh, hll, cards, prm)))
print('first_hop mh  ', timeit(lambda: eh._first_hop(csr, dev, mh, None, None, prm)))
print('first_hop hll ', timeit(lambda: eh._first_hop(csr, dev, None, hll, cards, prm)))
print('prop both     ', timeit(lambda: _propagate(csr, mh, hll, dev, cards_out=cards[:, 1], cards_stride=2, params=prm, mh_out=mh2, hll_out=hll2)))
print('prop mh       ', timeit(lambda: _propagate(csr, mh, None, dev, mh_out=mh2)))
print('prop hll+cards', timeit(lambda: _propagate(csr, None, hll, dev, cards_out=cards[:, 1], cards_stride=2, params=prm, hll_out=hll2)))
