"""probe: what pure store kernels reach on this box (the HLL first hop writes N x 256 B and reads almost nothing: its floor)
usage (GPU box): python tools/probe_store_rate.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch, bench
import subgraph_sketching_amd as ssa
dev = torch.device('cuda:0')
n = bench.N_NODES
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
x = torch.empty((n, 256), dtype=torch.uint8, device=dev)
y = torch.empty((n, 128), dtype=torch.int32, device=dev)
print('hll init (60 MB of stores)      ', timeit(lambda: eh._init_hll_u8(n, dev)))
print('minhash init (121 MB of stores) ', timeit(lambda: eh._init_minhash_u32(n, dev)))
print('torch fill 60 MB                ', timeit(lambda: x.fill_(1)))
print('torch fill 121 MB               ', timeit(lambda: y.fill_(1)))
