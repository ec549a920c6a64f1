"""probe: is a whole build+query step capturable into a HIP graph through torch.cuda.graph? replay time vs eager"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import subgraph_sketching_amd as ssa
from argparse import Namespace
dev = torch.device('cuda:0')
for n, e_und, B in ((2485, 3550, 1024), (235868, 1179052, 65536)):
    rng = np.random.RandomState(1)
    e = rng.randint(0, n, size=(2, e_und)).astype(np.int64)
    ei = torch.from_numpy(np.concatenate([e, e[::-1]], 1)).to(dev)
    links = torch.from_numpy(rng.randint(0, n, size=(B, 2)).astype(np.int64)).to(dev)
    eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
    eh.strict_bounds = False
    def step():
        t, c = eh.build_hash_tables(n, ei)
        return eh.get_subgraph_features(links, t, c)
    ref = step(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay(); torch.cuda.synchronize()
    print('equal after replay:', torch.equal(out, ref))
    links.copy_(links.flip(1)); g.replay(); torch.cuda.synchronize()
    ref2 = step(); torch.cuda.synchronize()
    print('equal after input change:', torch.equal(out, ref2))
    for name, fn in (('eager', step), ('graph', g.replay)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); print(n, name, round((time.perf_counter() - t0) / 50 * 1e6, 1), 'us/step')
