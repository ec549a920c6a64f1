import sys, time, os
sys.path.insert(0, '/root/repo')
from argparse import Namespace
import torch, bench
import subgraph_sketching_amd as ssa
dev = torch.device('cuda:0')
n, e_und, B = bench.N_NODES, bench.E_UND, bench.BATCH
ei = torch.from_numpy(bench.synthetic_graph(n, e_und)).to(dev)
links = torch.from_numpy(bench.synthetic_links(n, B, 2)).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
def step():
    t, c = eh.build_hash_tables(n, ei)
    return eh.get_subgraph_features(links, t, c)
for _ in range(5): step()
torch.cuda.synchronize()
for K in (20, 200):
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: host enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
