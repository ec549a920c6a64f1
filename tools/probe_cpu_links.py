"""probe: BUDDY's real call pattern -- CPU-resident links/cards in, CPU features out (datasets/elph.py:200-208)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import subgraph_sketching_amd as ssa
from argparse import Namespace
dev = torch.device('cuda:0')
n = bench.N_NODES
ei = torch.from_numpy(bench.synthetic_graph())           # CPU tensor, like HashDataset.edge_index
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
L = 2_662_400
links = torch.from_numpy(np.random.RandomState(0).randint(0, n, size=(L, 2)).astype(np.int64))
for rep in range(3):
    t0 = time.perf_counter()
    table, cards = eh.build_hash_tables(n, ei)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f = eh.get_subgraph_features(links, table, cards, 11_000_000)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'build (CPU edge_index in, cards out) {t1 - t0:.4f} s; features for {L} CPU links {t2 - t1:.4f} s = {L / (t2 - t1) / 1e6:.1f} M pairs/s; out {f.device} {tuple(f.shape)}', flush=True)
lg = links.to(dev); tg, cg = eh.build_hash_tables(n, ei.to(dev))
torch.cuda.synchronize(); t0 = time.perf_counter(); fg = eh.get_subgraph_features(lg, tg, cg); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'device-resident: {t1 - t0:.4f} s = {L / (t1 - t0) / 1e6:.1f} M pairs/s; equal: {torch.equal(fg.cpu(), f)}')
