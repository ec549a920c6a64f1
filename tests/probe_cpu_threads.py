import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import subgraph_sketching_amd as ssa
from oracle import oracle, torch_refstyle as tr
ei = bench.synthetic_graph(); links = bench.synthetic_links(seed=2)
t = ssa.hll_tables.load(8)
raw, bias = torch.tensor(t.raw_estimate, dtype=torch.float), torch.tensor(t.bias, dtype=torch.float)
mh0 = torch.from_numpy(oracle.minhash_init(bench.N_NODES, 128).astype(np.int64)); hll0 = torch.from_numpy(oracle.hll_init(bench.N_NODES, 8).view(np.int8))
tables = None
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    t0 = time.perf_counter()
    tables, cards = tr.build_tables(bench.N_NODES, torch.from_numpy(ei), 2, mh0, hll0, 8, t.alpha, t.threshold, raw, bias, hops_to_run=1)
    t1 = time.perf_counter()
    tables[2] = tables[1]
    tr.pair_intersections(torch.from_numpy(links), tables, 2, 128, 8, t.alpha, t.threshold, raw, bias)
    t2 = time.perf_counter()
    print(th, 'hop', round(t1 - t0, 2), 'query', round(t2 - t1, 3), flush=True)
