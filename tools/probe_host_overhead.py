"""probe: host (Python + launch) time per step against the GPU time of the step -- a step whose host side takes longer than its
kernels is host-bound whatever the kernels do.  Class defaults (deferred bounds reporting: no host round trip inside a step).
usage (GPU box): python tools/probe_host_overhead.py [build_query|elph] [batch]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import torch

import bench
import subgraph_sketching_amd as ssa

api = sys.argv[1] if len(sys.argv) > 1 else 'build_query'
dev = torch.device('cuda:0')
n, e_und = bench.N_NODES, bench.E_UND
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BATCH
h = 2
ei = torch.from_numpy(bench.synthetic_graph(n, e_und)).to(dev)
links = torch.from_numpy(bench.synthetic_links(n, B, 2)).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
state = {}


def step_build_query():
    t, c = eh.build_hash_tables(n, ei)
    return eh.get_subgraph_features(links, t, c)


def step_elph():  # reference models/elph.py:186-213 + runners/train.py:204
    loops = torch.arange(n, device=dev).repeat(2, 1)
    hei = torch.cat([ei, loops], dim=1)
    if 'mh0' not in state:
        state['mh0'], state['hll0'] = eh.initialise_minhash(n), eh.initialise_hll(n)
    table = {0: {'minhash': state['mh0'], 'hll': state['hll0']}}
    cards = torch.zeros((n, h), device=dev)
    for k in range(1, h + 1):
        table[k] = {'hll': eh.hll_prop(table[k - 1]['hll'], hei), 'minhash': eh.minhash_prop(table[k - 1]['minhash'], hei)}
        cards[:, k - 1] = eh.hll_count(table[k]['hll'])
    return eh.get_subgraph_features(links, table, cards)


step = {'build_query': step_build_query, 'elph': step_elph}[api]
for _ in range(5):
    step()
torch.cuda.synchronize()
for K in (20, 200):
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{api} B={B} K={K}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, total {1e3 * (t2 - t0) / K:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
