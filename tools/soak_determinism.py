"""soak: the same build + query step over and over, every result compared with the first one on the device (torch.equal of all
hop tables, cards and features) -- looks for rare races (cross-workgroup hand-offs, LDS landings of the fused stage, the
deferred bounds word) that a single parity run would not hit.  Graph shapes alternate between the uniform bench graph and a
skewed one with hub and mega rows.
usage (GPU box): python tools/soak_determinism.py [seconds per shape = 60]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import torch

import bench
import subgraph_sketching_amd as ssa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device('cuda:0')
n, e_und, B = bench.N_NODES, bench.E_UND, bench.BATCH
bad = 0
for kind, alpha, h in (('uniform', 0.5, 2), ('powerlaw', 0.9, 2), ('uniform', 0.5, 3), ('powerlaw', 0.5, 2), ('powerlaw', 0.9, 3)):
    ei = torch.from_numpy(bench.synthetic_graph(n, e_und, kind, alpha)).to(dev)
    links = torch.from_numpy(bench.synthetic_links(n, B, 2)).to(dev)
    eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))

    def step():
        table, cards = eh.build_hash_tables(n, ei)
        return [table[k].mh_u32 for k in range(1, h + 1)] + [table[k].hll_u8 for k in range(1, h + 1)] + [cards, eh.get_subgraph_features(links, table, cards)]

    ref = step()
    t0, steps, wrong = time.time(), 0, 0
    while time.time() - t0 < seconds:
        flags = []
        for _ in range(50):
            out = step()
            flags.append(torch.stack([torch.equal(a, b) if False else (a == b).all() for a, b in zip(out, ref)]).all())
            steps += 1
        wrong += 50 - int(torch.stack(flags).sum())
    eh.check_errors()
    bad += wrong
    print(f'{kind} alpha={alpha} h={h}: {steps} steps, {wrong} differing from the first', flush=True)
sys.exit(1 if bad else 0)
