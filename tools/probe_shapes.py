"""probe: whole step (build_hash_tables + one query batch) across the sketch parameterisations the reference exposes
(--minhash_num_perm, --hll_p, --max_hash_hops; reference defaults 128 / 8 / 2), on the collab-like graph.  Looks for
performance cliffs off the default shape: every row prints the step time, the bytes of the implemented schedule and
the fraction of the HBM peak they amount to.
usage (GPU box): python tools/probe_shapes.py [--json out.json]"""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import torch

import bench
import subgraph_sketching_amd as ssa

ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--only', default=None, help='P,p,h of a single shape (for a rocprofv3 --kernel-trace run)')
a = ap.parse_args()
dev = torch.device('cuda:0')
n, e_und, B = bench.N_NODES, bench.E_UND, bench.BATCH
ei = torch.from_numpy(bench.synthetic_graph(n, e_und)).to(dev)
links = torch.from_numpy(bench.synthetic_links(n, B, 2)).to(dev)
E = ei.size(1)
rows = []
SHAPES = [(128, 8, 2), (128, 8, 1), (128, 8, 3), (64, 8, 2), (192, 8, 2), (256, 8, 2), (128, 6, 2), (128, 10, 2), (128, 12, 2),
          (64, 6, 2), (256, 10, 3)]
for P, p, h in ([tuple(int(x) for x in a.only.split(','))] if a.only else SHAPES):
    eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=p, minhash_num_perm=P, floor_sf=False, use_zero_one=True))

    def step():
        tables, cards = eh.build_hash_tables(n, ei)
        return eh.get_subgraph_features(links, tables, cards)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    by = ssa.roofline.step_bytes_implemented(n, E, P, p, h, B)
    row = {'P': P, 'p': p, 'h': h, 'ms_per_step': ms, 'step_GB': by / 1e9, 'frac_of_hbm_peak': by / ms / 1e6 / ssa.roofline.HBM_PEAK_GBS}
    rows.append(row)
    print(f"P={P:3d} p={p:2d} h={h}: {ms:7.3f} ms/step  {row['step_GB']:6.2f} GB  {row['frac_of_hbm_peak']:.3f} of peak", flush=True)
if a.json:
    json.dump({'graph': f'collab-like N={n} E_und={e_und} B={B}', 'rows': rows}, open(a.json, 'w'), indent=1)
