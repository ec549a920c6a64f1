#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline_reference_style`: the REAL reference
(/root/reference/src/hashing.py, imported under the sys.modules stand-ins of tests/golden/make_golden.py) timed against
oracle/torch_refstyle.py on the same workload, same thread count, in the BUILD container (the reference never travels to
the GPU box).  Workload: the bench graph (N = 235 868, E_und = 1 179 052, seed 1), h = 2, B = 65 536 pairs (seed 2).
Writes profiles/round2_cpu_baseline_calibration.json.  The stand-in for PyG's propagate is torch.scatter_reduce('amax')
over the materialised messages -- the same operator torch_refstyle uses -- so the build ratio mostly checks the bookkeeping
around it; the query path is the reference's own code end to end.
usage: python tools/calibrate_cpu_baseline.py [--threads 8]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

ap = argparse.ArgumentParser()
ap.add_argument('--threads', type=int, default=os.cpu_count())
ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'round2_cpu_baseline_calibration.json'))
a = ap.parse_args()
torch.set_num_threads(a.threads)

import bench  # noqa: E402
import make_golden  # noqa: E402
import subgraph_sketching_amd as ssa  # noqa: E402
from oracle import oracle, torch_refstyle as tr  # noqa: E402

ref, _ = make_golden.load_reference(False)
n, h, B = bench.N_NODES, bench.H, bench.BATCH
ei = torch.from_numpy(bench.synthetic_graph())
links = torch.from_numpy(bench.synthetic_links())
args = make_golden.args(h=h)
res = {'threads': a.threads, 'cores': os.cpu_count(), 'workload': f'N={n}, E_dir={ei.shape[1]}, h={h}, B={B}', 'torch': torch.__version__}

t = ssa.hll_tables.load(8, prefer='regenerated')
raw, bias = torch.tensor(t.raw_estimate, dtype=torch.float), torch.tensor(t.bias, dtype=torch.float)
eh = ref.ElphHashes(args)
for rep in range(2):  # the second repetition is the record: the first one pays first-touch page faults for ~3 GB of messages
    t0 = time.perf_counter()
    tables, cards = eh.build_hash_tables(n, ei)
    res['reference_build_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    feats = eh.get_subgraph_features(links, tables, cards)
    res['reference_query_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mh0, hll0 = tr.init_sketches(n, 128, 8)
    tables2, cards2 = tr.build_tables(n, ei, h, mh0, hll0, 8, t.alpha, t.threshold, raw, bias)
    res['refstyle_build_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    inter = tr.pair_intersections(links, tables2, h, 128, 8, t.alpha, t.threshold, raw, bias)
    res['refstyle_query_s'] = time.perf_counter() - t0
assert torch.equal(tables[0]['minhash'], mh0) and torch.equal(tables[0]['hll'], hll0)
assert torch.equal(tables[2]['minhash'], tables2[2]['minhash']) and torch.equal(tables[2]['hll'], tables2[2]['hll'])
ref_inter = eh._get_intersections(links, tables)
res['max_abs_intersection_diff'] = max(float((ref_inter[k] - inter[k]).abs().max()) for k in inter)
res['build_ratio_refstyle_over_reference'] = res['refstyle_build_s'] / res['reference_build_s']
res['query_ratio_refstyle_over_reference'] = res['refstyle_query_s'] / res['reference_query_s']
res['step_ratio_refstyle_over_reference'] = ((res['refstyle_build_s'] + res['refstyle_query_s']) /
                                             (res['reference_build_s'] + res['reference_query_s']))
res['reference_pairs_per_s_one_step'] = B / (res['reference_build_s'] + res['reference_query_s'])
res['note'] = ('refstyle_query covers the h^2 intersections only (the reference additionally gathers cards and assembles the '
               'features: a few ms); within +-10 % => the torch-style figure bench.py prints on the GPU box stands for the reference')
json.dump(res, open(a.out, 'w'), indent=1)
print(json.dumps(res, indent=1))
