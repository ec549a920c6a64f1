"""probe: what one BUDDY training epoch pays for its structural features
   (a) the reference's hand-off: [L, F] tensor on the host, per batch `features[sf_indices].to(device)` (runners/train.py:58-60)
       -- host gather + PCIe copy; pinned and pageable variants;
   (b) DeviceFeatureStore: per batch the rows are recomputed on the GPU from the resident sketch tables.
usage (GPU box): python tools/probe_feature_store.py [--json out.json]"""
import argparse
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import numpy as np
import torch

import bench
import subgraph_sketching_amd as ssa

ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--links', type=int, default=2_662_400)
ap.add_argument('--only', default=None, help='time one variant only (host_tensor_pageable | host_tensor_pinned_source | device_feature_store)')
a = ap.parse_args()
dev = torch.device('cuda:0')
n = bench.N_NODES
ei = torch.from_numpy(bench.synthetic_graph()).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
L = a.links
links = torch.from_numpy(np.random.RandomState(0).randint(0, n, size=(L, 2)).astype(np.int64))
table, cards = eh.build_hash_tables(n, ei)
host = eh.get_subgraph_features(links.to(dev), table, cards).cpu()           # the reference's host-resident [L, 8] tensor
pinned = host.pin_memory()
store = eh.get_subgraph_features(links, table, cards, lazy=True)
rows = []
for batch in (1024, 65536, 1048576):
    perm = torch.randperm(L, generator=torch.Generator().manual_seed(0))
    batches = [perm[s:s + batch] for s in range(0, L - batch + 1, batch)][:200]
    res = {'batch': batch, 'batches_timed': len(batches)}
    # (the store first: after the two host-tensor loops -- hundreds of MB of freshly gathered pageable tensors -- the pageable upload
    # of ITS index tensor has been seen 20x slower; each variant is meant to be timed on a quiet allocator)
    for name, fn in (('device_feature_store', lambda idx: store[idx]),
                     ('host_tensor_pageable', lambda idx: host[idx].to(dev)),
                     ('host_tensor_pinned_source', lambda idx: pinned[idx].to(dev, non_blocking=True))):
        if a.only and name != a.only:
            continue
        fn(batches[0]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for idx in batches:
            out = fn(idx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name + '_Mpairs_per_s'] = len(batches) * batch / dt / 1e6
        res[name + '_us_per_batch'] = dt / len(batches) * 1e6
    assert torch.equal(store[batches[0]].cpu(), host[batches[0]])
    rows.append(res)
    print(res, flush=True)
if a.json:
    json.dump({'links': L, 'graph': 'collab-like (bench.py)', 'cores': os.cpu_count(), 'rows': rows,
               'note': 'indices are CPU tensors as the reference draws them (DataLoader(range(L))); the store uploads them (8 B per pair) '
                       'and gathers the link ids on the device'}, open(a.json, 'w'), indent=1)
