"""probe: ss_pair_features kernel time against the batch size (ELPH batches of 1-2 k pairs up to BUDDY chunks of millions)
and the hop count.  HIP events on the launch stream around 20 back-to-back launches (ss_time_pair_features).
usage (GPU box): python tools/probe_pairs.py [--json out.json] [--nodes N]
Tuning hooks read once per process by the library: SS_PAIR_GRID (max workgroups), SS_PAIR_PER_GROUP (pairs per 16-lane group
the grid is sized for)."""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from ctypes import byref, c_float, c_void_p

import torch

import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--nodes', type=int, default=235868)
ap.add_argument('--hops', type=int, nargs='*', default=[2, 3])
ap.add_argument('--batches', type=int, nargs='*', default=[1024, 2048, 8192, 65536, 262144, 1048576, 4194304])
ap.add_argument('--sorted', action='store_true', help='links ordered by source node (a coalesced edge list: consecutive pairs share u)')
a = ap.parse_args()
dev = torch.device('cuda:0')
n = a.nodes
lib = ssa._native.lib()
rows = []
for h in a.hops:
    eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
    prm = eh._params(dev)
    mh = [torch.randint(-2**31, 2**31 - 1, (n, 128), dtype=torch.int32, device=dev) for _ in range(h)]
    # realistic registers: geometric ranks
    hl = [torch.clamp((-torch.log2(torch.rand((n, 256), device=dev))).ceil() * (torch.rand((n, 256), device=dev) < 0.6), 0, 50).to(torch.uint8) for _ in range(h)]
    cards = torch.rand((n, h), device=dev) * 300
    mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh]); hl_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hl])
    for B in a.batches:
        links = torch.randint(0, n, (B, 2), device=dev)
        if a.sorted:
            links = links[torch.argsort(links[:, 0])].contiguous()
        out = torch.empty((B, h * (h + 2)), device=dev)
        ms = c_float()
        rc = lib.ss_time_pair_features(_ptr(links), B, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, _ptr(out), _stream(dev), 20, byref(ms))
        assert rc == 0
        bytes_ = B * ssa.roofline.pair_bytes(128, 8, h)
        row = {'h': h, 'B': B, 'us': ms.value * 1e3, 'GBps': bytes_ / ms.value / 1e6, 'frac_of_hbm_peak': bytes_ / ms.value / 1e6 / ssa.roofline.HBM_PEAK_GBS,
               'Mpairs_per_s': B / ms.value / 1e3}
        rows.append(row)
        print(f"h={h} B={B:8d}: {row['us']:8.1f} us  {row['GBps']:6.0f} GB/s  {row['frac_of_hbm_peak']:.3f} of peak  {row['Mpairs_per_s']:7.0f} Mpairs/s", flush=True)
if a.json:
    json.dump({'nodes': n, 'note': 'ss_time_pair_features: 20 back-to-back launches, HIP events on the launch stream; synthetic tables', 'env': {k: v for k, v in os.environ.items() if k.startswith('SS_PAIR')}, 'rows': rows},
              open(a.json, 'w'), indent=1)
