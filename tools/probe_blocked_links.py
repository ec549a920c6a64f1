"""probe (GPU box): the BUDDY query over a link set ordered by (block of the SECOND node, first node) instead of by first node alone.
Grouping by first node makes u's rows a cache hit and leaves v's rows as 120 random HBM gathers per node (ogbl-citation2: L / N = 120);
with the second nodes confined to a window of the tables that fits the Infinity Cache, v's rows come from there and u's rows stream
through in ascending order once per window: HBM traffic = windows x table bytes instead of L x 2.3 KB.
usage: python tools/probe_blocked_links.py [--config citation2] [--links 44500000] [--windows 8 16 32 64]"""
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
from subgraph_sketching_amd import engine as eng
from subgraph_sketching_amd.hashing import _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='citation2')
ap.add_argument('--links', type=int, default=44_500_000)
ap.add_argument('--windows', type=int, nargs='*', default=[8, 16, 32, 64])
ap.add_argument('--json', default=None)
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = bench.CONFIGS[a.config]
n, h = cfg['n'], cfg['h']
lib = ssa._native.lib()
eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
ei = torch.from_numpy(bench.synthetic_graph(n, cfg['e_und'])).to(dev)
table, cards = eh.build_hash_tables(n, ei)
del ei
L = a.links
links_rnd = bench.device_links(n, L, dev)
links_eval = links_rnd.clone()
links_eval[:, 0] = links_rnd[::1000, 0].repeat_interleave(1000)[:L]
plain_group = eng.group_links_by_source
state = {'windows': 0}


def blocked_group(links, num_nodes, device=None):
    w = state['windows']
    if not w:
        return plain_group(links, num_nodes, device)
    blk = (num_nodes + w - 1) // w
    ids = torch.div(links[:, 1], blk, rounding_mode='floor') * num_nodes + links[:, 0]
    num_ids = w * num_nodes
    E = ids.numel()
    rowptr = torch.empty(num_ids + 1, dtype=torch.int64, device=links.device)
    order = torch.empty(E, dtype=torch.int32, device=links.device)
    err = torch.zeros(1, dtype=torch.int32, device=links.device)
    ws_bytes = lib.ss_csr_workspace_bytes(num_ids, E)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=links.device)
    rc = lib.ss_csr_group_ids(_ptr(ids), E, num_ids, _ptr(order), _ptr(rowptr), _ptr(err), _ptr(ws), ws_bytes, _stream(links.device))
    assert rc == 0, rc
    return order


eng.group_links_by_source = blocked_group
rows = []


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = torch.empty((L, h * (h + 2)), device=dev)
for name, links in (('random', links_rnd), ('eval1000', links_eval)):
    ref = None
    for mode in ('as_listed', 0) + tuple(a.windows):
        if mode == 'as_listed':
            eh.group_links = False
            state['windows'] = 0
        else:
            eh.group_links = True
            state['windows'] = mode
        t = timed(lambda: eh.get_subgraph_features(links, table, cards, out=out))
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        # the grouping alone
        tg = 0.0 if mode == 'as_listed' else timed(lambda: blocked_group(links, n, dev))
        row = {'links': name, 'L': L, 'order': 'as listed' if mode == 'as_listed' else ('by first node' if mode == 0 else f'{mode} windows of the second node, then first node'),
               'ms': t * 1e3, 'grouping_ms': tg * 1e3, 'Gpairs_per_s': L / t / 1e9, 'bit_identical': same}
        rows.append(row)
        print(json.dumps(row), flush=True)
eng.group_links_by_source = plain_group
if a.json:
    json.dump({'config': a.config, 'rows': rows}, open(a.json, 'w'), indent=1)
