"""probe: the query over link sets with and without locality -- random pairs, pairs ordered by source, an evaluation-style set
(`--negs` negatives per source listed together) -- through ss_pair_features (every pair reads its 2h rows), ss_pair_features_grouped
without an order (runs as listed) and with the order of ss_group_links_by_source (grouping time included).  Rows are compared bit for bit.
usage (GPU box): python tools/probe_pair_runs.py [--nodes N] [--hops 3] [--links L] [--json out.json]
hooks: SS_PAIR_RUN_CAP=0 (uncapped registers), SS_PAIR_RUN_CHUNK=k (pairs per lane-group chunk)"""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from ctypes import byref, c_void_p

import torch

import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--nodes', type=int, default=2927963)
ap.add_argument('--hops', type=int, default=3)
ap.add_argument('--links', type=int, default=4194304)
ap.add_argument('--sources', type=int, default=0, help='random links drawn with first nodes from this many distinct nodes (0 = all): L / sources pairs per source, e.g. 120 for ogbl-citation2')
ap.add_argument('--negs', type=int, default=1000)
ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
dev = torch.device('cuda:0')
n, h, L = a.nodes, a.hops, a.links
lib = ssa._native.lib()
eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
prm = eh._params(dev)
mh = [torch.randint(-2**31, 2**31 - 1, (n, 128), dtype=torch.int32, device=dev) for _ in range(h)]
hl = [torch.clamp((-torch.log2(torch.rand((n, 256), device=dev))).ceil() * (torch.rand((n, 256), device=dev) < 0.6), 0, 50).to(torch.uint8) for _ in range(h)]
cards = torch.rand((n, h), device=dev) * 300
mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh]); hl_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hl])
nf = h * (h + 2)
ws_bytes = lib.ss_csr_workspace_bytes(n, L)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
order = torch.empty(L, dtype=torch.int32, device=dev)
rowptr = torch.empty(n + 1, dtype=torch.int64, device=dev)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


rows = []
g = torch.Generator(device=dev).manual_seed(5)
rnd = torch.randint(0, n, (L, 2), device=dev, generator=g)
if a.sources:
    rnd[:, 0] = torch.randint(0, a.sources, (L,), device=dev, generator=g) * (n // a.sources)
sets = {'random': rnd, 'sorted_by_source': rnd[torch.argsort(rnd[:, 0])].contiguous()}
src = torch.randint(0, n, (L // a.negs + 1,), device=dev, generator=g).repeat_interleave(a.negs)[:L]
sets[f'eval_{a.negs}_negatives_per_source'] = torch.stack([src, rnd[:, 1]], dim=1).contiguous()
for name, links in sets.items():
    out0 = torch.empty((L, nf), device=dev); out1 = torch.empty_like(out0); out2 = torch.empty_like(out0)

    def plain():
        assert lib.ss_pair_features(_ptr(links), L, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, _ptr(out0), None, None, None, None, _stream(dev)) == 0

    def group():
        assert lib.ss_group_links_by_source(_ptr(links), L, n, _ptr(order), _ptr(rowptr), _ptr(ws), ws_bytes, _stream(dev)) == 0

    def grouped(which):
        group()
        assert lib.ss_pair_features_grouped_kernel(which, _ptr(links), _ptr(order), L, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, None, _ptr(out2), None, _stream(dev)) == 0

    def runs():
        assert lib.ss_pair_features_grouped_kernel(1, _ptr(links), None, L, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, None, _ptr(out1), None, _stream(dev)) == 0

    def plain_capped():  # the ordinary kernel under its capped register budget (what ss_pair_features_grouped launches) on the pairs as listed
        assert lib.ss_pair_features_grouped_kernel(0, _ptr(links), None, L, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, None, _ptr(out2), None, _stream(dev)) == 0
    t_plain, t_runs, t_group = timed(plain), timed(runs), timed(group)
    t_plain_capped = timed(plain_capped)
    same3 = bool(torch.equal(out0, out2))
    t_grouped_plain = timed(lambda: grouped(0))
    same2 = bool(torch.equal(out0, out2))
    t_grouped_runs = timed(lambda: grouped(1))
    same1, same2 = bool(torch.equal(out0, out1)), same2 and bool(torch.equal(out0, out2))
    row = {'links': name, 'L': L, 'h': h, 'nodes': n, 'plain_ms': t_plain, 'runs_as_listed_ms': t_runs, 'grouping_ms': t_group,
           'grouped_plain_kernel_total_ms': t_grouped_plain, 'grouped_runs_kernel_total_ms': t_grouped_runs,
           'plain_Mpairs_s': L / t_plain / 1e3, 'runs_Mpairs_s': L / t_runs / 1e3,
           'grouped_plain_Mpairs_s': L / t_grouped_plain / 1e3, 'grouped_runs_Mpairs_s': L / t_grouped_runs / 1e3,
           'plain_capped_as_listed_Mpairs_s': L / t_plain_capped / 1e3, 'bit_identical': same1 and same2 and same3}
    rows.append(row)
    print(json.dumps(row), flush=True)
    assert same1 and same2
if a.json:
    json.dump({'env': {k: v for k, v in os.environ.items() if k.startswith('SS_PAIR')}, 'rows': rows}, open(a.json, 'w'), indent=1)
