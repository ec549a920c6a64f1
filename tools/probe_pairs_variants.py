"""probe (GPU box): the query kernel's two register budgets (default: 117 / 151 VGPRs at h = 2 / 3 = four / three wavefronts per SIMD;
capped: 96 / 128 = five / four, what the grouped walks use) on as-listed random pairs, against table size, batch size and -- new --
how long the kernel has been running: a cold GPU boosts, a query loop of a BUDDY precompute runs at sustained clocks.
Every figure: mean HIP-event span of the launches recorded inside the library (ss_profile_*).
usage: python tools/probe_pairs_variants.py [--json out.json]"""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from ctypes import byref, c_float, c_int32, c_void_p

import torch

import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--hops', type=int, nargs='*', default=[3, 2])
ap.add_argument('--nodes', type=int, nargs='*', default=[100000, 235868, 576289, 1200000, 2927963])
ap.add_argument('--batches', type=int, nargs='*', default=[65536, 261424, 4194304])
ap.add_argument('--sustain', type=int, nargs='*', default=[0, 2000])
a = ap.parse_args()
dev = torch.device('cuda:0')
nat = ssa._native
lib = nat.lib()
rows = []
g = torch.Generator(device=dev).manual_seed(1)
for h in a.hops:
    eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
    prm = eh._params(dev)
    for n in a.nodes:
        mh = [torch.randint(-2**31, 2**31 - 1, (n, 128), dtype=torch.int32, device=dev, generator=g) for _ in range(h)]
        hl = [torch.clamp((-torch.log2(torch.rand((n, 256), device=dev, generator=g))).ceil() * (torch.rand((n, 256), device=dev, generator=g) < 0.6), 0, 50).to(torch.uint8) for _ in range(h)]
        cards = torch.rand((n, h), device=dev, generator=g) * 300
        mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh])
        hl_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hl])
        table_mb = h * n * 768 / 2**20
        for B in a.batches:
            links = torch.randint(0, n, (B, 2), device=dev, generator=g)
            out = torch.empty((B, h * (h + 2)), device=dev)

            def launch(capped):
                if capped:
                    rc = lib.ss_pair_features_grouped_kernel(0, _ptr(links), None, B, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, None,
                                                             _ptr(out), None, _stream(dev))
                else:
                    rc = lib.ss_pair_features(_ptr(links), B, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, _ptr(out), None, None, None, None,
                                              _stream(dev))
                assert rc == 0
            for warm in a.sustain:
                res = {}
                for capped in (False, True):
                    torch.cuda.synchronize()
                    if warm == 0:
                        torch.cuda._sleep(20000000)  # ~10 ms idle: the launches below meet a rested GPU
                    for _ in range(max(warm, 3)):
                        launch(capped)
                    lib.ss_profile_enable(1 << nat.PROF_PAIRS)
                    for _ in range(20):
                        launch(capped)
                    torch.cuda.synchronize()
                    ms, cnt = c_float(), c_int32()
                    lib.ss_profile_read(nat.PROF_PAIRS, byref(ms), byref(cnt))
                    lib.ss_profile_enable(0)
                    res['capped' if capped else 'default'] = ms.value * 1e3
                row = {'h': h, 'nodes': n, 'table_MiB': round(table_mb), 'B': B, 'warm_launches': warm, 'default_us': res['default'], 'capped_us': res['capped'],
                       'capped_over_default': res['capped'] / res['default']}
                rows.append(row)
                print(row, flush=True)
        del mh, hl, cards
        torch.cuda.empty_cache()
if a.json:
    json.dump({'rows': rows}, open(a.json, 'w'), indent=1)
