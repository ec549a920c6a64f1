"""probe (GPU box): why is the h = 3 query on citation2-size tables slower inside a build + query step than alone?
VERDICT r5 weak #6: tracked profile 237.7-239.2 us in-step, stand-alone probe 223.5 us (synthetic tables).  Separates the two suspects:
the TABLE CONTENTS (tools/probe_pairs.py fills its tables at random; the estimator's branches depend on the registers) and the
LAUNCH BEFORE it (the hop-3 MinHash launch has just written 1.5 GB: dirty lines, TLB, clocks).  Every figure = mean HIP-event span
of the query launches recorded inside the library (ss_profile_*), B = 261 424 random pairs.
usage: python tools/probe_query_instep.py [--json out.json]"""
import argparse
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from ctypes import byref, c_float, c_int32, c_void_p

import torch

import bench
import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _propagate, _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='citation2')
ap.add_argument('--json', default=None)
ap.add_argument('--reps', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = bench.CONFIGS[a.config]
n, h, B = cfg['n'], cfg['h'], cfg['batch']
nat = ssa._native
lib = nat.lib()
eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
ei = torch.from_numpy(bench.synthetic_graph(n, cfg['e_und'])).to(dev)
links = torch.from_numpy(bench.synthetic_links(n, B)).to(dev)
table, cards = eh.build_hash_tables(n, ei)
csr = ssa.build_csr(ei, n, dev, check=False)
csr.use_inferred_self_loops = True
csr.has_hub_rows = False
mh = [table[k].mh_u32 for k in range(1, h + 1)]
hl = [table[k].hll_u8 for k in range(1, h + 1)]
prm = eh._params(dev)
res = {}


def span(name, body, reps=a.reps):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    lib.ss_profile_enable(1 << nat.PROF_PAIRS)
    for _ in range(reps):
        body()
    torch.cuda.synchronize()
    ms, cnt = c_float(), c_int32()
    lib.ss_profile_read(nat.PROF_PAIRS, byref(ms), byref(cnt))
    lib.ss_profile_enable(0)
    res[name] = {'us': ms.value * 1e3, 'launches': cnt.value}
    print(f'{name:58s} {ms.value * 1e3:8.1f} us  ({cnt.value} launches)', flush=True)


query = lambda: eh.get_subgraph_features(links, table, cards)
span('query alone, real tables, back to back', query)


def spaced():
    query()
    torch.cuda._sleep(200000)  # ~100 us of idle stream between two queries
span('query alone, real tables, ~100 us idle before each', spaced)


def after_hop(gap=0):
    _propagate(csr, mh[h - 2], None, dev, mh_out=mh[h - 1])  # the launch that precedes the query in a step (rewrites the same rows)
    if gap:
        torch.cuda._sleep(gap)
    query()
span('query right after the last MinHash table hop', after_hop)
span('... with ~50 us idle between hop and query', lambda: after_hop(100000))
span('... with ~500 us idle between hop and query', lambda: after_hop(1000000))
scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def after_fill():
    scratch.fill_(1)  # 1 GiB of plain stores (dirty lines that are not the tables')
    query()
span('query right after a 1 GiB fill of another buffer', after_fill)


def step():
    t, c = eh.build_hash_tables(n, ei)
    eh.get_subgraph_features(links, t, c)
span('query inside the bench step (build + query)', step, reps=10)



def step_then(gap_cycles=0, sync=False, twice=False):
    def body():
        t, c = eh.build_hash_tables(n, ei)
        if sync:
            torch.cuda.synchronize()
        if gap_cycles:
            torch.cuda._sleep(gap_cycles)
        eh.get_subgraph_features(links, t, c)
        if twice:
            lib.ss_profile_enable(0)  # (only the SECOND query of the step is timed)
            lib.ss_profile_enable(1 << nat.PROF_PAIRS)
    return body


def second_query():
    t, c = eh.build_hash_tables(n, ei)
    lib.ss_profile_enable(0)
    eh.get_subgraph_features(links, t, c)      # untimed: warms whatever the build left cold
    lib.ss_profile_enable(1 << nat.PROF_PAIRS)
    eh.get_subgraph_features(links, t, c)
    lib.ss_profile_enable(0)
    lib.ss_profile_enable(1 << nat.PROF_PAIRS)
span('step: build, host synchronise, query', step_then(sync=True), reps=10)
span('step: build, ~1 ms idle stream, query', step_then(gap_cycles=2000000), reps=10)
span('step: build, query (untimed), query (timed)', second_query, reps=10)
fixed_t, fixed_c = eh.build_hash_tables(n, ei)


def other_traffic():
    # 19 ms of somebody else's traffic over OTHER memory (a build whose tables are thrown away), then the query on tables that have
    # not been written since: separates "the tables were just written" from "the GPU was just busy elsewhere"
    eh.build_hash_tables(n, ei)
    eh.get_subgraph_features(links, fixed_t, fixed_c)
span('another build (other memory), then query on OLD tables', other_traffic, reps=10)

# synthetic tables as tools/probe_pairs.py makes them
g = torch.Generator(device=dev).manual_seed(1)
smh = [torch.randint(-2**31, 2**31 - 1, (n, 128), dtype=torch.int32, device=dev, generator=g) for _ in range(h)]
shl = [torch.clamp((-torch.log2(torch.rand((n, 256), device=dev, generator=g))).ceil() * (torch.rand((n, 256), device=dev, generator=g) < 0.6), 0, 50).to(torch.uint8) for _ in range(h)]
stab = {k: ssa.HopSketch(smh[k - 1], shl[k - 1], dev) for k in range(1, h + 1)}
scards = torch.rand((n, h), device=dev, generator=g) * 300
span('query alone, SYNTHETIC tables (tools/probe_pairs.py)', lambda: eh.get_subgraph_features(links, stab, scards))

# the 4-waves-per-SIMD build of the same kernel (128 VGPRs at h = 3; what the grouped walks use) on the as-listed random pairs
mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh])
hl_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hl])
out = torch.empty((B, h * (h + 2)), device=dev)


def capped():
    rc = lib.ss_pair_features_grouped_kernel(0, _ptr(links), None, B, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, None,
                                             _ptr(out), None, _stream(dev))
    assert rc == 0
span('query alone, real tables, 4 waves / SIMD build of the kernel', capped)
assert torch.equal(out, eh.get_subgraph_features(links, table, cards))
if a.json:
    json.dump({'config': a.config, 'B': B, 'rows': res}, open(a.json, 'w'), indent=1)
