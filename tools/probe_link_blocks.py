"""probe (GPU box): what the rounds of dist.LinkRounds cost the query.  One rank's share of a BUDDY link set (L / 8 random pairs) is
computed (a) in one call -- the whole share grouped by first node at once, what gather='none' does -- and (b) block by block, each
block grouped on its own and stored straight into its slice of the output (out=), what gather='all' / 'rank0' do so that a round can
be gathered while the next one runs.  Random pairs are the worst case for (b) (a block holds a source far fewer times than the share);
evaluation-style lists (every source's negatives together) lose nothing.
usage: python tools/probe_link_blocks.py [--config citation2|ppa] [--json out.json]"""
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
ap.add_argument('--config', default='citation2')
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--blocks', type=int, nargs='*', default=[1 << 20, 1 << 21, 1 << 22, 5562500, 1 << 23, 11000000])
ap.add_argument('--json', default=None)
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = bench.CONFIGS[a.config]
n, h = cfg['n'], cfg['h']
eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
ei = torch.from_numpy(bench.synthetic_graph(n, cfg['e_und'])).to(dev)
table, cards = eh.build_hash_tables(n, ei)
share = cfg['buddy_links'] // a.world
g = torch.Generator(device=dev).manual_seed(2)
links = torch.randint(0, n, (share, 2), device=dev, generator=g)
nf = h * (h + 2)
out = torch.empty((share, nf), device=dev)
rows = []


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for style in ('random', 'eval1000'):
    lk = links
    if style == 'eval1000':
        lk = links.clone()
        lk[:, 0] = links[::1000, 0].repeat_interleave(1000)[:share]
    t = timed(lambda: eh.get_subgraph_features(lk, table, cards, out=out))
    rows.append({'links': style, 'block': 'whole share', 'ms': t * 1e3, 'Gpairs_per_s': share / t / 1e9})
    print(rows[-1], flush=True)
    ref = out.clone()
    for blk in a.blocks:
        def run():
            for s in range(0, share, blk):
                eh.get_subgraph_features(lk[s:s + blk], table, cards, out=out[s:s + blk])
        t = timed(run)
        assert torch.equal(out, ref)
        rows.append({'links': style, 'block': blk, 'ms': t * 1e3, 'Gpairs_per_s': share / t / 1e9})
        print(rows[-1], flush=True)
if a.json:
    json.dump({'config': a.config, 'share': share, 'world': a.world, 'rows': rows}, open(a.json, 'w'), indent=1)
