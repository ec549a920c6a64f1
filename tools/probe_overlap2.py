"""probe: the fused hop stage's kernel (MinHash first hop + HLL table hop interleaved inside every wavefront) against the same two
kernels launched on two streams (co-resident wavefronts of both kinds on every CU) and back to back.
usage (GPU box): python tools/probe_overlap2.py [collab|ppa|citation2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
from ctypes import byref
import torch
import bench
import subgraph_sketching_amd as ssa
from subgraph_sketching_amd import _native
from subgraph_sketching_amd.hashing import build_csr, _propagate, _ptr, _stream
dev = torch.device('cuda:0')
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'collab']
n, e_und = cfg['n'], cfg['e_und']
ei = torch.from_numpy(bench.synthetic_graph(n, e_und)).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
csr = build_csr(ei, n, dev, check=False); csr.use_inferred_self_loops = True
csr.has_hub_rows = False  # (uniform graph: no hub passes in any variant)
prm = eh._params(dev)
mh = torch.empty((n, 128), dtype=torch.int32, device=dev); hll = torch.empty((n, 256), dtype=torch.uint8, device=dev)
hll2 = torch.empty_like(hll); hll2b = torch.empty_like(hll); mhb = torch.empty_like(mh)
cards = torch.empty((n, 2), device=dev); cardsb = torch.empty((n, 2), device=dev)
eh._first_hop(csr, dev, None, hll, cards, prm)  # hop-1 HLL table: input of all variants
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
ab = eh._perms(dev)
graph = csr.struct()
lib = _native.lib()


def fused():
    _native.check(lib.ss_fused_hop_stage(byref(graph), _ptr(ab[0]), _ptr(ab[1]), 128, _ptr(mh), None, 8, _ptr(hll), None, _ptr(hll2),
                                         _ptr(cards[:, 1]), 2, byref(prm.struct), _stream(dev)), 'fused')


def mh_first():
    eh._first_hop(csr, dev, mhb, None, None, prm)


def hll_hop():
    _propagate(csr, None, hll, dev, cards_out=cardsb[:, 1], cards_stride=2, params=prm, hll_out=hll2b)


def back_to_back():
    mh_first(); hll_hop()


def two_streams():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        mh_first()
    hll_hop()
    main.wait_stream(side)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, fn in (('fused kernel', fused), ('mh first hop alone', mh_first), ('hll table hop alone', hll_hop), ('back to back', back_to_back),
                 ('two streams', two_streams), ('fused kernel', fused)):
    print(f'{name:28s} {timeit(fn):9.1f} us', flush=True)
torch.cuda.synchronize()
print('identical:', bool(torch.equal(mh, mhb) and torch.equal(hll2, hll2b) and torch.equal(cards[:, 1], cardsb[:, 1])))
