"""probe: does the VALU-bound MinHash first hop overlap with the memory-bound HLL chain (HLL first hop -> HLL table hop) when
the two run on different streams?  (DESIGN 3.4 records the round-1 attempt: no gain.)  Prints sequential vs concurrent time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
import bench
import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import build_csr, _propagate
dev = torch.device('cuda:0')
n = bench.N_NODES
ei = torch.from_numpy(bench.synthetic_graph()).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
csr = build_csr(ei, n, dev); csr.use_inferred_self_loops = True
prm = eh._params(dev)
mh = torch.empty((n, 128), dtype=torch.int32, device=dev); hll = torch.empty((n, 256), dtype=torch.uint8, device=dev)
hll2 = torch.empty_like(hll); mh2 = torch.empty_like(mh)
cards = torch.empty((n, 2), device=dev)
side = torch.cuda.Stream(device=dev, priority=int(os.environ.get('SIDE_PRIORITY', '0')))
main = torch.cuda.current_stream(dev)


def hll_chain():
    eh._first_hop(csr, dev, None, hll, cards, prm)
    _propagate(csr, None, hll, dev, cards_out=cards[:, 1], cards_stride=2, params=prm, hll_out=hll2)


def mh_first():
    eh._first_hop(csr, dev, mh, None, None, prm)


def mh_hop():
    _propagate(csr, mh, None, dev, mh_out=mh2)


def sequential():
    hll_chain(); mh_first(); mh_hop()


def concurrent():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        mh_first()
    hll_chain()
    main.wait_stream(side)
    mh_hop()


def concurrent_all():  # both chains fully apart
    side.wait_stream(main)
    with torch.cuda.stream(side):
        mh_first(); mh_hop()
    hll_chain()
    main.wait_stream(side)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, fn in (('hll chain alone', hll_chain), ('mh first hop alone', mh_first), ('mh table hop alone', mh_hop), ('sequential', sequential),
                 ('mh first hop || hll chain, then mh hop', concurrent), ('mh chain || hll chain', concurrent_all)):
    print(f'{name:45s} {timeit(fn):8.1f} us', flush=True)
