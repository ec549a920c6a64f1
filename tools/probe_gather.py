"""probe: achieved bandwidth of the propagation kernel per sketch (512 B MinHash rows vs 256 B HLL rows) on a
large random graph -- is the random-row gather limited by chunk size?   (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import byref, c_float
import numpy as np, torch
import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _ptr, _stream, build_csr
from argparse import Namespace

dev = torch.device('cuda:0')
n, e_und = int(sys.argv[1]) if len(sys.argv) > 1 else 2927963, int(sys.argv[2]) if len(sys.argv) > 2 else 30387995
rng = np.random.RandomState(1)
e = rng.randint(0, n, size=(2, e_und)).astype(np.int64)
ei = torch.from_numpy(np.concatenate([e, e[::-1]], axis=1)).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
csr = build_csr(ei, n, dev)
mh = eh._init_minhash_u32(n, dev); hll = eh._init_hll_u8(n, dev)
mh2 = torch.empty_like(mh); hll2 = torch.empty_like(hll)
cards = torch.empty(n, device=dev)
prm = eh._params(dev)
lib = ssa._native.lib()
E1 = ei.shape[1] + n
def run(use_mh, use_hll, use_cards):
    ms = c_float()
    csr.use_inferred_self_loops = True
    graph = csr.struct()
    rc = lib.ss_time_propagate(byref(graph), _ptr(mh) if use_mh else None, _ptr(mh2) if use_mh else None, 128,
                               _ptr(hll) if use_hll else None, _ptr(hll2) if use_hll else None, 256, _ptr(cards) if use_cards else None, 1,
                               byref(prm.struct) if use_cards else None, _stream(dev), 5, byref(ms))
    assert rc == 0, rc
    row = (512 if use_mh else 0) + (256 if use_hll else 0)
    b = (E1 + n) * row + 4 * E1 + 8 * n
    print(f'mh={use_mh} hll={use_hll} cards={use_cards}: {ms.value:.3f} ms  {b / ms.value / 1e6:.0f} GB/s')
run(True, True, True); run(True, True, False); run(True, False, False); run(False, True, False)
