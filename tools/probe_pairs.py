"""probe: pair_features kernel time vs batch size and hop count (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import byref, c_float, c_void_p
import numpy as np, torch
import subgraph_sketching_amd as ssa
from subgraph_sketching_amd.hashing import _ptr, _stream
from argparse import Namespace

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 235868
lib = ssa._native.lib()
g = torch.Generator(device='cpu').manual_seed(0)
for h in (2, 3):
    eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
    prm = eh._params(dev)
    mh = [torch.randint(-2**31, 2**31 - 1, (n, 128), dtype=torch.int32, device=dev) for _ in range(h)]
    # realistic registers: geometric ranks
    hl = [torch.clamp((-torch.log2(torch.rand((n, 256), device=dev))).ceil() * (torch.rand((n, 256), device=dev) < 0.6), 0, 50).to(torch.uint8) for _ in range(h)]
    cards = torch.rand((n, h), device=dev) * 300
    mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh]); hl_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hl])
    for B in (65536, 262144, 1048576, 4194304):
        links = torch.randint(0, n, (B, 2), device=dev)
        out = torch.empty((B, h * (h + 2)), device=dev)
        ms = c_float()
        rc = lib.ss_time_pair_features(_ptr(links), B, n, h, mh_ptrs, 128, hl_ptrs, _ptr(cards), h, byref(prm.struct), 1, _ptr(out), _stream(dev), 10, byref(ms))
        assert rc == 0
        bytes_ = B * (2 * h * 768 + 16 + 8 * h + 4 * h * (h + 2))
        print(f'h={h} B={B}: {ms.value*1e3:.1f} us  {bytes_/ms.value/1e6:.0f} GB/s  {B/ms.value/1e3:.0f} Mpairs/s')
