"""Where the time of the CSR build's finish launch goes (library built with SS_EXTRA_FLAGS=-DSS_CSR_TIMING):
  python tools/probe_csr_timing.py [config] [uniform|powerlaw] [alpha]
sums over workgroups of the phases of a bucket workgroup (prepare / count / scan / place / stream), and the launch's timeline: the
latest time any workgroup passed each mark, relative to the earliest workgroup start (one build per reading)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import subgraph_sketching_amd as ssa

dev = torch.device('cuda:0')
lib = ssa._native.lib()
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'collab']
kind = sys.argv[2] if len(sys.argv) > 2 else 'uniform'
alpha = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
n = cfg['n']
ei = torch.from_numpy(bench.synthetic_graph(n, cfg['e_und'], kind, alpha)).to(dev)
lib.ss_csr_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()


def reset():
    lib.ss_csr_timing_read(buf, 1)
    # slot 8 is a minimum: start it at the largest value
    big = (ctypes.c_ulonglong * 16)(*([0] * 8 + [2 ** 64 - 1] + [0] * 7))
    lib.ss_csr_timing_write(big)


lib.ss_csr_timing_write.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ssa.build_csr(ei, n, dev, check=False)
torch.cuda.synchronize()
lines = []
for rep in range(6):
    reset()
    ssa.build_csr(ei, n, dev, check=False)
    torch.cuda.synchronize()
    lib.ss_csr_timing_read(buf, 0)
    v = list(buf)
    t0 = v[8]
    rel = {name: (v[i] - t0) / 100.0 if v[i] else None for i, name in
           ((15, 'last bucket workgroup decided'), (14, 'last dense bucket registered'), (9, 'last ordinary bucket done'),
            (10, 'last helper saw every bucket arrive'), (11, 'last helper counted'), (12, 'last helper past the counter barrier'),
            (13, 'last helper placed'))}
    lines.append(rel)
    if rep == 5:
        print(f'{sys.argv[1:]}: phases summed over workgroups (us):', [round(x / 100.0, 1) for x in v[:5]])
for name in lines[0]:
    vals = [l[name] for l in lines[1:] if l[name] is not None]
    print(f'  {name:42s}', ' '.join(f'{x:7.1f}' for x in vals), 'us after the first workgroup started')
