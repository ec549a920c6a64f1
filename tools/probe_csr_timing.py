import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import bench
import subgraph_sketching_amd as ssa
dev = torch.device('cuda:0')
lib = ssa._native.lib()
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'ppa']
n = cfg['n']
ei = torch.from_numpy(bench.synthetic_graph(n=n, e_und=cfg['e_und'])).to(dev)
for _ in range(3): ssa.build_csr(ei, n, dev, check=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.ss_csr_timing_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ss_csr_timing_read(buf, 1)
reps = 5
for _ in range(reps): ssa.build_csr(ei, n, dev, check=False)
torch.cuda.synchronize()
lib.ss_csr_timing_read(buf, 1)
fine = (n + 127) // 128
t = np.array(list(buf)[:6], dtype=np.float64) / reps
print('ticks per build per phase (100 MHz wall clock -> us summed over workgroups):', t / 100)
print('per workgroup us (assuming', fine, 'buckets):', t / 100 / fine)
