"""probe: sign.generate_sign_features at ogbl-collab size, F = 128 (the SIGN preprocessing of HashDataset): wall time per call and
the per-kernel times under rocprofv3 (tools/kstats_cmd.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import subgraph_sketching_amd as ssa
dev = torch.device('cuda:0')
n = bench.N_NODES
ei = torch.from_numpy(bench.synthetic_graph()).to(dev)
w = torch.ones(ei.size(1), device=dev)
x = torch.randn(n, 128, device=dev)
for k in (0, 2):
    for _ in range(3): ssa.sign.generate_sign_features(x, ei, w, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ssa.sign.generate_sign_features(x, ei, w, k)
    torch.cuda.synchronize()
    print(f'sign_k={k}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call (N={n}, E={ei.size(1)}, F=128)')
