"""probe: SIGN propagation (gcn_norm + spmm) on the collab-like graph, F = 128: hand-written ss_spmm_csr vs torch.sparse.mm"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from subgraph_sketching_amd import sign, _native
from subgraph_sketching_amd.hashing import _ptr, _stream
dev = torch.device('cuda:0')
n, F = bench.N_NODES, 128
ei = torch.from_numpy(bench.synthetic_graph()).to(dev)
w = torch.ones(ei.size(1), device=dev)
x = torch.randn(n, F, device=dev)
gei, gw = sign.gcn_norm(ei, w, n)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


print(f'generate_sign_features (gcn_norm + stable sort + kernel): {timed(lambda: sign.generate_sign_features(x, ei, w, 0)):.3f} ms')
order = torch.sort(gei[0], stable=True)[1]
col = gei[1][order].to(torch.int32).contiguous(); val = gw[order].contiguous()
rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev); rowptr[1:] = torch.cumsum(torch.bincount(gei[0], minlength=n), 0)
out = torch.empty((n, F), device=dev)
t = timed(lambda: _native.lib().ss_spmm_csr(_ptr(rowptr), _ptr(col), _ptr(val), n, _ptr(x), F, _ptr(out), _stream(dev)), 50)
nnz = col.numel()
print(f'ss_spmm_csr alone: {t * 1e3:.1f} us; algorithmic bytes {(nnz + n) * F * 4 + nnz * 8 + n * 8:,} -> {((nnz + n) * F * 4 + nnz * 8 + n * 8) / t / 1e6:.0f} GB/s')
A = torch.sparse_coo_tensor(gei, gw, (n, n)).coalesce().to_sparse_csr()
t2 = timed(lambda: torch.sparse.mm(A, x), 50)
print(f'torch.sparse.mm (rocSPARSE, CSR): {t2 * 1e3:.1f} us')
