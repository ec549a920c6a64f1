"""probe: RA (resource allocation, reference heuristics.py:51-70) on the collab-like graph: GPU kernel vs the scipy
expression the reference evaluates (restated here, numpy indices) vs the C oracle, same links"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as ssp
import bench
from subgraph_sketching_amd import heuristics as hz
from oracle import oracle
dev = torch.device('cuda:0')
n = bench.N_NODES
ei = bench.synthetic_graph()
A = ssp.csr_matrix((np.ones(ei.shape[1], dtype=int), (ei[0], ei[1])), shape=(n, n))
L = 2_662_400
links = np.random.RandomState(0).randint(0, n, size=(L, 2)).astype(np.int64)
links[: L // 2] = ei.T[np.random.RandomState(1).randint(0, ei.shape[1], size=L // 2)]  # positives share neighbours
lk = torch.from_numpy(links).to(dev)
t0 = time.perf_counter(); adj = hz.DeviceAdjacency(A, dev); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'DeviceAdjacency (host canonicalise + upload): {1e3 * (t1 - t0):.1f} ms')
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = hz.RA(adj, lk, batch_size=2000000)[0]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'GPU RA: {L} links in {1e3 * (t1 - t0):.2f} ms = {L / (t1 - t0) / 1e6:.0f} M links/s', flush=True)
nnz_bytes = (np.diff(A.indptr)[links[:, 0]].sum() + np.diff(A.indptr)[links[:, 1]].sum()) * 4
print(f'row bytes touched (col ids only, unit weights): {nnz_bytes / 1e6:.0f} MB -> {nnz_bytes / (t1 - t0) / 1e9:.0f} GB/s of random 4-B-granular row reads')
S = 200_000
t0 = time.perf_counter(); want = oracle.common_neighbour_scores(A, links[:S], 'RA'); t1 = time.perf_counter()
print(f'C oracle (1 core): {S} links in {t1 - t0:.3f} s = {S / (t1 - t0) / 1e6:.2f} M links/s; equal to GPU: {np.array_equal(want, s[:S].cpu().numpy())}')
t0 = time.perf_counter()
with np.errstate(divide='ignore'):
    mult = 1 / A.sum(axis=0)
mult[np.isinf(mult)] = 0
A_ = A.multiply(mult).tocsr()
ref = np.array(np.sum(A[links[:S, 0]].multiply(A_[links[:S, 1]]), 1)).flatten().astype(np.float32)
t1 = time.perf_counter()
print(f'scipy expression of the reference: {S} links in {t1 - t0:.3f} s = {S / (t1 - t0) / 1e6:.2f} M links/s; equal to GPU: {np.array_equal(ref, s[:S].cpu().numpy())}')
