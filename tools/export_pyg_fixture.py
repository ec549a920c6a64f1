#!/usr/bin/env python3
"""Export golden vectors of PyG's gcn_norm and torch_sparse.spmm -- run this WHERE torch_geometric AND torch_sparse ARE INSTALLED.

The SIGN node-feature preprocessing of BUDDY (reference src/datasets/elph.py:87-110) calls
`torch_geometric.nn.conv.gcn_conv.gcn_norm(edge_index, edge_weight.float(), num_nodes)` and
`torch_sparse.spmm(edge_index, edge_weight, N, N, x)`.  Neither package is in the build / GPU image, so
subgraph_sketching_amd/sign.py and oracle.gcn_norm / oracle.spmm RESTATE their semantics (parity unpinned).
This tool pins them from any machine that has the packages:

    python tools/export_pyg_fixture.py       # writes tests/golden/g13_pyg_sign.npz
    git add tests/golden/g13_pyg_sign.npz

Inputs are a fixed graph with the cases the restatement had to decide on (existing self loops incl. a duplicated one,
duplicate edges, isolated nodes, integer-valued and fractional weights) and fixed features; outputs are gcn_norm's
(edge_index, edge_weight) and the spmm product, plus the sign_k = 0 and sign_k = 2 results of the reference's own loop
restated from datasets/elph.py:96-110 with the REAL packages.  tests/test_oracle_golden.py::test_pyg_sign_fixture (CPU, oracle)
and tests/test_gpu_parity.py::test_sign_features_vs_pyg_fixture (GPU kernel) consume the file and skip loudly while it is absent.
Data only: no PyG / torch_sparse source text is copied."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden', 'g13_pyg_sign.npz')


def inputs():
    rng = np.random.RandomState(13)
    n, e = 300, 2400
    ei = rng.randint(0, n - 10, size=(2, e)).astype(np.int64)          # nodes n-10 .. n-1 stay isolated
    ei[:, :8] = np.array([[5, 5, 9, 9, 11, 40, 40, 77], [5, 5, 9, 9, 11, 41, 41, 77]])  # self loops (5 twice), duplicate edges
    w_int = rng.randint(1, 5, size=e).astype(np.float32)
    w_frac = (rng.random_sample(e) * 3 + 0.1).astype(np.float32)
    x = rng.randn(n, 20).astype(np.float32)
    return n, ei, w_int, w_frac, x


def main():
    try:
        from torch_geometric.nn.conv.gcn_conv import gcn_norm
        import torch_geometric
        import torch_sparse
    except ImportError as exc:
        sys.exit(f'{exc}: run this tool on a machine with torch_geometric and torch_sparse installed')
    n, ei, w_int, w_frac, x = inputs()
    g = {'num_nodes': np.asarray(n), 'edge_index': ei, 'w_int': w_int, 'w_frac': w_frac, 'x': x,
         'versions': np.asarray(f'torch_geometric {torch_geometric.__version__}, torch_sparse {torch_sparse.__version__}, torch {torch.__version__}')}
    tx = torch.from_numpy(x)
    for tag, w in (('int', w_int), ('frac', w_frac)):
        nei, nw = gcn_norm(torch.from_numpy(ei), torch.from_numpy(w).float(), n)
        g[f'norm_edge_index_{tag}'], g[f'norm_weight_{tag}'] = nei.numpy(), nw.numpy()
        prod = torch_sparse.spmm(nei, nw, n, n, tx)
        g[f'spmm_{tag}'] = prod.numpy()
        g[f'sign_k0_{tag}'] = prod.numpy()                                          # datasets/elph.py:101-103
        g[f'sign_k2_{tag}'] = torch.cat([tx, prod, torch_sparse.spmm(nei, nw, n, n, tx)], dim=-1).numpy()   # :105-109 (multiplies data.x each time)
    np.savez_compressed(OUT, **g)
    print('wrote', OUT, g['versions'])


if __name__ == '__main__':
    main()
