"""Host mirror of HashDataset._generate_sign_features (reference datasets/elph.py:87-110): the
SIGN-style node-feature preprocessing BUDDY runs once per split -- `gcn_norm` of the adjacency, then `torch_sparse.spmm`.

Both PyG's `gcn_norm` and `torch_sparse.spmm` are absent from this image, so their semantics are RESTATED here (parity
unpinned for this row): gcn_norm = add_remaining_self_loops(fill 1) + D^-1/2 A D^-1/2 with D the weighted in-degree over
`col`; spmm(index, value, m, n, x) = scatter_add(value * x[index[1]], index[0]).  The normalisation is elementwise torch
(plumbing); the product is the hand-written ss_spmm_csr kernel, which accumulates every output element in the
reference's edge order (see include/subgraph_sketch.h), so only the rounding of deg^-1/2 can differ from a CPU run.
"""
import torch

from . import _native
from .hashing import _compute_device, _ptr, _stream


def gcn_norm(edge_index, edge_weight, num_nodes):
    """torch_geometric.nn.conv.gcn_conv.gcn_norm(edge_index, edge_weight, num_nodes) with its defaults
    (improved=False, add_self_loops=True, flow='source_to_target'); returns (edge_index', edge_weight')"""
    row, col = edge_index[0], edge_index[1]
    w = edge_weight.to(torch.float32)
    keep = row != col
    loop_w = torch.ones(num_nodes, dtype=torch.float32, device=w.device)
    loop_w[row[~keep]] = w[~keep]  # an existing self loop keeps its weight (add_remaining_self_loops)
    loops = torch.arange(num_nodes, device=row.device, dtype=row.dtype)
    ei = torch.cat([edge_index[:, keep], loops.repeat(2, 1)], dim=1)
    w = torch.cat([w[keep], loop_w])
    deg = torch.zeros(num_nodes, dtype=torch.float32, device=w.device).scatter_add_(0, ei[1], w)
    dinv = deg.pow(-0.5)
    dinv.masked_fill_(dinv == float('inf'), 0)
    return ei, dinv[ei[0]] * w * dinv[ei[1]]


def spmm(index, value, m, n, matrix):
    """torch_sparse.spmm(index, value, m, n, matrix): out[index[0]] += value * matrix[index[1]]; [m, F] float32"""
    if m != n or matrix.size(0) != n:
        raise ValueError('square propagation matrices only (the reference passes x.shape[0] for both)')
    device = _compute_device(matrix, index)
    home = matrix.device
    x = matrix.to(device=device, dtype=torch.float32).contiguous()
    F = x.size(1)
    pad = (-F) % 4
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    row = index[0].to(device)
    order = torch.sort(row, stable=True)[1]  # CSR order == the reference's edge order inside every row
    col = index[1].to(device)[order].to(torch.int32).contiguous()
    val = value.to(device=device, dtype=torch.float32)[order].contiguous()
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=m), 0)
    out = torch.empty((m, F + pad), dtype=torch.float32, device=device)
    if col.numel() == 0:
        col = torch.zeros(1, dtype=torch.int32, device=device)
        val = torch.zeros(1, dtype=torch.float32, device=device)
    _native.check(_native.lib().ss_spmm_csr(_ptr(rowptr), _ptr(col), _ptr(val), m, _ptr(x), F + pad, _ptr(out), _stream(device)),
                  'ss_spmm_csr')
    out = out[:, :F] if pad else out
    return out if home == device else out.to(home)


def generate_sign_features(x, edge_index, edge_weight, sign_k):
    """HashDataset._generate_sign_features (reference datasets/elph.py:87-110) for node features x [N, F].
    sign_k == 0: one propagation step.  sign_k > 0: [x, A x, A x, ...] -- the reference's loop multiplies `data.x`, not the
    previous product, every time (datasets/elph.py:105-107); that behaviour is kept."""
    n = x.size(0)
    ei, w = gcn_norm(edge_index, edge_weight.float(), n)
    if sign_k == 0:
        return spmm(ei, w, n, n, x)
    xs = [x]
    for _ in range(sign_k):
        xs.append(spmm(ei, w, n, n, x))
    return torch.cat(xs, dim=-1)
