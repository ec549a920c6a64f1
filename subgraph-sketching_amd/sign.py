"""Host mirror of HashDataset._generate_sign_features (reference datasets/elph.py:87-110): the
SIGN-style node-feature preprocessing BUDDY runs once per split -- `gcn_norm` of the adjacency, then `torch_sparse.spmm`.

Both PyG's `gcn_norm` and `torch_sparse.spmm` are absent from this image, so their semantics are RESTATED here: gcn_norm =
add_remaining_self_loops(fill 1) + D^-1/2 A D^-1/2 with D the weighted in-degree over `col`; spmm(index, value, m, n, x) =
scatter_add(value * x[index[1]], index[0]).  G15 (tests/golden) pins the reference's OWN _generate_sign_features under that
restatement; a fixture from the real packages (tools/export_pyg_fixture.py) would pin the restatement itself.

`generate_sign_features` is one chain of library calls (round 4): the edge indices grouped stably by column and by row on the
device (ss_csr_group_ids + ss_csr_sort_rows: the CSR builder with the edge index as payload), ss_gcn_degree (weighted degrees
summed in edge order, deg^-1/2), ss_sign_spmm (normalised weights formed on the fly, every output element accumulated in the
reference's edge order) -- no normalised edge list, no torch sort.  `gcn_norm` and `spmm` stay as the two functions the
reference calls, for callers that want them separately.
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


def group_ids_stable(ids, num_ids, err=None, only_if=None):
    """(rowptr int64[num_ids + 1], order int32[E]): the positions of `ids` (device int64[E]) grouped by id, ascending inside every
    group -- what a stable sort of `ids` gives, built by the CSR builder (ss_csr_group_ids) and a per-row sort (ss_csr_sort_rows).
    Ids outside [0, num_ids) raise IndexError (one host read of a flag word) unless the caller passes its own `err` word and
    checks it later.  only_if: device int32 word -- the per-row sort runs only if it is non-zero (else: grouped, any order)."""
    lib = _native.lib()
    device = ids.device
    E = ids.numel()
    rowptr = torch.empty(num_ids + 1, dtype=torch.int64, device=device)
    order = torch.empty(max(E, 1), dtype=torch.int32, device=device)
    own_err = err is None
    if own_err:
        err = torch.zeros(1, dtype=torch.int32, device=device)
    ws_bytes = max(lib.ss_csr_workspace_bytes(num_ids, E), lib.ss_csr_sort_workspace_bytes(E))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=device)
    _native.check(lib.ss_csr_group_ids(_ptr(ids), E, num_ids, _ptr(order), _ptr(rowptr), _ptr(err), _ptr(ws), ws_bytes, _stream(device)),
                  'ss_csr_group_ids')
    _native.check(lib.ss_csr_sort_rows(_ptr(rowptr), _ptr(order), num_ids, E, _ptr(only_if) if only_if is not None else None, _ptr(ws), ws_bytes,
                                       _stream(device)), 'ss_csr_sort_rows')
    if own_err and int(err.item()):
        raise IndexError('ids outside [0, num_ids)')
    return rowptr, order[:E]


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
    row = index[0].to(device).contiguous()
    rowptr, order = group_ids_stable(row, m)  # CSR order == the reference's edge order inside every row
    order = order.to(torch.int64)
    col = index[1].to(device)[order].to(torch.int32).contiguous()
    val = value.to(device=device, dtype=torch.float32)[order].contiguous()
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
    previous product, every time (datasets/elph.py:105-107); that behaviour is kept (the product is computed once)."""
    lib = _native.lib()
    n = x.size(0)
    device = _compute_device(x, edge_index)
    home = x.device
    xd = x.to(device=device, dtype=torch.float32).contiguous()
    F = xd.size(1)
    pad = (-F) % 4
    if pad:
        xd = torch.nn.functional.pad(xd, (0, pad))
    row = edge_index[0].to(device).contiguous()
    col = edge_index[1].to(device).contiguous()
    w = edge_weight.to(device=device, dtype=torch.float32).contiguous()
    if w.numel() == 0:  # (kernels never dereference an empty edge list, but the pointers must exist)
        w = torch.zeros(1, dtype=torch.float32, device=device)
    E = row.numel()
    err = torch.zeros(1, dtype=torch.int32, device=device)
    scan = torch.empty(lib.ss_gcn_scan_bytes(n), dtype=torch.uint8, device=device)  # first int32 word: some weight differs from 1
    _native.check(lib.ss_gcn_scan_edges(_ptr(row), _ptr(col), _ptr(w), E, n, _ptr(scan), _stream(device)), 'ss_gcn_scan_edges')
    # for the degrees: index_add over `col` in edge order -- the order only matters when some weight differs from 1
    rowptr_c, order_c = group_ids_stable(col, n, err, only_if=scan)
    rowptr_r, order_r = group_ids_stable(row, n, err)  # for the product: scatter-add over `row` in edge order
    dinv = torch.empty(n, dtype=torch.float32, device=device)
    loop_w = torch.empty(n, dtype=torch.float32, device=device)
    out = torch.empty((n, F + pad), dtype=torch.float32, device=device)
    if E == 0:
        row = col = torch.zeros(1, dtype=torch.int64, device=device)
    _native.check(lib.ss_gcn_degree(_ptr(rowptr_c), _ptr(order_c), _ptr(row), _ptr(w), n, _ptr(scan), _ptr(dinv), _ptr(loop_w), _stream(device)),
                  'ss_gcn_degree')
    _native.check(lib.ss_sign_spmm(_ptr(rowptr_r), _ptr(order_r), _ptr(col), _ptr(w), _ptr(dinv), _ptr(loop_w), _ptr(scan), n, _ptr(xd), F + pad, _ptr(out),
                                   _stream(device)), 'ss_sign_spmm')
    # (the one host read of the chain) the groupings' flag, or ss_gcn_scan_edges' own word for ids outside [0, N) -- negative ones
    # included, which the groupings wrap torch-style; the product kernel does not follow an id out of range
    if int((err + scan[4:8].view(torch.int32)).item()):
        raise IndexError('edge_index refers to nodes outside [0, num_nodes)')
    ax = out[:, :F] if pad else out
    ax = ax if home == device else ax.to(home)
    if sign_k == 0:
        return ax
    return torch.cat([x.to(torch.float32)] + [ax] * sign_k, dim=-1)
