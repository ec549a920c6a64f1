"""The destination-grouped adjacency the propagation kernels pull from (replaces PyG's materialised x[src] + scatter-max,
reference hashing.py:28-45,148): CsrGraph (struct ss_csr_graph + its tensors), build_csr (ss_csr_build / ss_csr_build_cached),
the one-entry cache that recognises ELPH.forward's per-step edge_index by identity or by content, and the grouping of a link
set by its first node (the same builder, reference hashing.py:270-274 reads the rows of u once per pair)."""
import logging
import os
import weakref
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from . import _native, hll_tables, knobs
from ._runtime import CsrProtocolFault, mark_csr_protocol_faults_reported, _Span, _error_flag, _ptr, _stream


def default_hub_threshold(num_edges):
    return int(min(max(num_edges // 16384, 128), 1024))


class CsrGraph(object):
    """destination-grouped adjacency resident on the device (struct ss_csr_graph + the tensors it points to).
    `n_self_dev` (device int64[1]) holds max(edge_index) + 1 as computed by ss_csr_build; `use_inferred_self_loops`
    says whether the propagation adds those implicit self loops (build_hash_tables) or none (hll_prop / minhash_prop
    receive them explicitly in edge_index)."""

    def __init__(self, rowptr, col, num_nodes, n_self_dev, err, hub_rows, hub_count, hub_threshold, mega=None):
        self.rowptr, self.col, self.num_nodes, self.n_self_dev, self.err = rowptr, col, num_nodes, n_self_dev, err
        self.hub_rows, self.hub_count, self.hub_threshold = hub_rows, hub_count, hub_threshold
        self.mega_rows, self.mega_count, self.mega_scratch = mega if mega is not None else (None, None, None)
        self.has_hub_rows = True  # unknown (no host read of the device counters): keep the hub units
        self.pending_minhash = None  # (weakref to a LazyMinhash, perms, P, p): a deferred hop-1 MinHash table on this graph
        self.pending_lazies = []     # weakrefs to every LazyMinhash whose deferred launch refers to this graph
        self.num_edges = None
        self.fingerprint = None      # device buffer of ss_csr_build_cached (None: never reused)
        self.hub_report = None       # pinned host int32 the first-hop kernels report this graph's hub + mega row count into
        self.use_inferred_self_loops = False

    def struct(self, rows=None, mirrors=None):
        """rows = (begin, end): only those destination rows are computed (multi-GPU destination-range sharding).
        mirrors = (mh_ptrs, hll_ptrs, cards_ptrs): lists of device addresses (0 / None = absent) of the OTHER ranks' tables that
        receive every finished row as well (peer-write build, dist.PeerShard)"""
        begin, end = (0, 0) if rows is None else rows
        if rows is not None and end == 0:  # (0, 0) would mean "all rows" to the library: express the empty range at N
            begin = end = self.num_nodes
        hubs = self.has_hub_rows
        mega = hubs and self.mega_rows is not None
        extra = {}
        if self.hub_report is not None:  # (the device counters of the CSR build -> a pinned host word, see ElphHashes._hub_hint)
            extra.update(hub_report=self.hub_report.data_ptr(), report_hub_count=self.hub_count.data_ptr(),
                         report_mega_count=self.mega_count.data_ptr() if self.mega_count is not None else None)
        if mirrors is not None and len(mirrors[0]) > 0:
            n_mir = len(mirrors[0])
            if n_mir > _native.MAX_MIRRORS:
                raise ValueError(f'a peer-write build reaches at most {_native.MAX_MIRRORS} other ranks, got {n_mir}')
            arr = lambda ptrs: (c_void_p * 7)(*[int(p or 0) for p in ptrs] + [0] * (7 - n_mir))
            extra.update(n_mirrors=n_mir, mirror_mh=arr(mirrors[0]), mirror_hll=arr(mirrors[1]), mirror_cards=arr(mirrors[2]))
        return _native.CsrGraphStruct(**extra, rowptr=self.rowptr.data_ptr(), col=self.col.data_ptr(), num_nodes=self.num_nodes,
                                      n_self_loops=0,
                                      n_self_loops_dev=self.n_self_dev.data_ptr() if self.use_inferred_self_loops else None,
                                      hub_threshold=self.hub_threshold, reserved=0,
                                      hub_rows=self.hub_rows.data_ptr() if hubs else None,
                                      hub_count=self.hub_count.data_ptr() if hubs else None,
                                      mega_rows=self.mega_rows.data_ptr() if mega else None,
                                      mega_count=self.mega_count.data_ptr() if mega else None,
                                      mega_scratch=self.mega_scratch.data_ptr() if mega else None,
                                      row_begin=begin, row_end=end)


def _rebuild_csr_if_changed(csr, src, dst, err_flag):
    """ss_csr_build_cached into the buffers of `csr`: a device-side content check, then either nothing or an ordinary build"""
    lib = _native.lib()
    device, E, N = csr.rowptr.device, src.numel(), csr.num_nodes
    # deferred launches that still refer to this CSR run now, while it describes the graph they were recorded on
    for ref in csr.pending_lazies:
        lazy = ref()
        if lazy is not None:
            lazy.resolve()
    csr.pending_lazies = []
    csr.pending_minhash = None
    ws_bytes = lib.ss_csr_workspace_bytes(N, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    with _Span('csr_build', device):
        _native.check(lib.ss_csr_build_cached(_ptr(src), _ptr(dst), E, N, _ptr(csr.rowptr), _ptr(csr.col), _ptr(csr.n_self_dev),
                                              csr.hub_threshold, _ptr(csr.hub_rows), _ptr(csr.hub_count), _ptr(csr.mega_rows),
                                              _ptr(csr.mega_count), _ptr(err_flag), _ptr(ws), ws_bytes, _ptr(csr.fingerprint),
                                              _stream(device)), 'ss_csr_build_cached')
    return csr


def build_csr(edge_index, num_nodes, device, check=True, hub_threshold=None, err_flag=None, reuse=None, fingerprint=False):
    """CSR-by-destination of edge_index [2, E] (flow source -> target, reference hashing.py:34,44).
    check=True synchronises once to raise IndexError for endpoints outside [0, num_nodes); err_flag (a device-visible
    int32 tensor, see _DeferredErrors) takes the report instead and nothing synchronises.
    reuse: a CsrGraph built earlier for the same shape (num_nodes, number of edges, device, hub threshold) by a non-strict build:
    its buffers are rebuilt only if the CONTENT of edge_index differs (device-side fingerprint, no host read) -- ELPH.forward hands
    over the same self-looped edges in a fresh tensor every training step (reference models/elph.py:186).
    fingerprint=True: this build leaves the sums behind that a later `reuse` compares with (one extra streaming pass over the edges)"""
    lib = _native.lib()
    ei = edge_index.to(device=device, dtype=torch.int64)
    if ei.dim() != 2 or ei.size(0) != 2:
        raise ValueError('edge_index must have shape [2, num_edges]')
    src, dst = ei[0].contiguous(), ei[1].contiguous()
    E = src.numel()
    if hub_threshold is None:
        hub_threshold = knobs.HUB_THRESHOLD if knobs.HUB_THRESHOLD is not None else default_hub_threshold(E)
    if (reuse is not None and not check and E > 0 and num_nodes > 0 and reuse.num_nodes == num_nodes and reuse.num_edges == E
            and reuse.hub_threshold == hub_threshold and reuse.rowptr.device == device and reuse.fingerprint is not None):
        return _rebuild_csr_if_changed(reuse, src, dst, err_flag)
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int64, device=device)
    col = torch.empty(max(E, 1), dtype=torch.int32, device=device)
    # one small block of device counters, all cleared by the kernels: int64 n_self | int32 hub rows, error | int32 mega rows, slices
    flags = torch.empty(3, dtype=torch.int64, device=device)
    flags32 = flags.view(torch.int32)
    n_self_dev = flags[0:1]
    hub_count = flags32[2:3]
    mega_count = flags32[4:6]
    # strict mode reads its own flag together with the counters below; a non-strict build passes NO flag (a shared one
    # would stay set and make the next strict call raise for valid inputs)
    if err_flag is not None:
        check, err = False, err_flag
    else:
        err = flags32[3:4] if check else None
    if check:
        err.zero_()
    hub_rows = torch.empty(max(num_nodes, 1), dtype=torch.int32, device=device)
    # rows with more than SS_MEGA_SLICE in-edges ("mega rows") are walked slice by slice by all hub workgroups: list +
    # counters + one scratch slot per slice (a row has > MEGA_SLICE edges, so there are at most E / MEGA_SLICE of them and
    # at most three times as many slices)
    max_mega = E // _native.MEGA_SLICE + 1
    mega_rows = torch.empty((max_mega, _native.MEGA_DESC_WORDS), dtype=torch.int32, device=device)
    mega_scratch = torch.empty(3 * max_mega * _native.MEGA_SLOT_BYTES, dtype=torch.uint8, device=device)
    ws_bytes = lib.ss_csr_workspace_bytes(num_nodes, E)
    if ws_bytes == 0:
        raise NotImplementedError(f'graphs with {num_nodes} nodes are not supported by the CSR builder')
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    fp = None
    if fingerprint and not check and E > 0 and num_nodes > 0:
        fp = torch.zeros(_native.CSR_FINGERPRINT_BYTES, dtype=torch.uint8, device=device)
    with _Span('csr_build', device):
        if fp is not None:
            _native.check(lib.ss_csr_build_cached(_ptr(src), _ptr(dst), E, num_nodes, _ptr(rowptr), _ptr(col), _ptr(n_self_dev),
                                                  hub_threshold, _ptr(hub_rows), _ptr(hub_count), _ptr(mega_rows), _ptr(mega_count),
                                                  _ptr(err), _ptr(ws), ws_bytes, _ptr(fp), _stream(device)), 'ss_csr_build_cached')
        else:
            _native.check(lib.ss_csr_build(_ptr(src), _ptr(dst), E, num_nodes, _ptr(rowptr), _ptr(col), _ptr(n_self_dev),
                                           hub_threshold, _ptr(hub_rows), _ptr(hub_count), _ptr(mega_rows), _ptr(mega_count), _ptr(err),
                                           _ptr(ws), ws_bytes, _stream(device)), 'ss_csr_build')
    csr = CsrGraph(rowptr, col, num_nodes, n_self_dev, _error_flag(device), hub_rows, hub_count, hub_threshold,
                   mega=(mega_rows, mega_count, mega_scratch))
    csr.num_edges = E
    csr.fingerprint = fp  # a later build_csr(..., reuse=csr) compares contents with these sums
    if check:
        # the one synchronising read of strict mode brings the hub / mega row counts along: a graph without such rows
        # (every unskewed graph) then serves no hub units (leading workgroups that would find nothing to do)
        host = flags32.cpu()
        if int(host[3]) & _native.SS_CSR_ERR_PROTOCOL:
            mark_csr_protocol_faults_reported()
            raise CsrProtocolFault('this CSR build gave up a cross-workgroup wait after its time bound: the adjacency is incomplete')
        if int(host[3]):
            raise IndexError(f'edge_index refers to nodes outside [0, {num_nodes})')
        csr.has_hub_rows = bool(int(host[2]) or int(host[4]))
    return csr



class _CsrCache(object):
    """one-entry cache keyed on the identity + version of the edge_index tensor object.  ELPH.forward
    (reference models/elph.py:209-212) calls hll_prop and minhash_prop h times each with the SAME
    self-looped edge_index object; this builds its CSR once per forward.  A dead weak reference or a
    bumped `_version` (in-place edit) invalidates the entry, so recycled allocations are never trusted."""

    def __init__(self, check=lambda device, what: (True, None), hub_hint=None):
        self._ref, self._version, self._key, self._csr = None, None, None, None
        # (device, what) -> (check, err_flag) of build_csr: whether a build may synchronise to raise IndexError, or where it
        # reports instead (ElphHashes._bounds of the owner)
        self._check = check
        # (device, num_nodes, edge_index) -> (pinned report word, whether an earlier build of the shape listed no hub rows) or None
        self._hub_hint = hub_hint

    def get(self, edge_index, num_nodes, device):
        key = (num_nodes, tuple(edge_index.shape), str(device))
        if self._ref is not None and self._ref() is edge_index and self._version == edge_index._version and self._key == key:
            return self._csr
        check, err_flag = self._check(device, 'sketch propagation (edge_index)')
        # another tensor object (or an edited one) of the SAME shape: the cached CSR's buffers are rebuilt only if the contents
        # differ, decided on the device (knobs.REUSE_CSR_BY_CONTENT; strict builds read their flags back and always rebuild)
        reuse = self._csr if (knobs.REUSE_CSR_BY_CONTENT and self._key == key) else None
        csr = build_csr(edge_index, num_nodes, device, check=check, err_flag=err_flag, reuse=reuse, fingerprint=knobs.REUSE_CSR_BY_CONTENT)
        if self._hub_hint is not None and not check:  # (ElphHashes._hub_hint: no hub units for shapes that listed no hub rows)
            hint = self._hub_hint(device, num_nodes, edge_index)
            if hint is not None:
                csr.hub_report, csr.has_hub_rows = hint[0], not hint[1]
        self._ref, self._version, self._key, self._csr = weakref.ref(edge_index), edge_index._version, key, csr
        return csr


_default_csr_cache = _CsrCache()



def group_links_by_source(links, num_nodes, device=None):
    """int32 [L] permutation of the pair indices of `links` (int64 [L, 2] on the device) in which the pairs of one first node
    are consecutive (torch-style negative ids wrapped, ids out of range grouped with node 0 -- nothing is dropped)"""
    device = device or links.device
    lib = _native.lib()
    L = links.size(0)
    if L >= 1 << 31:
        raise ValueError('link sets of 2^31 pairs and more cannot be grouped in one call')
    order = torch.empty(max(L, 1), dtype=torch.int32, device=device)
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int64, device=device)
    ws_bytes = lib.ss_csr_workspace_bytes(num_nodes, L)
    if ws_bytes == 0:
        raise NotImplementedError(f'link grouping is not supported for {num_nodes} nodes')
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    with _Span('group_links', device):
        _native.check(lib.ss_group_links_by_source(_ptr(links), L, num_nodes, _ptr(order), _ptr(rowptr), _ptr(ws), ws_bytes, _stream(device)),
                      'ss_group_links_by_source')
    return order[:L]


