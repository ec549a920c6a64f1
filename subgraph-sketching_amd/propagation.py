"""MinhashPropagation / HllPropagation (reference hashing.py:28-45: PyG MessagePassing(aggr='max') wrappers) as launches of
the HIP propagation kernels over a cached CSR, with the deferred forms ELPH.forward's call sequence profits from."""
import logging
import os
import weakref
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from . import _native, hll_tables, knobs
from ._runtime import _Span, _check_sizes, _compute_device, _ptr, _stream
from .containers import LazyMinhash, _packed_hll_of, _packed_minhash_of, _tag, unpack_minhash
from .csr import _default_csr_cache


def _propagate(csr, mh_in, hll_in, device, cards_out=None, cards_stride=0, params=None, mh_out=None, hll_out=None, rows=None, mirrors=None):
    """one hop; returns (mh_out or None, hll_out or None).  mh_in packed int32 [N,P], hll_in uint8 [N,M];
    rows = (begin, end) restricts the destination rows written (inputs are always the full tables)"""
    N = csr.num_nodes
    if mh_in is not None and mh_out is None:
        mh_out = torch.empty_like(mh_in)
    if hll_in is not None and hll_out is None:
        hll_out = torch.empty_like(hll_in)
    P = mh_in.size(1) if mh_in is not None else 0
    M = hll_in.size(1) if hll_in is not None else 0
    prm = byref(params.struct) if params is not None else None
    graph = csr.struct(rows, mirrors)
    with _Span('propagate' if (mh_in is not None and hll_in is not None) else ('propagate_mh' if hll_in is None else 'propagate_hll'), device):
        _native.check(_native.lib().ss_propagate(byref(graph), _ptr(mh_in), _ptr(mh_out), P, _ptr(hll_in), _ptr(hll_out), M,
                                                 _ptr(cards_out), cards_stride, prm, _stream(device)), 'ss_propagate')
    return mh_out, hll_out


def _hop0_marker(x, device):
    """(perms, p) if x is an unmodified hop-0 tensor produced by initialise_minhash / initialise_hll on `device`"""
    tag = getattr(x, '_ss_hop0', None)
    if tag is None or tag[0] != x._version or x.device != device:
        return None
    return tag[1], tag[2]


def _first_hop_from_ids(csr, device, perms, num_perm, p, mh_out, hll_out, cards_out=None, params=None):
    """ss_first_hop for one sketch; returns False when the fused kernel has no variant for (num_perm, p)"""
    graph = csr.struct()
    with _Span('first_hop', device):
        rc = _native.lib().ss_first_hop(byref(graph), _ptr(perms[0]) if perms is not None else None,
                                        _ptr(perms[1]) if perms is not None else None, num_perm, _ptr(mh_out), p,
                                        _ptr(hll_out), _ptr(cards_out), 1 if cards_out is not None else 0,
                                        byref(params.struct) if cards_out is not None else None, _stream(device))
    if rc == -4:
        return False
    _native.check(rc, 'ss_first_hop')
    return True


class MinhashPropagation(object):
    """drop-in for reference hashing.py:28-35: out[i] = min over in-neighbours (edges j -> i) of x[j];
    rows without an in-edge are 0.  x: int64 [N, P] with values in [0, 2^32)."""

    def __init__(self, csr_cache=None, after_host_copy=None, defer_first_hop=None, defer_table_hop=None):
        """after_host_copy: called once a result has been copied back to a CPU caller (the copy has waited for the launches, so
        the owner's deferred bounds report is final and is raised from the offending call itself).
        defer_first_hop / defer_table_hop: None = the module defaults knobs.DEFER_FIRST_HOP / knobs.DEFER_TABLE_HOP (environment overrides
        SS_FUSED_STAGE / SS_DEFER_TABLE_HOP are for tests and measurements); ElphHashes passes its constructor arguments"""
        self._cache = csr_cache or _default_csr_cache
        self._after_host_copy = after_host_copy
        self.defer_first_hop, self.defer_table_hop = defer_first_hop, defer_table_hop

    def _to_caller(self, out, x, device):
        if x.device == device:
            return out
        out = out.to(x.device)
        if self._after_host_copy is not None:
            self._after_host_copy()
        return out

    @torch.no_grad()
    def forward(self, x, edge_index):
        _check_sizes(x.size(1), 8)
        device = _compute_device(x, edge_index)
        csr = self._cache.get(edge_index, x.size(0), device)
        hop0 = _hop0_marker(x, device)
        out_u32 = None
        if hop0 is not None and hop0[0] is not None:
            out_u32 = torch.empty((x.size(0), x.size(1)), dtype=torch.int32, device=device)
            P, p = x.size(1), hop0[1]
            defer_first = knobs.DEFER_FIRST_HOP if self.defer_first_hop is None else self.defer_first_hop
            if (defer_first and knobs.LAZY_MINHASH and x.device == device and p == 8 and P % 64 == 0 and P <= 256
                    and x.size(0) * 256 <= knobs.FUSED_STAGE_MAX_TABLE_BYTES):
                perms = hop0[0]

                def fill(csr=csr, perms=perms, P=P, p=p, out=out_u32, device=device):
                    csr.pending_minhash = None
                    if not _first_hop_from_ids(csr, device, perms, P, p, out, None):  # pragma: no cover (shapes checked above)
                        raise RuntimeError('deferred MinHash first hop has no kernel for this shape')
                lazy = LazyMinhash(out_u32, pending=fill)
                csr.pending_minhash = (weakref.ref(lazy), perms, P, p)  # the hop-2 hll_prop on this CSR may take it over
                csr.pending_lazies.append(weakref.ref(lazy))
                return lazy
            if not _first_hop_from_ids(csr, device, hop0[0], x.size(1), hop0[1], out_u32, None):
                out_u32 = None
        if out_u32 is None:
            mh_in = _packed_minhash_of(x, device)
            if (knobs.DEFER_TABLE_HOP if self.defer_table_hop is None else self.defer_table_hop) and knobs.LAZY_MINHASH and x.device == device:
                out_u32 = torch.empty_like(mh_in)

                def fill(csr=csr, mh_in=mh_in, out=out_u32, device=device):
                    _propagate(csr, mh_in, None, device, mh_out=out)

                def fill_rows(rows, csr=csr, mh_in=mh_in, out=out_u32, device=device):
                    graph = csr.struct()
                    with _Span('propagate_mh_rows', device):
                        _native.check(_native.lib().ss_minhash_hop_rows(byref(graph), _ptr(mh_in), _ptr(out), mh_in.size(1), _ptr(rows),
                                                                        rows.numel(), _stream(device)), 'ss_minhash_hop_rows')
                lazy = LazyMinhash(out_u32, pending=fill, partial=fill_rows)
                csr.pending_lazies = [r for r in csr.pending_lazies if r() is not None] + [weakref.ref(lazy)]
                return lazy
            out_u32, _ = _propagate(csr, mh_in, None, device)
        if knobs.LAZY_MINHASH and x.device == device:
            return LazyMinhash(out_u32)
        out = unpack_minhash(out_u32)
        _tag(out, '_ss_u32', out_u32)
        return self._to_caller(out, x, device)

    __call__ = forward


class HllPropagation(object):
    """drop-in for reference hashing.py:38-45: out[i] = element-wise max over in-neighbours of x[j]"""

    def __init__(self, csr_cache=None, params_of=None, m=None, after_host_copy=None):
        """after_host_copy: see MinhashPropagation.  params_of(device) -> _DeviceParams and m: given by the ElphHashes that owns this module; the kernels then also
        produce the HLL++ cardinality of every output row (free: the registers are in flight) and ElphHashes.hll_count of
        that very tensor (reference models/elph.py:213) is answered without another pass over the table"""
        self._cache = csr_cache or _default_csr_cache
        self._params_of, self._m = params_of, m
        self._after_host_copy = after_host_copy

    _to_caller = MinhashPropagation._to_caller

    @torch.no_grad()
    def forward(self, x, edge_index):
        M = x.size(1)
        if M < 16 or M & (M - 1) or M > 65536:
            raise NotImplementedError(f'HLL rows must have 2^p registers, 4 <= p <= 16, got {M}')
        device = _compute_device(x, edge_index)
        csr = self._cache.get(edge_index, x.size(0), device)
        hop0 = _hop0_marker(x, device)
        out_u8 = None
        params = self._params_of(device) if (self._params_of is not None and M == self._m) else None
        counts = torch.empty(x.size(0), dtype=torch.float32, device=device) if params is not None else None
        if hop0 is not None and hop0[0] is None and M == 256:
            out_u8 = torch.empty((x.size(0), M), dtype=torch.uint8, device=device)
            if not _first_hop_from_ids(csr, device, None, 128, hop0[1], None, out_u8, counts, params):
                out_u8 = None
        pend = getattr(csr, 'pending_minhash', None)
        lazy = pend[0]() if pend is not None else None
        if out_u8 is None and lazy is not None and lazy._pending is not None and M == 256 and params is not None:
            # a hop-1 MinHash table is still owed on this CSR (deferred by minhash_prop): compute it together with these HLL rows
            _, perms, P, p = pend
            out_u8 = torch.empty((x.size(0), M), dtype=torch.uint8, device=device)
            graph = csr.struct()
            with _Span('fused_hop_stage', device):
                rc = _native.lib().ss_fused_hop_stage(byref(graph), _ptr(perms[0]), _ptr(perms[1]), P, _ptr(lazy._packed), None, p,
                                                      _ptr(_packed_hll_of(x, device)), None, _ptr(out_u8), _ptr(counts), 1,
                                                      byref(params.struct), _stream(device))
            if rc == 0:
                lazy._pending = None
                csr.pending_minhash = None
            else:  # pragma: no cover (shapes were checked when the work was deferred)
                out_u8 = None
        if out_u8 is None:
            _, out_u8 = _propagate(csr, None, _packed_hll_of(x, device), device, cards_out=counts, cards_stride=1, params=params)
        out = out_u8.view(torch.int8) if x.dtype != torch.uint8 else out_u8
        if out.dtype != x.dtype:
            out = out.to(x.dtype)
        _tag(out, '_ss_u8', out_u8)
        if counts is not None:
            _tag(out, '_ss_count', counts)
        return self._to_caller(out, x, device)

    __call__ = forward


