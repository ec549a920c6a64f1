"""What build_hash_tables / minhash_prop hand back (reference hashing.py:139-165 returns {hop: {'hll': int8 [N, M], 'minhash':
int64 [N, P]}}): HopSketch -- the reference-shaped leaves, materialised lazily over the packed tables the kernels use --,
LazyMinhash -- the int64 tensor minhash_prop owes its caller --, the packed on-disk format, pack / unpack."""
from collections import OrderedDict
from collections.abc import Mapping
import logging
import os
import weakref
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from . import _native, hll_tables, knobs
from ._runtime import _compute_device, _ptr, _stream


class HopSketch(Mapping):
    """{'hll': int8[N, M], 'minhash': int64[N, P]} of one hop, backed by the packed device tables.

    `mh_u32` (torch.int32 holding uint32 bit patterns) and `hll_u8` are what the kernels read.  The
    reference-shaped leaves are created on first access, on `home` (the device the reference would have
    left them on: where edge_index lived).  A read-only Mapping rather than a dict subclass so that every way
    of reading it (indexing, dict(x), {**x}, .items()) goes through the lazy materialisation."""
    _KEYS = ('hll', 'minhash')

    def __init__(self, mh_u32, hll_u8, home, make_packed=None):
        self._mh_u32 = mh_u32
        self._hll_u8 = hll_u8
        self._make_packed = make_packed  # deferred producer of (mh_u32, hll_u8): hop 0 is only built if somebody reads it
        self._leaves = {}
        self._leaf_versions = {}
        self.home = home

    def _ensure_packed(self):
        if self._mh_u32 is None and self._make_packed is not None:
            self._mh_u32, self._hll_u8 = self._make_packed()
            self._make_packed = None

    @property
    def mh_u32(self):
        self._ensure_packed()
        return self._mh_u32

    @property
    def hll_u8(self):
        self._ensure_packed()
        return self._hll_u8

    def __getitem__(self, key):
        if key not in self._KEYS:
            raise KeyError(key)
        val = self._leaves.get(key)
        if val is None:
            val = self.hll_u8.view(torch.int8) if key == 'hll' else unpack_minhash(self.mh_u32)
            if val.device != self.home:
                val = val.to(self.home)
            self._leaves[key] = val
            self._leaf_versions[key] = val._version
        return val

    def packed(self, device):
        """(mh_u32, hll_u8) for the kernels.  If a caller edited a materialised leaf in place (the reference's dict
        holds ordinary tensors, so that is legal) the packed twin is rebuilt from the edited leaf first."""
        for key in self._KEYS:
            leaf = self._leaves.get(key)
            if leaf is not None and leaf._version != self._leaf_versions[key]:
                if key == 'minhash':
                    self._mh_u32 = pack_minhash(leaf, device)
                elif leaf.data_ptr() != self.hll_u8.data_ptr():  # a view of the packed table edits it directly
                    self._hll_u8 = leaf.to(device).contiguous().view(torch.uint8)
                self._leaf_versions[key] = leaf._version
        return self.mh_u32, self.hll_u8

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)

    def __reduce__(self):
        # pickles (torch.save, datasets/elph.py:204) as a plain mapping of the two reference-shaped tensors; OrderedDict
        # because it is what torch.load's default weights_only unpickler accepts as a callable (torch >= 2.6)
        return (OrderedDict, ([(k, self[k]) for k in self._KEYS],))


# {hop: HopSketch}: a plain dict, so that torch.save / torch.load (weights_only) treat it exactly like the reference's
SketchTable = dict


PACKED_FORMAT = 'subgraph-sketch-packed-v1'


def _stamp_tables(cards, tables_id):
    """remember which HLL++ tables produced these cardinalities (python attribute: survives as long as the tensor object)"""
    try:
        cards._ss_tables = tables_id
    except Exception:  # pragma: no cover
        pass
    return cards


def save_sketches(path, table, cards, hll_tables_id=None):
    """packed on-disk cache: uint32 MinHash + uint8 HLL per hop (768 B per node and hop at the defaults instead of the
    1 280 B of the reference's int64/int8 `torch.save(hashes)` cache, datasets/elph.py:204).  Plain tensors and
    scalars only, so `torch.load(..., weights_only=True)` reads it.  The identity of the HLL++ tables that produced
    `cards` (hll_tables.table_id; taken from the stamp build_hash_tables leaves on `cards` unless given) is stored too:
    load_sketches / get_subgraph_features refuse to combine it with another table."""
    if hll_tables_id is None:
        hll_tables_id = getattr(cards, '_ss_tables', None)
    hops = {}
    for k, entry in table.items():
        if isinstance(entry, HopSketch):
            mh, hll = entry.packed(entry.mh_u32.device)
        else:
            device = _compute_device(entry['minhash'], entry['hll'])
            mh, hll = _packed_minhash_of(entry['minhash'], device), _packed_hll_of(entry['hll'], device)
        hops[int(k)] = {'minhash_u32': mh.cpu(), 'hll_u8': hll.cpu()}
    torch.save({'format': PACKED_FORMAT, 'hops': hops, 'cards': cards.cpu(), 'hll_tables': hll_tables_id or 'unknown'}, path)


def load_sketches(path, device=None, expect=None):
    """read a packed cache (save_sketches) or the reference's own cache files back into (SketchTable, cards).
    The reference's format ({k: {'hll': int8, 'minhash': int64}}) is returned as loaded -- get_subgraph_features
    accepts it directly; pass the cards file separately in that case.
    expect: an ElphHashes (or a table id string); a packed cache whose cardinalities were produced with OTHER HLL++ tables
    raises ValueError instead of being mixed with this engine's estimates."""
    blob = torch.load(path, map_location='cpu', weights_only=True)
    if not (isinstance(blob, dict) and blob.get('format') == PACKED_FORMAT):
        return blob, None
    device = device or _compute_device()
    table = SketchTable()
    for k, entry in blob['hops'].items():
        table[int(k)] = HopSketch(entry['minhash_u32'].to(device), entry['hll_u8'].to(device), device)
    cached_id = blob.get('hll_tables', 'unknown')
    want = expect if (expect is None or isinstance(expect, str)) else expect.tables_id  # an ElphHashes, or its table id
    if want is not None and cached_id != 'unknown' and not hll_tables.same_tables(cached_id, want):
        raise ValueError(f'{path} holds cardinalities made with HLL++ tables {cached_id}, this engine uses {want}')
    cards = blob['cards'].to(device)
    return table, (_stamp_tables(cards, cached_id) if cached_id != 'unknown' else cards)


def pack_minhash(x, device=None):
    """int64 [.., P] (values < 2^32, reference hashing.py:124) -> packed uint32 bit patterns in torch.int32"""
    device = device or _compute_device(x)
    x = x.to(device=device, dtype=torch.int64).contiguous()
    out = torch.empty(x.shape, dtype=torch.int32, device=device)
    _native.check(_native.lib().ss_pack_minhash(_ptr(x), _ptr(out), x.numel(), _stream(device)), 'ss_pack_minhash')
    return out


def unpack_minhash(x_u32):
    out = torch.empty(x_u32.shape, dtype=torch.int64, device=x_u32.device)
    _native.check(_native.lib().ss_unpack_minhash(_ptr(x_u32), _ptr(out), x_u32.numel(), _stream(x_u32.device)),
                  'ss_unpack_minhash')
    return out



class LazyMinhash(torch.Tensor):
    """The int64 [N, P] tensor `minhash_prop` owes its caller (reference hashing.py:28-35 returns int64), backed by the
    packed uint32 table the kernel actually wrote.  ELPH.forward (reference models/elph.py:209-212) only ever hands the
    tensor back to this engine (next hop, get_subgraph_features), which reads the packed table directly; the 8-byte
    copy -- 241 MB per hop at ogbl-collab size, 70 us -- is made the first time anything ELSE touches the tensor: every
    torch operator (indexing, comparison, .cpu(), printing, torch.save ...) sees an ordinary int64 tensor from then on.
    From that moment the materialised tensor is the truth and the packed table is dropped: views handed out from inside
    __torch_dispatch__ do not share a version counter with their base, so edits through them cannot be detected -- the
    engine therefore re-packs a materialised LazyMinhash every time it is given one (the rare path)."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def __new__(cls, packed, pending=None, partial=None):
        return torch.Tensor._make_wrapper_subclass(cls, packed.shape, dtype=torch.int64, device=packed.device, requires_grad=False)

    def __init__(self, packed, pending=None, partial=None):
        """pending: a zero-argument callable that FILLS `packed` (deferred hop, see MinhashPropagation.forward); it is
        run the first time the table is needed -- or never, when HllPropagation computes the table on the way (fused stage).
        partial: optional callable(rows int64 [n]) that fills THOSE rows of `packed` only (knobs.DEFER_TABLE_HOP)"""
        self._packed, self._real, self._pending, self._partial, self._partial_rows = packed, None, pending, partial, 0

    def resolve(self):
        """run the deferred computation of the packed table, if there is one"""
        if self._pending is not None:
            fill, self._pending, self._partial = self._pending, None, None
            fill()

    materialisations = 0  # class-wide count of 8-byte copies made (tests assert that the ELPH call sequence makes none)

    def materialise(self):
        if self._real is None:
            LazyMinhash.materialisations += 1
            self.resolve()
            self._real = unpack_minhash(self._packed)
            self._packed = None
        return self._real

    def packed_for_rows(self, rows):
        """the packed table with at least `rows` (int64 node ids, any shape) computed, for a reader of those rows alone"""
        if self._real is not None:
            return None
        if self._pending is not None and self._partial is not None:
            # ONE row-list launch per table (ELPH's training step: one forward, one batch).  A second reader of the same table
            # -- the reference's inference loop: one forward, many get_subgraph_features batches -- completes it instead: every
            # partial launch also serves ALL hub rows (its hub units), and the pending closure pins the previous hop's table
            if self._partial_rows == 0 and rows.numel() <= self._packed.size(0):
                self._partial_rows = rows.numel()
                self._partial(rows.reshape(-1))
                return self._packed
        self.resolve()
        return self._packed

    def packed_if_valid(self):
        """the packed table while nothing outside the engine has seen (and possibly edited) the int64 form"""
        if self._real is not None:
            return None
        self.resolve()
        return self._packed

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map

        def real(x):
            return x.materialise() if isinstance(x, LazyMinhash) else x
        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))

    # entry points that bypass the dispatcher
    def numpy(self, *args, **kwargs):
        return self.materialise().numpy(*args, **kwargs)

    def tolist(self):
        return self.materialise().tolist()

    def data_ptr(self):
        return self.materialise().data_ptr()

    def __array__(self, *args, **kwargs):
        return self.materialise().__array__(*args, **kwargs)

    def __reduce_ex__(self, proto):
        return self.materialise().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.materialise().clone()


def _tag(t, name, twin):
    """attach a packed twin to a reference-shaped tensor, stamped with the tensor's version counter so that any
    in-place edit invalidates it"""
    try:
        setattr(t, name, (t._version, twin))
    except Exception:  # pragma: no cover
        pass


def _packed_minhash_of(t, device):
    """packed twin of a reference-shaped int64 MinHash tensor (cached on the tensor object)"""
    if isinstance(t, LazyMinhash):
        tw = t.packed_if_valid()
        if tw is not None and tw.device == device:
            return tw
        return pack_minhash(t.materialise(), device)  # never cached: see the class docstring
    tag = getattr(t, '_ss_u32', None)
    if tag is not None and tag[0] == t._version and tag[1].device == device and tag[1].shape == t.shape:
        return tag[1]  # still valid: the tensor has not been edited in place since the twin was made
    if t.dtype == torch.int32:
        tw = t.to(device).contiguous()
    else:
        tw = pack_minhash(t, device)
    _tag(t, '_ss_u32', tw)
    return tw


def _packed_hll_of(t, device):
    tag = getattr(t, '_ss_u8', None)
    if tag is not None and tag[0] == t._version and tag[1].device == device and tag[1].shape == t.shape:
        return tag[1]
    if t.dtype in (torch.int8, torch.uint8):
        tw = t.to(device).contiguous().view(torch.uint8)
    else:
        tw = t.to(device=device, dtype=torch.uint8).contiguous()
    _tag(t, '_ss_u8', tw)
    return tw


