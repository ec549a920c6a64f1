"""MI355X-native subgraph sketching behind the reference's `ElphHashes` API.

Host-side mirror of /root/reference/src/hashing.py: same class / method / attribute names, argument
meaning and error behaviour, so `from src.hashing import ElphHashes, LABEL_LOOKUP` can be pointed at
this module unchanged.  All sketch arithmetic runs in the hand-written HIP kernels
of csrc/ through the C ABI of include/subgraph_sketch.h -- there is no CPU fallback: without a HIP
device or without the built library every compute entry point raises.

Data layout: the engine keeps sketches "packed" in HBM -- MinHash uint32[N, P] (stored in torch.int32
tensors), HyperLogLog uint8[N, M].  `build_hash_tables` returns a `SketchTable`, a dict
{hop: {'hll': int8[N, M], 'minhash': int64[N, P]}} exactly like the reference's, whose int64 MinHash
leaves are only materialised if somebody reads them (torch.save does); the kernels use the packed twin.
"""
import atexit
import logging
import os
import weakref
from collections import OrderedDict
from collections.abc import Mapping
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from . import _native, hll_tables

logger = logging.getLogger(__name__)
logger.setLevel(logging.INFO)

# reference hashing.py:22-25 -- primary key = max hops, secondary key = feature index, value = (hops from u, hops from v)
LABEL_LOOKUP = {1: {0: (1, 1), 1: (0, 1), 2: (1, 0)},
                2: {0: (1, 1), 1: (2, 1), 2: (1, 2), 3: (2, 2), 4: (0, 1), 5: (1, 0), 6: (0, 2), 7: (2, 0)},
                3: {0: (1, 1), 1: (2, 1), 2: (1, 2), 3: (2, 2), 4: (3, 1), 5: (1, 3), 6: (3, 2), 7: (2, 3), 8: (3, 3),
                    9: (0, 1), 10: (1, 0), 11: (0, 2), 12: (2, 0), 13: (0, 3), 14: (3, 0)}}


# ------------------------------------------------------------------------------------------------
# device plumbing
# ------------------------------------------------------------------------------------------------
KERNEL_TIMER = None  # bench.py installs an object with record(name, stream) / span(name, start, end)


class _Span(object):
    """optional HIP-event bracket around a launch, recorded on the launch stream"""

    def __init__(self, name, device):
        timer = KERNEL_TIMER
        if timer is not None and hasattr(timer, 'wants') and not timer.wants(name):
            timer = None
        self.name, self.device, self.timer = name, device, timer

    def __enter__(self):
        if self.timer is not None:
            self.start = self.timer.record(self.name, torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.span(self.name, self.start, self.timer.record(self.name, torch.cuda.current_stream(self.device)))
        return False


def _compute_device(*tensors):
    """the HIP device the kernels run on: the device of the first GPU tensor, else the current one"""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError('subgraph-sketching_amd needs a HIP device (MI355X): torch.cuda.is_available() is False '
                           'and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _stream(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


_ERROR_FLAGS = {}


def _error_flag(device):
    """one persistent device int32 per GPU that kernels set to 1 on out-of-range ids (never allocated per call)"""
    key = str(device)
    if key not in _ERROR_FLAGS:
        _ERROR_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _ERROR_FLAGS[key]


def _take_error(device):
    """synchronising read-and-clear of the device error flag"""
    flag = _error_flag(device)
    bad = bool(int(flag.item()))
    if bad:
        flag.zero_()
    return bad


_LIVE_DEFERRED = weakref.WeakSet()


@atexit.register
def _warn_unreported_bounds_errors():  # pragma: no cover (interpreter exit)
    try:
        if any(d.unreported() for d in list(_LIVE_DEFERRED)):
            logger.warning('subgraph_sketching_amd: a launch met node ids outside its num_nodes and no later call reported it '
                           '(strict_bounds="deferred"): out-of-range edges were dropped / pairs returned NaN rows. '
                           'Call ElphHashes.check_errors() after the last call, or set strict_bounds = True.')
    except Exception:
        pass


class _DeferredErrors(object):
    """strict_bounds = 'deferred': kernels report out-of-range node ids into a PINNED HOST int32 (hipHostMalloc memory is
    mapped into the device's address space at the same address; the store only happens on an error), which the host reads
    without synchronising: at the next call into the engine, or in ElphHashes.check_errors().  The error therefore
    surfaces late -- like the device-side assert the reference's torch indexing triggers for CUDA tensors -- but a
    build + query step stays free of host round trips."""

    def __init__(self):
        self._flags, self._calls = {}, []
        _LIVE_DEFERRED.add(self)

    def unreported(self):
        """non-waiting look at the report words (for the exit hook: a program whose LAST call had bad ids never comes back to raise)"""
        return any(int(f[0]) for f in self._flags.values())

    def flag(self, device, what):
        key = str(device)
        if key not in self._flags:
            self._flags[key] = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._calls.append(what)
        del self._calls[:-8]
        return self._flags[key]

    def raise_if_set(self, synchronize=False):
        for key, flag in self._flags.items():
            if synchronize:
                torch.cuda.synchronize(torch.device(key))
            if int(flag[0]):
                if not synchronize:  # the word is only cleared once nothing in flight can still write it
                    torch.cuda.synchronize(torch.device(key))
                flag.zero_()
                calls, self._calls = ', '.join(self._calls), []
                raise IndexError(f'an earlier call on this engine was given node ids outside its num_nodes (reported late: '
                                 f'strict_bounds="deferred"); calls since the last clean check: {calls}. Out-of-range edges '
                                 f'were dropped and out-of-range pairs returned NaN rows')
        if synchronize:
            self._calls = []


def _check_sizes(num_perm, p):
    if num_perm <= 0 or num_perm % 4 or num_perm > 2048:
        raise NotImplementedError(f'minhash_num_perm must be a multiple of 4 in [4, 2048], got {num_perm}')
    if not 4 <= p <= 16:
        raise NotImplementedError(f'hll_p must be in [4, 16], got {p}')


class _DeviceParams(object):
    """HLL++ estimator constants resident on one device (struct ss_hll_params + the tensors it points to)"""

    def __init__(self, tables, device):
        p = tables.p
        m = 1 << p
        raw32 = tables.raw_estimate.astype(np.float32)
        order = np.argsort(raw32, kind='stable')
        if not 6 <= len(raw32) <= _native.SS_MAX_TABLE:
            raise ValueError(f'HLL++ bias table must have 6..{_native.SS_MAX_TABLE} entries, got {len(raw32)}')
        self.raw = torch.from_numpy(raw32[order].copy()).to(device)
        self.bias = torch.from_numpy(tables.bias.astype(np.float32)[order].copy()).to(device)
        lc_host = linear_counting_table(m)
        thr32 = np.float32(tables.threshold)
        ok = lc_host[1:].numpy() <= thr32
        if not ok.any() or not np.all(ok[np.argmax(ok):]):
            raise ValueError('linear-counting table is not monotone against the threshold')
        self.lc = lc_host.to(device)
        self.struct = _native.HllParams(p=p, n_tbl=len(raw32), alpha_mm=float(np.float32(tables.alpha * m ** 2)),
                                        threshold=float(thr32), lc_min_zeros=int(np.argmax(ok)) + 1, reserved=0,
                                        raw_est=self.raw.data_ptr(), bias=self.bias.data_ptr(),
                                        lc_table=self.lc.data_ptr())


def linear_counting_table(m):
    """lc[V] = m * log(m / V) for V = 0..m, evaluated by torch on the host in fp32 exactly like the
    reference's `_linearcounting` (hashing.py:194-195) does for an int64 zero count; entry 0 is unused"""
    num_zero = torch.arange(0, m + 1, dtype=torch.int64)
    lc = m * torch.log(m / num_zero)
    lc[0] = float('inf')
    return lc.to(torch.float32)


# ------------------------------------------------------------------------------------------------
# sketch containers
# ------------------------------------------------------------------------------------------------
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
    want = expect.tables_id if isinstance(expect, ElphHashes) else expect
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


# Link sets of at least this many pairs are walked GROUPED BY THEIR FIRST NODE (ss_group_links_by_source + ss_pair_features_grouped):
# BUDDY's precompute hands get_subgraph_features every link of a split (reference datasets/elph.py:207-208) and every source
# occurs many times -- 120 times on average in ogbl-citation2's 356 M links -- so the rows of u are read once per GROUP instead
# of once per pair (they meet in the L1 / L2, or stay in registers).  The whole set is grouped at once (not chunk by chunk: a chunk
# of 11 M links holds a source 3.8 times, the set 120 times); `batch_size` then only bounds the pairs per launch.  Rows are
# bit-identical and in the caller's order.  The grouping costs ~35 ps per link (0.15 ms for 4 M links); SS_GROUP_LINKS_MIN=0 disables.
GROUP_LINKS_MIN = int(os.environ.get('SS_GROUP_LINKS_MIN', str(1 << 20)))
# from this many links on, the grouped query does not walk the order itself: every chunk's links are gathered first and its rows
# scattered afterwards by two streaming kernels (ss_gather_links / ss_scatter_feature_rows): random 16-byte reads and 60-byte
# writes over arrays of gigabytes from inside the query's latency chain cost it more than half its rate (csrc/ss_pairs.hip)
GROUP_GATHER_MIN = int(os.environ.get('SS_GROUP_GATHER_MIN', str(1 << 24)))


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


LAZY_MINHASH = True  # minhash_prop returns its int64 result as a LazyMinhash (materialised on first outside use)
# ELPH.forward (reference models/elph.py:209-212) calls hll_prop then minhash_prop per hop.  With this on, the hop-1 minhash_prop
# (input: an unmodified hop-0 tensor) only RECORDS its work; the hop-2 hll_prop that follows on the same edge_index computes the
# hop-1 MinHash rows together with its own HLL rows in one launch (ss_fused_hop_stage: the VALU-bound first hop under the
# memory-bound table hop).  Anything else that needs the table first (the next minhash_prop, get_subgraph_features, any torch
# operator on the tensor) triggers the ordinary first-hop launch.  Same results either way.
DEFER_FIRST_HOP = os.environ.get('SS_FUSED_STAGE', '1') != '0'
# Deferred table hop: `minhash_prop` on any other input only RECORDS the hop as well.  The next consumer decides how much of it is
# computed: another `minhash_prop` (or any torch operator, torch.save ...) needs the whole table; `get_subgraph_features` reads two
# rows per link, and ELPH's training step (models/elph.py:209-212, runners/train.py:204) queries ONE batch after every full-graph
# propagation -- the rows of that batch are computed through ss_minhash_hop_rows (2 B rows instead of N; same values) and the
# table stays owed for everybody else.  Batches on the same table whose rows add up to more than N make it complete instead.
DEFER_TABLE_HOP = os.environ.get('SS_DEFER_TABLE_HOP', '1') != '0'


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
        partial: optional callable(rows int64 [n]) that fills THOSE rows of `packed` only (DEFER_TABLE_HOP)"""
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
            # partial launch also pays a hub pass over ALL hub rows, and the pending closure pins the previous hop's table
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


# ------------------------------------------------------------------------------------------------
# CSR cache
# ------------------------------------------------------------------------------------------------
# Rows with more in-edges than the hub threshold are propagated by a 16-wave workgroup instead of one wavefront (MinHash) /
# one 16-lane group (HLL).  None = adaptive: a single wavefront walking d neighbour rows takes ~0.35 us * d, which must
# stay well below the whole hop (~E * 40 ps): d <= E / 16384, clamped to [128, 1024].  Measured (power-law endpoints,
# alpha 0.5): collab size 0.754 -> 0.674 ms per step with 144 instead of 512; ppa size flat between 512 and 2048 and 11 %
# slower at 128 (too many rows on the cooperative path).  SS_HUB_THRESHOLD / this constant force a value.
HUB_THRESHOLD = int(os.environ['SS_HUB_THRESHOLD']) if 'SS_HUB_THRESHOLD' in os.environ else None


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
        self.has_hub_rows = True  # unknown (no host read of the device counters): keep the hub passes
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
        hub_threshold = HUB_THRESHOLD if HUB_THRESHOLD is not None else default_hub_threshold(E)
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
    mega_rows = torch.empty((max_mega, 4), dtype=torch.int32, device=device)
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
        # (every unskewed graph) then skips both hub-pass launches of every hop (4 us each)
        host = flags32.cpu()
        if int(host[3]):
            raise IndexError(f'edge_index refers to nodes outside [0, {num_nodes})')
        csr.has_hub_rows = bool(int(host[2]) or int(host[4]))
    return csr


# ELPH.forward builds a fresh self-looped edge_index every step (reference models/elph.py:186): the CSR cache below is keyed on the
# tensor OBJECT, so that step rebuilt an identical CSR every time.  With this on, a cache miss on a tensor of the cached shape goes
# through ss_csr_build_cached: one streaming pass over the edges + a device-side comparison; the build only runs if the edges differ.
REUSE_CSR_BY_CONTENT = os.environ.get('SS_REUSE_CSR', '1') != '0'


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
        # differ, decided on the device (REUSE_CSR_BY_CONTENT; strict builds read their flags back and always rebuild)
        reuse = self._csr if (REUSE_CSR_BY_CONTENT and self._key == key) else None
        csr = build_csr(edge_index, num_nodes, device, check=check, err_flag=err_flag, reuse=reuse, fingerprint=REUSE_CSR_BY_CONTENT)
        if self._hub_hint is not None and not check:  # (ElphHashes._hub_hint: no hub-pass launches for shapes that listed no hub rows)
            hint = self._hub_hint(device, num_nodes, edge_index)
            if hint is not None:
                csr.hub_report, csr.has_hub_rows = hint[0], not hint[1]
        self._ref, self._version, self._key, self._csr = weakref.ref(edge_index), edge_index._version, key, csr
        return csr


_default_csr_cache = _CsrCache()


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
        defer_first_hop / defer_table_hop: None = the module defaults DEFER_FIRST_HOP / DEFER_TABLE_HOP (environment overrides
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
            defer_first = DEFER_FIRST_HOP if self.defer_first_hop is None else self.defer_first_hop
            if (defer_first and LAZY_MINHASH and x.device == device and p == 8 and P % 64 == 0 and P <= 256
                    and x.size(0) * 256 <= ElphHashes.FUSED_STAGE_MAX_TABLE_BYTES):
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
            if (DEFER_TABLE_HOP if self.defer_table_hop is None else self.defer_table_hop) and LAZY_MINHASH and x.device == device:
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
        if LAZY_MINHASH and x.device == device:
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


# ------------------------------------------------------------------------------------------------
# the engine
# ------------------------------------------------------------------------------------------------
class ElphHashes(object):
    """class to store hashes and retrieve subgraph features (mirror of reference hashing.py:48-323)"""
    # largest hop-1 HLL table (bytes) ss_fused_hop_stage is used for.  The first version of the stage lost on tables that do not fit
    # the 256 MiB Infinity Cache (citation2-like: 4.40 against 4.09 ms for the two launches) and was capped there; with LDS landings
    # and batched tail walks it wins there too (3.70 against 3.92 ms), so there is no cap any more.  SS_FUSED_STAGE_MAX_MB:
    # measurement hook
    FUSED_STAGE_MAX_TABLE_BYTES = int(os.environ.get('SS_FUSED_STAGE_MAX_MB', str(1 << 30))) << 20
    HUB_HINT_SHAPES = 256  # graph shapes that get a hub hint word (one pinned int32 each, kept for the engine's lifetime)

    def __init__(self, args, fuse_hop_stage=None, defer_first_hop=None, defer_table_hop=None):
        """args: the reference's namespace (max_hash_hops, floor_sf, minhash_num_perm, hll_p, use_zero_one).  Extensions (keyword
        only in spirit; None = the engine's defaults): fuse_hop_stage -- hop-1 MinHash + hop-2 HLL in one launch inside
        build_hash_tables; defer_first_hop / defer_table_hop -- the deferred launches of the ELPH call sequence (minhash_prop
        records its hop, the next consumer decides how much of it runs).  The environment variables
        SS_FUSED_STAGE / SS_DEFER_TABLE_HOP only set the defaults (tests, A/B measurements)."""
        assert args.max_hash_hops in {1, 2, 3}, f'hashing is not implemented for {args.max_hash_hops} hops'
        self._defer_first_hop, self._defer_table_hop = defer_first_hop, defer_table_hop
        self.max_hops = args.max_hash_hops
        self.floor_sf = args.floor_sf  # if true set minimum sf to 0
        # minhash params (reference hashing.py:58-63)
        self._mersenne_prime = np.uint64((1 << 61) - 1)
        self._max_minhash = np.uint64((1 << 32) - 1)
        self._minhash_range = (1 << 32)
        self.minhash_seed = 1
        self.num_perm = args.minhash_num_perm
        self._csr_cache = _CsrCache(self._bounds, self._prop_hub_hint)
        self.minhash_prop = MinhashPropagation(self._csr_cache, self._report_after_host_copy, defer_first_hop, defer_table_hop)
        # hll params (reference hashing.py:65-81)
        self.p = args.hll_p
        self.m = 1 << self.p
        self.use_zero_one = args.use_zero_one
        self.label_lookup = LABEL_LOOKUP[self.max_hops]
        self.hll_tables = hll_tables.load(self.p)
        self.hll_hashfunc = None  # datasketch's sha1 hashfunc is never used on the path (reference :71)
        self.alpha = self.hll_tables.alpha
        self.max_rank = self.hll_tables.max_rank
        assert self.max_rank == 64 - self.p, 'not using 64 bits for hll++ hashing'
        self.hll_size = self.m
        self.hll_threshold = self.hll_tables.threshold
        self.bias_vector = torch.tensor(self.hll_tables.bias, dtype=torch.float)
        self.estimate_vector = torch.tensor(self.hll_tables.raw_estimate, dtype=torch.float)
        self.hll_prop = HllPropagation(self._csr_cache, self._params, self.m, self._report_after_host_copy)
        self._dev_params = {}
        self._dev_perms = {}
        self.fuse_first_hop = True  # compute hop 1 straight from node ids when the fused kernel supports (num_perm, p)
        # hop-1 MinHash + hop-2 HLL in one launch (ss_fused_hop_stage; num_perm == 128, hll_p == 8, max_hops >= 2, unsharded build)
        self.fuse_hop_stage = (os.environ.get('SS_FUSED_STAGE', '1') != '0') if fuse_hop_stage is None else bool(fuse_hop_stage)
        # node ids outside [0, num_nodes): 'deferred' (default) = IndexError at the NEXT call into this engine or at
        # check_errors(), no host synchronisation inside a step; True = IndexError from the offending call itself (one
        # synchronising 4-byte read per CSR build / query call); False = never reported (edges dropped, NaN feature rows)
        self.strict_bounds = 'deferred'
        self._deferred = _DeferredErrors()
        # link sets of >= GROUP_LINKS_MIN pairs: 'auto' = grouped by their first node unless the list already has its runs (one
        # host read per such call), True = always grouped, False = walked as listed (no host read)
        self.group_links = 'auto'
        # skip the hub-pass launches of build_hash_tables for shapes whose earlier builds listed no hub rows (see _hub_hint)
        self.hub_hints = os.environ.get('SS_HUB_HINTS', '1') != '0'
        self._hub_words, self._hub_arena = {}, None

    # no device handles in pickled state (SURVEY.md section 8(b) threading row)
    def __getstate__(self):
        state = dict(self.__dict__)
        state['_dev_params'], state['_dev_perms'] = {}, {}
        state['_csr_cache'], state['_deferred'] = None, None
        state['_hub_words'], state['_hub_arena'] = {}, None
        state.pop('_tables_id', None)
        state['minhash_prop'], state['hll_prop'] = None, None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault('hub_hints', os.environ.get('SS_HUB_HINTS', '1') != '0')
        self.__dict__.setdefault('group_links', 'auto')
        self._hub_words, self._hub_arena = {}, None
        self._deferred = _DeferredErrors()
        self._csr_cache = _CsrCache(self._bounds, self._prop_hub_hint)
        self.minhash_prop = MinhashPropagation(self._csr_cache, self._report_after_host_copy, self.__dict__.get('_defer_first_hop'),
                                               self.__dict__.get('_defer_table_hop'))
        self.hll_prop = HllPropagation(self._csr_cache, self._params, self.m, self._report_after_host_copy)

    def _report_after_host_copy(self):
        """a result has just been copied to a CPU caller (ELPH on CPU tensors): the launches behind it are complete, so a
        deferred bounds report is final -- raise it from the offending call, as the reference's CPU indexing would"""
        if self.strict_bounds == 'deferred':
            self._deferred.raise_if_set()

    # ---- host-side helpers -------------------------------------------------------------------------
    @property
    def tables_id(self):
        """identity of the HLL++ tables in use (hll_tables.table_id), recomputed if `hll_tables` is replaced"""
        cached = self.__dict__.get('_tables_id')
        if cached is None or cached[0] is not self.hll_tables:
            cached = (self.hll_tables, hll_tables.table_id(self.hll_tables))
            self.__dict__['_tables_id'] = cached
        return cached[1]

    def _params(self, device):
        key = str(device)
        if key not in self._dev_params:
            _check_sizes(self.num_perm, self.p)
            self._dev_params[key] = _DeviceParams(self.hll_tables, device)
        return self._dev_params[key]

    def _np_bit_length(self, bits):
        """number of bits needed to represent each (non-negative) int in `bits` (reference :83-89), computed
        exactly in integer arithmetic"""
        b = np.asarray(bits).astype(np.uint64)
        n = np.zeros(b.shape, dtype=np.int64)
        for s in (32, 16, 8, 4, 2, 1):
            big = b >= (np.uint64(1) << np.uint64(s))
            n = np.where(big, n + s, n)
            b = np.where(big, b >> np.uint64(s), b)
        return (n + (b > 0)).astype(int)

    def _get_hll_rank(self, bits):
        """rank = leading zeros of `bits` seen as a (64 - p)-bit word, plus one (reference :91-104)"""
        rank = self.max_rank - self._np_bit_length(bits) + 1
        if rank.size and rank.min() <= 0:
            raise ValueError("Hash value overflow, maximum size is %d bits" % self.max_rank)
        return rank

    def _init_permutations(self, num_perm):
        """universal-hash parameters (a_j, b_j), j < num_perm, from numpy's legacy RandomState(seed): the draws
        interleave a_0, b_0, a_1, b_1, ... (reference :106-116).  uint64 [2, num_perm]."""
        gen = np.random.RandomState(self.minhash_seed)
        ab = np.empty((2, num_perm), dtype=np.uint64)
        for j in range(num_perm):
            ab[0, j] = gen.randint(1, self._mersenne_prime, dtype=np.uint64)
            ab[1, j] = gen.randint(0, self._mersenne_prime, dtype=np.uint64)
        return ab

    def _perms(self, device):
        key = str(device)
        if key not in self._dev_perms:
            ab = self._init_permutations(self.num_perm).view(np.int64)
            self._dev_perms[key] = torch.from_numpy(ab.copy()).to(device)
        return self._dev_perms[key]

    # ---- hop-0 sketches ------------------------------------------------------------------------------
    def _init_minhash_u32(self, n_nodes, device):
        _check_sizes(self.num_perm, self.p)
        ab = self._perms(device)
        out = torch.empty((n_nodes, self.num_perm), dtype=torch.int32, device=device)
        _native.check(_native.lib().ss_minhash_init(_ptr(out), 0, n_nodes, _ptr(ab[0]), _ptr(ab[1]), self.num_perm,
                                                    _stream(device)), 'ss_minhash_init')
        return out

    def _init_hll_u8(self, n_nodes, device):
        _check_sizes(self.num_perm, self.p)
        out = torch.empty((n_nodes, self.m), dtype=torch.uint8, device=device)
        _native.check(_native.lib().ss_hll_init(_ptr(out), 0, n_nodes, self.p, _stream(device)), 'ss_hll_init')
        return out

    def initialise_minhash(self, n_nodes):
        """int64 [n_nodes, num_perm] hop-0 MinHash rows (reference :118-124); lives on the HIP device"""
        device = _compute_device()
        packed = self._init_minhash_u32(n_nodes, device)
        out = unpack_minhash(packed)
        _tag(out, '_ss_u32', packed)
        # hop-0 marker: lets minhash_prop compute the first hop straight from node ids (ss_first_hop) instead of
        # gathering this table; voided by any in-place edit (version counter)
        out._ss_hop0 = (out._version, self._perms(device), self.p)
        return out

    def initialise_hll(self, n_nodes):
        """int8 [n_nodes, m] hop-0 HLL rows, one non-zero register each (reference :126-137)"""
        device = _compute_device()
        packed = self._init_hll_u8(n_nodes, device)
        out = packed.view(torch.int8)
        _tag(out, '_ss_u8', packed)
        out._ss_hop0 = (out._version, None, self.p)
        return out

    # ---- build ---------------------------------------------------------------------------------------
    def _bounds(self, device, what):
        """-> (check, err_flag) for a launch that validates node ids, after raising what an earlier deferred launch reported"""
        if self.strict_bounds == 'deferred':
            self._deferred.raise_if_set()
            return False, self._deferred.flag(device, what)
        return bool(self.strict_bounds), None

    def _prop_hub_hint(self, device, num_nodes, edge_index):
        """the same hint for the CSR of hll_prop / minhash_prop (the ELPH call sequence)"""
        return self._hub_hint(device, num_nodes, edge_index) if getattr(self, 'hub_hints', False) else None

    def _hub_hint(self, device, num_nodes, edge_index):
        """-> (pinned host word a build of this shape reports its hub + mega row count into, whether an EARLIER build of
        the shape reported none).  The word is read without synchronising -- it holds whatever the most recent COMPLETED build of
        the shape left (-1: none yet) -- and is only a hint: with it, an unskewed graph is built without the two hub-pass launches
        per hop that find nothing to do (9 us of a 0.455 ms step at ogbl-collab size); if the hint is stale (another graph of the
        same shape that does have hub rows) those rows are walked by single wavefronts once -- slow, never wrong -- and the next
        build of the shape has its hub passes back.  `eh.hub_hints = False` keeps the passes unconditionally."""
        key = (str(device), int(num_nodes), tuple(edge_index.shape), HUB_THRESHOLD)
        word = self._hub_words.get(key)
        if word is None:
            # One pinned arena for the engine's lifetime, one word per shape, NEVER handed back while the engine lives: a first-hop
            # launch still in flight stores into its word (system-scope store, csrc report_hub_rows) -- a word returned to torch's
            # pinned-memory cache could be given to somebody else by then.  More shapes than words: no hint for them.
            if self._hub_arena is None:
                self._hub_arena = torch.full((self.HUB_HINT_SHAPES,), -1, dtype=torch.int32).pin_memory()
            if len(self._hub_words) >= self.HUB_HINT_SHAPES:
                return None, False
            word = self._hub_words[key] = self._hub_arena[len(self._hub_words):len(self._hub_words) + 1]
        return word, int(word[0]) == 0

    def check_errors(self):
        """strict_bounds = 'deferred': wait for the launches issued so far and raise IndexError if any of them met a node id
        outside its num_nodes (call once after preprocessing / at the end of an epoch; every call into the engine also
        performs the non-waiting form of this check)"""
        self._deferred.raise_if_set(synchronize=True)

    def build_hash_tables(self, num_nodes, edge_index):
        """k-hop sketches of every node, k = 0..max_hops, and their HLL cardinalities (reference :139-165).
        @return: (SketchTable {k: {'hll','minhash'}}, cards float32 [num_nodes, max_hops])"""
        return self._build(num_nodes, edge_index, None)

    def _build(self, num_nodes, edge_index, shard):
        """shard = None: this process computes every row.  Otherwise (dist.sharded_build_hash_tables) an object with
        `rows` = (begin, end) owned by this rank, `padded_rows` >= num_nodes (allocation size, a multiple of the world
        size) and `gather(tensor) -> handle` / `wait(handle)`: in-place all-gather of the owned row blocks.  The two
        sketches are launched separately so that the gather of one overlaps the kernel of the other."""
        home = edge_index.device
        device = _compute_device(edge_index)
        params = self._params(device)
        # add_self_loops without num_nodes (reference :148): loops for i < max(edge_index) + 1 only; the count is
        # produced on the device by ss_csr_build and read by the propagation kernel -- no host round trip
        check, err_flag = self._bounds(device, f'build_hash_tables(num_nodes={num_nodes})')
        report, no_hubs = (None, False) if (check or not self.hub_hints) else self._hub_hint(device, num_nodes, edge_index)
        csr = build_csr(edge_index, num_nodes, device, check=check, err_flag=err_flag)
        csr.hub_report = report  # (the first-hop kernels of this build leave its hub + mega row count there)
        if no_hubs:  # an earlier build of this shape listed no hub / mega rows: no hub passes (the row kernels walk every row)
            csr.has_hub_rows = False
        csr.use_inferred_self_loops = True
        rows = None if shard is None else shard.rows
        n_alloc = num_nodes if shard is None else shard.padded_rows
        cards = torch.empty((n_alloc, self.max_hops), dtype=torch.float32, device=device)
        table = SketchTable()
        h = self.max_hops
        # hop 1 from node ids: MinHash for 64 / 128 / 192 / 256 permutations at any hll_p, HLL at hll_p == 8
        fused_mh = self.fuse_first_hop and self.num_perm % 64 == 0 and self.num_perm <= 256
        fused = fused_mh and self.p == 8
        peer = shard is not None and getattr(shard, 'peer_write', False)
        if peer:  # the shard's persistent, IPC-shared tables (every rank's launches store into every rank's copy)
            mh, hll, cards = shard.tables(h, self.num_perm, self.m, device)
        else:
            mh = [torch.empty((n_alloc, self.num_perm), dtype=torch.int32, device=device) for _ in range(h)]
            hll = [torch.empty((n_alloc, self.m), dtype=torch.uint8, device=device) for _ in range(h)]
        if fused:
            # hop 1 is computed straight from node ids (ss_first_hop); the hop-0 tables (pure functions of the node id,
            # never read by get_subgraph_features) are produced only if a caller actually looks at them
            table[0] = HopSketch(None, None, home, make_packed=lambda n=num_nodes, d=device: (self._init_minhash_u32(n, d),
                                                                                            self._init_hll_u8(n, d)))
            mh_prev = hll_prev = None
        elif fused_mh:
            hll_prev = self._init_hll_u8(num_nodes, device)
            table[0] = HopSketch(None, None, home, make_packed=lambda n=num_nodes, d=device, l=hll_prev: (self._init_minhash_u32(n, d), l))
            mh_prev = None
        else:
            mh_prev = self._init_minhash_u32(num_nodes, device)  # hop 0 is replicated: a pure function of the node id
            hll_prev = self._init_hll_u8(num_nodes, device)
            table[0] = HopSketch(mh_prev, hll_prev, home)
        # (collab-like -4 %, ppa-like -5 %, citation2-like -1.2 % per step against the unfused schedule)
        if (shard is None and fused and h >= 2 and self.num_perm == 128 and self.fuse_hop_stage
                and num_nodes * self.m <= self.FUSED_STAGE_MAX_TABLE_BYTES):
            # ONE call for hops 1 and 2: hop-1 HLL first (hop-2 HLL rows need that table complete), then hop-1 MinHash + hop-2 HLL
            # in one launch (the VALU-bound first hop and the memory-bound table hop interleaved inside every wavefront,
            # csrc/ss_fused_hop.hip), then hop-2 MinHash; further hops unfused
            ab = self._perms(device)
            graph = csr.struct()
            with _Span('fused_hop_stage', device):
                _native.check(_native.lib().ss_fused_hop_stage(byref(graph), _ptr(ab[0]), _ptr(ab[1]), self.num_perm, _ptr(mh[0]), _ptr(mh[1]),
                                                               self.p, _ptr(hll[0]), _ptr(cards), _ptr(hll[1]), _ptr(cards[:, 1]), h,
                                                               byref(params.struct), _stream(device)), 'ss_fused_hop_stage')
            for k in range(3, h + 1):
                _propagate(csr, mh[k - 2], hll[k - 2], device, cards_out=cards[:, k - 1], cards_stride=h, params=params,
                           mh_out=mh[k - 1], hll_out=hll[k - 1])
        elif shard is None:
            # (inside the library each of these calls is one launch per sketch + one hub pass: measured faster than
            # two-sketch kernels -- first hop 37 + 134 us vs 184, table hop 111 + 192 us vs 326 on the bench graph)
            for k in range(1, h + 1):
                if k == 1 and fused:
                    self._first_hop(csr, device, mh[0], hll[0], cards, params)
                elif k == 1 and fused_mh:
                    self._first_hop(csr, device, mh[0], None, None, params)
                    _propagate(csr, None, hll_prev, device, cards_out=cards[:, 0], cards_stride=h, params=params, hll_out=hll[0])
                else:
                    logger.info(f"Calculating hop {k} hashes")
                    _propagate(csr, mh_prev, hll_prev, device, cards_out=cards[:, k - 1], cards_stride=h, params=params,
                               mh_out=mh[k - 1], hll_out=hll[k - 1])
                mh_prev, hll_prev = mh[k - 1], hll[k - 1]
        elif peer:
            # peer-write: no exchange step -- the kernels store every finished row into all ranks' tables while they run
            # (csrc: mirror_* stores); a hop may start once EVERY rank's launches of the previous hop are complete
            if not (fused_mh and fused):
                raise NotImplementedError('the peer-write build is built for the default sketch shape (128 permutations, hll_p = 8)')
            for k in range(1, h + 1):
                mir_mh, mir_hll = shard.mirrors('mh', k - 1), shard.mirrors('hll', k - 1)
                none = [0] * len(mir_mh)
                if k == 1:
                    self._first_hop(csr, device, mh[0], None, None, params, rows=rows, mirrors=(mir_mh, none, none))
                    self._first_hop(csr, device, None, hll[0], cards, params, rows=rows, mirrors=(none, mir_hll, shard.mirrors('cards', 0)))
                else:
                    _propagate(csr, mh[k - 2], None, device, mh_out=mh[k - 1], rows=rows, mirrors=(mir_mh, none, none))
                    _propagate(csr, None, hll[k - 2], device, cards_out=cards[:, k - 1], cards_stride=h, params=params, hll_out=hll[k - 1],
                               rows=rows, mirrors=(none, mir_hll, shard.mirrors('cards', k - 1)))
                shard.hop_barrier()
            cards = cards[:num_nodes]
        else:
            pending_mh = pending_hll = None
            for k in range(1, h + 1):
                shard.wait(pending_mh)  # hop k-1 MinHash rows of every rank have arrived
                if k == 1 and fused_mh:
                    self._first_hop(csr, device, mh[0], None, None, params, rows=rows)
                else:
                    _propagate(csr, mh_prev, None, device, mh_out=mh[k - 1], rows=rows)
                pending_mh = shard.gather(mh[k - 1])
                shard.wait(pending_hll)
                if k == 1 and fused:
                    self._first_hop(csr, device, None, hll[0], cards, params, rows=rows)
                else:
                    _propagate(csr, None, hll_prev, device, cards_out=cards[:, k - 1], cards_stride=h, params=params,
                               hll_out=hll[k - 1], rows=rows)
                pending_hll = shard.gather(hll[k - 1])
                mh_prev, hll_prev = mh[k - 1], hll[k - 1]
            shard.wait(pending_mh)
            shard.wait(pending_hll)
            shard.wait(shard.gather(cards))
            cards = cards[:num_nodes]
        for k in range(1, h + 1):
            table[k] = HopSketch(mh[k - 1][:num_nodes], hll[k - 1][:num_nodes], home)
        if home != device:
            cards = cards.to(home)
            if self.strict_bounds == 'deferred':  # (the copy has waited for the build: the report is final, see get_subgraph_features)
                self._deferred.raise_if_set()
        return table, _stamp_tables(cards, self.tables_id)

    def _first_hop(self, csr, device, mh_out, hll_out, cards, params, rows=None, mirrors=None):
        """fused hop-0 + hop-1 (ss_first_hop) for either or both sketches"""
        ab = self._perms(device)
        graph = csr.struct(rows, mirrors)
        with _Span('first_hop_mh' if hll_out is None else ('first_hop_hll' if mh_out is None else 'first_hop'), device):
            _native.check(_native.lib().ss_first_hop(byref(graph), _ptr(ab[0]), _ptr(ab[1]), self.num_perm, _ptr(mh_out), self.p,
                                                     _ptr(hll_out), _ptr(cards) if hll_out is not None else None, self.max_hops,
                                                     byref(params.struct), _stream(device)), 'ss_first_hop')

    # ---- query ---------------------------------------------------------------------------------------
    def _resolve_tables(self, hash_table, device, rows=None):
        """packed tables of hops 1 .. max_hops; rows: the node ids the caller is going to read (a deferred LAST hop -- nobody
        else's input -- is then computed for those rows only, DEFER_TABLE_HOP)"""
        mh, hll = [], []
        for k in range(1, self.max_hops + 1):
            entry = hash_table[k]
            if isinstance(entry, HopSketch) and entry.mh_u32.device == device:
                m, l = entry.packed(device)
                mh.append(m)
                hll.append(l)
            else:
                t = entry['minhash']
                tw = None
                if rows is not None and k == self.max_hops and isinstance(t, LazyMinhash) and t.device == device:
                    tw = t.packed_for_rows(rows)
                mh.append(tw if tw is not None else _packed_minhash_of(t, device))
                hll.append(_packed_hll_of(entry['hll'], device))
        N, P = mh[0].shape
        for a, b in zip(mh, hll):
            if a.shape != (N, P) or b.shape != (N, self.m):
                raise ValueError('hash tables of different hops must have the same shape')
        return mh, hll, N, P

    def _pair_kernel(self, links, hash_table, cards, want_debug=False, degrees=None, floor_sf=None, group_batch=None):
        """runs ss_pair_features for links [B,2]; returns (features [B,nf] (or [B,2nf] with degrees) on device, debug dict or None).
        group_batch: the links are first grouped by their first node and walked in that order, `group_batch` pairs per launch
        (GROUP_LINKS_MIN; same rows, in the caller's order)"""
        # where the links live, else where the packed tables already are, else cards, else the current device
        first = hash_table.get(1) if hasattr(hash_table, 'get') else None
        device = _compute_device(links, first.mh_u32 if isinstance(first, HopSketch) else None, cards)
        params = self._params(device)
        lk = links.to(device=device, dtype=torch.int64).contiguous()
        mh, hll, N, P = self._resolve_tables(hash_table, device, rows=lk)
        h = self.max_hops
        B = lk.size(0)
        if cards is None:
            cd = torch.zeros((N, h), dtype=torch.float32, device=device)
        else:
            made_with = getattr(cards, '_ss_tables', None)
            if made_with is not None and not hll_tables.same_tables(made_with, self.tables_id):
                raise ValueError(f'cards were estimated with HLL++ tables {made_with}, this engine uses {self.tables_id}: '
                                 f'a feature row would mix two bias tables (rebuild the cache or load the same tables)')
            # ELPH keeps `cards` on the CPU and the reference re-uploads it on every call (hashing.py:274): keep a device
            # twin on the tensor, invalidated by in-place edits, so repeated eval batches do not pay the copy again
            tag = getattr(cards, '_ss_cards', None)
            if cards.device == device and cards.dtype == torch.float32:
                cd = cards
            elif tag is not None and tag[0] == cards._version and tag[1].device == device:
                cd = tag[1]
            else:
                cd = cards.to(device=device, dtype=torch.float32)
                _tag(cards, '_ss_cards', cd)
            if cd.dim() != 2 or cd.size(0) != N or cd.size(1) < h:
                raise ValueError(f'cards must have shape [{N}, >= {h}], got {tuple(cd.shape)}')
            if cd.stride(1) != 1:
                cd = cd.contiguous()
        nf = h * (h + 2)
        mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh])
        hll_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hll])
        floor = self.floor_sf if floor_sf is None else floor_sf  # DeviceFeatureStore records HashDataset's post-hoc floor
        flags = (_native.SS_FLAG_USE_ZERO_ONE if self.use_zero_one else 0) | (_native.SS_FLAG_FLOOR_SF if floor else 0)
        strict, err = self._bounds(device, f'get_subgraph_features({B} links, num_nodes={N})')
        if strict:
            err = _error_flag(device)  # (non-strict launches never touch the shared flag)
        dg = None
        if degrees is not None:
            dg = degrees.to(device=device, dtype=torch.float32).contiguous()
            if dg.dim() != 1 or dg.numel() != N:
                raise ValueError(f'degrees must have shape [{N}], got {tuple(dg.shape)}')
        if group_batch and not want_debug and 1 < B < (1 << 31) and N < (1 << 31):
            # a list that already has its runs (a coalesced edge list, an evaluation set listing every source's negatives together)
            # is walked as it is: grouping it again costs ~8 % and scatters the output rows.  One small reduction + ONE host read
            # per call of >= GROUP_LINKS_MIN links (0.5 ms of query and more); ElphHashes.group_links = True / False skips it.
            mode = getattr(self, 'group_links', 'auto')  # (instances pickled before the attribute existed)
            if mode == 'auto':
                mode = float((lk[1:, 0] == lk[:-1, 0]).sum().item()) < 0.5 * (B - 1)
            order = group_links_by_source(lk, N, device) if mode else None
            nf_out = 2 * nf if dg is not None else nf
            out = torch.empty((B, nf_out), dtype=torch.float32, device=device)
            for s0 in range(0, B, group_batch):
                nb = min(group_batch, B - s0)
                lib = _native.lib()
                with _Span('pair_features', device):
                    if order is not None and B >= GROUP_GATHER_MIN:
                        # a set of gigabytes: gather the chunk's links, query the (now contiguous, grouped) chunk, scatter its rows
                        o = c_void_p(order.data_ptr() + 4 * s0)
                        lk_c = torch.empty((nb, 2), dtype=torch.int64, device=device)
                        rows_c = torch.empty((nb, nf_out), dtype=torch.float32, device=device)
                        _native.check(lib.ss_gather_links(_ptr(lk), o, nb, _ptr(lk_c), _stream(device)), 'ss_gather_links')
                        _native.check(lib.ss_pair_features_grouped(_ptr(lk_c), None, nb, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0),
                                                                   byref(params.struct), flags, _ptr(dg), _ptr(rows_c), _ptr(err),
                                                                   _stream(device)), 'ss_pair_features_grouped')
                        _native.check(lib.ss_scatter_feature_rows(_ptr(rows_c), o, nb, nf_out, _ptr(out), _stream(device)), 'ss_scatter_feature_rows')
                        continue
                    if order is not None:  # every launch writes rows out[order[s0 + t]] of the ONE output tensor
                        args = (_ptr(lk), c_void_p(order.data_ptr() + 4 * s0), nb, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0),
                                byref(params.struct), flags, _ptr(dg), _ptr(out), _ptr(err), _stream(device))
                    else:                  # as listed: a slice of the links and the matching slice of the output
                        args = (c_void_p(lk.data_ptr() + 16 * s0), None, nb, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0),
                                byref(params.struct), flags, _ptr(dg), c_void_p(out.data_ptr() + 4 * nf_out * s0), _ptr(err), _stream(device))
                    _native.check(lib.ss_pair_features_grouped(*args), 'ss_pair_features_grouped')
            if strict and _take_error(device):
                raise IndexError(f'links refer to nodes outside [-{N}, {N})')
            return out, None
        if degrees is not None:
            out = torch.empty((B, 2 * nf), dtype=torch.float32, device=device)
            with _Span('pair_features', device):
                _native.check(_native.lib().ss_pair_features_normalised(
                    _ptr(lk), B, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0), byref(params.struct), flags, _ptr(dg),
                    _ptr(out), _ptr(err), _stream(device)), 'ss_pair_features_normalised')
            if strict and B > 0 and _take_error(device):
                raise IndexError(f'links refer to nodes outside [-{N}, {N})')
            return out, None
        out = torch.empty((B, nf), dtype=torch.float32, device=device)
        dbg = None
        if want_debug:
            dbg = {'match': torch.empty((B, h, h), dtype=torch.int32, device=device),
                   'zeros': torch.empty((B, h, h), dtype=torch.int32, device=device),
                   'inter': torch.empty((B, h, h), dtype=torch.float32, device=device)}
        with _Span('pair_features', device):
            _native.check(_native.lib().ss_pair_features(
                _ptr(lk), B, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0), byref(params.struct), flags, _ptr(out),
                _ptr(dbg['match']) if dbg else None, _ptr(dbg['zeros']) if dbg else None,
                _ptr(dbg['inter']) if dbg else None, _ptr(err), _stream(device)), 'ss_pair_features')
        if strict and B > 0 and _take_error(device):
            raise IndexError(f'links refer to nodes outside [-{N}, {N})')
        return out, dbg

    def _get_intersections(self, edge_list, hash_table):
        """set-intersection estimates jaccard * union for every (k1, k2) (reference :167-189).
        @return: {(k1, k2): float32 [n_edges]} on edge_list.device"""
        _, dbg = self._pair_kernel(edge_list, hash_table, None, want_debug=True)
        inter = dbg['inter'].to(edge_list.device)
        return {(k1, k2): inter[:, k1 - 1, k2 - 1].contiguous()
                for k1 in range(1, self.max_hops + 1) for k2 in range(1, self.max_hops + 1)}

    def get_hashval(self, x):
        return x.hashvals

    def _linearcounting(self, num_zero):
        return self.m * torch.log(self.m / num_zero)

    def _estimate_bias_or_refine(self, e, refine):
        device = _compute_device(e)
        params = self._params(device)
        x = e.to(device=device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        _native.check(_native.lib().ss_estimate_bias(_ptr(x), x.numel(), byref(params.struct), _ptr(out), int(refine),
                                                     _stream(device)), 'ss_estimate_bias')
        return out.to(e.device)

    def _estimate_bias(self, e):
        """mean bias of the 6 table entries nearest to each estimate (reference :197-204)"""
        return self._estimate_bias_or_refine(e, False)

    def _refine_hll_count_estimate(self, estimate):
        """subtract the bias from estimates <= 5m, in place like the reference (:206-210)"""
        refined = self._estimate_bias_or_refine(estimate, True)
        estimate.copy_(refined)
        return estimate

    def hll_count(self, regs):
        """HLL++ cardinality estimate of each register row (reference :212-232).
        @param regs: integer tensor [n, m] (or [m])  @return: float32 [n] on regs.device"""
        if regs.dim() == 1:
            regs = regs.unsqueeze(dim=0)
        if regs.size(1) != self.m:
            raise ValueError(f'expected rows of {self.m} registers, got {regs.size(1)}')
        device = _compute_device(regs)
        tag = getattr(regs, '_ss_count', None)
        if tag is not None:  # produced together with `regs` by hll_prop (same kernel arithmetic); handed out once
            regs._ss_count = None
            if tag[0] == regs._version and tag[1].device == device and tag[1].numel() == regs.size(0):
                return tag[1] if regs.device == device else tag[1].to(regs.device)
        params = self._params(device)
        packed = _packed_hll_of(regs, device)
        out = torch.empty(regs.size(0), dtype=torch.float32, device=device)
        _native.check(_native.lib().ss_hll_count(_ptr(packed), regs.size(0), byref(params.struct), _ptr(out), 1,
                                                 _stream(device)), 'ss_hll_count')
        return out if regs.device == device else out.to(regs.device)

    def _hll_merge(self, src, dst):
        if src.shape != dst.shape:
            raise ValueError('source and destination register shapes must be the same')
        return torch.maximum(src, dst)

    def hll_neighbour_merge(self, root, neighbours):
        all_regs = torch.cat([root.unsqueeze(dim=0), neighbours], dim=0)
        return torch.max(all_regs, dim=0)[0]

    def minhash_neighbour_merge(self, root, neighbours):
        all_regs = torch.cat([root.unsqueeze(dim=0), neighbours], dim=0)
        return torch.min(all_regs, dim=0)[0]

    def jaccard(self, src, dst):
        """minhash Jaccard estimate of [n_edges, num_perm] hash-value tensors (reference :247-256)"""
        if src.shape != dst.shape:
            raise ValueError('source and destination hash value shapes must be the same')
        return torch.count_nonzero(src == dst, dim=-1) / self.num_perm

    def get_subgraph_features(self, links, hash_table, cards, batch_size=11000000, degrees=None, lazy=False):
        """structural features of node pairs: approximations of the number of nodes at distance (d_u, d_v)
        from (u, v), for the (d_u, d_v) listed in LABEL_LOOKUP[max_hops] (reference :258-323).
        @param links: int tensor [n_edges, 2] (or [2])
        @param hash_table: {hop: {'hll': [N, m], 'minhash': [N, num_perm]}} (a SketchTable or plain tensors)
        @param cards: float tensor [N, max_hops] of neighbourhood cardinality estimates
        @param batch_size: pairs per kernel launch (results do not depend on it)
        @param degrees: optional float tensor [N] (HashDataset.degrees, datasets/elph.py:74).  Extension beyond the
               reference signature: when given, BUDDY's degree-normalised copy (models/elph.py:276-293: feature /
               sqrt(d_u * d_v), NaN / Inf -> 0) is appended in the same kernel and the result is [n_edges, 2 * F].
        @param lazy: extension: return a `DeviceFeatureStore` (feature_store.py) instead of the tensor -- rows are computed on
               the GPU when a batch indexes it (runners/train.py:58-60), nothing of size [n_edges, F] is materialised
        @return: float32 [n_edges, max_hops * (max_hops + 2)] on links.device"""
        if self.max_hops not in (1, 2, 3):
            raise NotImplementedError("Only 1, 2 and 3 hop hashes are implemented")
        if links.dim() == 1:
            links = links.unsqueeze(0)
        if lazy:
            from .feature_store import DeviceFeatureStore
            return DeviceFeatureStore(self, links, hash_table, cards, degrees=degrees, batch_size=batch_size)
        n = links.size(0)
        if GROUP_LINKS_MIN and n >= GROUP_LINKS_MIN and n < (1 << 31):
            feats, _ = self._pair_kernel(links, hash_table, cards, degrees=degrees, group_batch=max(int(batch_size), 1))
        elif n <= batch_size:
            feats, _ = self._pair_kernel(links, hash_table, cards, degrees=degrees)
        else:
            chunks = [self._pair_kernel(links[s:s + batch_size], hash_table, cards, degrees=degrees)[0]
                      for s in range(0, n, batch_size)]
            feats = torch.cat(chunks, dim=0)
        if feats.device == links.device:
            return feats
        out = feats.to(links.device)
        # links on another device (BUDDY keeps them on the CPU): the copy back has waited for the launches, so the deferred
        # bounds report is final and can be raised from the offending call itself -- as the reference's CPU indexing does
        if self.strict_bounds == 'deferred':
            self._deferred.raise_if_set()
        return out
