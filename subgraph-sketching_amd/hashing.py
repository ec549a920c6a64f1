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
# The implementation lives in five modules (round 4: this file held all of it, 1 470 lines): _runtime (device plumbing, deferred
# bounds errors), containers (HopSketch / LazyMinhash / packed format), csr (CsrGraph, build_csr, caches, link grouping),
# propagation (MinhashPropagation / HllPropagation), engine (ElphHashes, LABEL_LOOKUP); knobs holds the switches.  This module is
# the reference-shaped front door: everything `src/hashing.py` exports, under the same names, plus what the tests reach for.
from ctypes import byref  # noqa: F401  (tests build argument structs through this module)

from . import knobs
from ._runtime import (_DeferredErrors, _DeviceParams, _Span, _check_sizes, _compute_device, _error_flag, _ptr, _stream, _take_error,  # noqa: F401
                       linear_counting_table, logger)
from .containers import (PACKED_FORMAT, HopSketch, LazyMinhash, SketchTable, _packed_hll_of, _packed_minhash_of, _stamp_tables, _tag,  # noqa: F401
                         load_sketches, pack_minhash, save_sketches, unpack_minhash)
from .csr import CsrGraph, _CsrCache, _default_csr_cache, build_csr, default_hub_threshold, group_links_by_source  # noqa: F401
from .propagation import HllPropagation, MinhashPropagation, _first_hop_from_ids, _hop0_marker, _propagate  # noqa: F401
from .engine import LABEL_LOOKUP, ElphHashes  # noqa: F401

_KNOBS = ('KERNEL_TIMER', 'GROUP_LINKS_MIN', 'GROUP_GATHER_MIN', 'LAZY_MINHASH', 'DEFER_FIRST_HOP', 'DEFER_TABLE_HOP', 'HUB_THRESHOLD',
          'REUSE_CSR_BY_CONTENT', 'FUSED_STAGE_MAX_TABLE_BYTES')


def __getattr__(name):  # hashing.DEFER_TABLE_HOP etc. read the live value in knobs
    if name in _KNOBS:
        return getattr(knobs, name)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')


class _ForwardingModule(type(knobs)):
    """`hashing.LAZY_MINHASH = False` (how rounds 1-3 documented the switches) must keep working after the split: a write to one of
    the former hashing.* switches goes to knobs, where the engine reads it -- not into a shadow attribute nobody looks at"""

    def __setattr__(self, name, value):
        if name in _KNOBS:
            setattr(knobs, name, value)
        else:
            super().__setattr__(name, value)


import sys as _sys  # noqa: E402

_sys.modules[__name__].__class__ = _ForwardingModule
