"""Drop-in `src/hashing.py` for a melifluos/subgraph-sketching checkout: the MI355X engine behind the reference's own names.

Copy this file over `<reference checkout>/src/hashing.py` (keep the original as hashing_reference.py if you want to compare).
Every import site of the reference keeps working unchanged:
    from src.hashing import ElphHashes                    (src/models/elph.py:16, src/datasets/elph.py:18)
    from src.hashing import ElphHashes, LABEL_LOOKUP      (test/test_hashing.py:15)
The engine is located through, in this order: an installed / importable `subgraph_sketching_amd`, the environment variable
SUBGRAPH_SKETCH_AMD_ROOT (the root of the engine's repository), or this file's own location when it is used in place
(<engine repo>/integration/src/hashing.py).

Opt-in (environment, read at import): SS_LAZY_FEATURES=1 makes `get_subgraph_features` return a DeviceFeatureStore when it is
called with more than SS_LAZY_MIN_LINKS (default 1 000 000) links -- the BUDDY precompute of datasets/elph.py:207-208 -- so
HashDataset never materialises the [L, h(h+2)] tensor; runners/train.py:58-60 / inference.py:119-120 index it per batch and
get device tensors computed from the resident sketch tables.
"""
import importlib
import os
import sys


def _load_engine():
    try:
        return importlib.import_module('subgraph_sketching_amd')
    except ImportError:
        pass
    here = os.path.dirname(os.path.abspath(__file__))
    for root in (os.environ.get('SUBGRAPH_SKETCH_AMD_ROOT'), os.path.dirname(os.path.dirname(here))):
        if root and os.path.exists(os.path.join(root, 'subgraph_sketching_amd.py')):
            if root not in sys.path:
                sys.path.insert(0, root)
            return importlib.import_module('subgraph_sketching_amd')
    raise ImportError('subgraph_sketching_amd not found: install it, or set SUBGRAPH_SKETCH_AMD_ROOT to the engine repository '
                      '(and build it once with `python __graft_entry__.py`)')


_engine = _load_engine()
_h = _engine.hashing

LABEL_LOOKUP = _h.LABEL_LOOKUP
MinhashPropagation = _h.MinhashPropagation
HllPropagation = _h.HllPropagation
logger = _h.logger

_LAZY = os.environ.get('SS_LAZY_FEATURES', '0') not in ('', '0')
_LAZY_MIN = int(os.environ.get('SS_LAZY_MIN_LINKS', '1000000'))


class ElphHashes(_h.ElphHashes):
    """the engine's ElphHashes under the reference's module path (same constructor arguments, methods and attributes)"""

    def get_subgraph_features(self, links, hash_table, cards, batch_size=11000000, **kwargs):
        if _LAZY and 'lazy' not in kwargs and links.dim() == 2 and links.size(0) >= _LAZY_MIN:
            kwargs['lazy'] = True
        return super().get_subgraph_features(links, hash_table, cards, batch_size, **kwargs)
