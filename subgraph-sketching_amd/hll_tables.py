"""HyperLogLog++ constants for precision p: alpha, linear-counting threshold, bias tables.

The reference takes these from the third-party `datasketch` package (reference hashing.py:69-80):
`HyperLogLogPlusPlus(p).alpha / .max_rank`, `hyperloglog_const._thresholds/_bias/_raw_estimate`.
They are INPUTS of the engine, looked for in this order:
  1. `datasketch` importable (true on any machine that can run the reference's runners, reference
     README.md:45): its objects are used verbatim                            -> provenance "datasketch";
  2. data/hllpp_tables_datasketch.npz present: the same objects exported once by
     tools/export_datasketch_fixture.py on a machine that has the package     -> provenance "datasketch-export";
  3. otherwise alpha comes from the standard HLL closed form, the thresholds are the ones published in
     the HLL++ paper, and the bias tables are the simulated ones in data/hllpp_tables_regenerated.npz
     (data/make_hllpp_tables.py)                                              -> provenance "regenerated".
     A warning is logged once per precision: outputs that went through the bias-corrected branch are then
     NOT pinned to the reference's values (parity unpinned).
`table_id(tables)` = provenance + a digest of the numbers; it is stamped on every `cards` tensor the engine
produces and stored in the packed cache (hashing.save_sketches), so cardinalities made with one table are never
silently mixed with intersections estimated with another (hashing.get_subgraph_features raises).
"""
import hashlib
import logging
import os
from collections import namedtuple

import numpy as np

logger = logging.getLogger(__name__)

HllTables = namedtuple('HllTables', 'p alpha max_rank threshold raw_estimate bias provenance')

_HERE = os.path.dirname(os.path.abspath(__file__))
REGENERATED = os.path.join(_HERE, 'data', 'hllpp_tables_regenerated.npz')
EXPORTED = os.path.join(_HERE, 'data', 'hllpp_tables_datasketch.npz')
_THRESHOLDS = (10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000)
_warned = set()


def hll_alpha(p):
    """bias constant of the raw HLL estimator (Flajolet et al. 2007), as datasketch computes it"""
    m = 1 << p
    if p == 4:
        return 0.673
    if p == 5:
        return 0.697
    if p == 6:
        return 0.709
    return 0.7213 / (1.0 + 1.079 / m)


def table_id(tables):
    """'<provenance>:<12 hex digits>' -- identifies the numbers, not only where they came from"""
    h = hashlib.sha1()
    h.update(np.asarray([tables.p, tables.max_rank], dtype=np.int64).tobytes())
    h.update(np.asarray([tables.alpha, tables.threshold], dtype=np.float64).tobytes())
    h.update(np.ascontiguousarray(tables.raw_estimate, dtype=np.float64).tobytes())
    h.update(np.ascontiguousarray(tables.bias, dtype=np.float64).tobytes())
    return f'{tables.provenance}:{h.hexdigest()[:12]}'


def same_tables(id_a, id_b):
    """two table ids name the same NUMBERS: only the digest counts -- the tables served by the datasketch package
    ('datasketch:<digest>') and by its shipped export ('datasketch-export:<digest>') are bit-identical, and a cache built
    next to one must load next to the other.  The provenance part is for display."""
    if id_a is None or id_b is None:
        return id_a == id_b
    return str(id_a).rsplit(':', 1)[-1] == str(id_b).rsplit(':', 1)[-1]


def _from_datasketch(p):
    from datasketch import HyperLogLogPlusPlus, hyperloglog_const
    tmp = HyperLogLogPlusPlus(p=p)
    return HllTables(p, float(tmp.alpha), int(tmp.max_rank), float(hyperloglog_const._thresholds[p - 4]),
                     np.asarray(hyperloglog_const._raw_estimate[p - 4], dtype=np.float64),
                     np.asarray(hyperloglog_const._bias[p - 4], dtype=np.float64), 'datasketch')


def _from_export(p, path=None):
    path = path or EXPORTED
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        if f'raw_p{p}' not in z.files:
            return None
        return HllTables(p, float(z[f'alpha_p{p}']), int(z[f'max_rank_p{p}']), float(z[f'threshold_p{p}']),
                         z[f'raw_p{p}'].astype(np.float64), z[f'bias_p{p}'].astype(np.float64), 'datasketch-export')


def load(p, prefer='auto'):
    """prefer: 'auto' (datasketch if importable, else its exported constants if shipped, else regenerated + a warning),
    'datasketch' (the package or its export, else ImportError), 'regenerated' (silently: an explicit choice)"""
    if not 4 <= p <= 18:
        raise ValueError(f'hll_p must be in [4, 18], got {p}')
    if prefer in ('auto', 'datasketch'):
        try:
            return _from_datasketch(p)
        except ImportError:
            exported = _from_export(p)
            if exported is not None:
                return exported
            if prefer == 'datasketch':
                raise
    with np.load(REGENERATED) as z:
        if not int(z['p_min']) <= p <= int(z['p_max']):
            raise ValueError(f'no regenerated HLL++ table for p={p}; install datasketch or extend '
                             f'data/make_hllpp_tables.py')
        tables = HllTables(p, hll_alpha(p), 64 - p, float(_THRESHOLDS[p - 4]), z[f'raw_p{p}'].astype(np.float64),
                           z[f'bias_p{p}'].astype(np.float64), 'regenerated')
    if prefer == 'auto' and p not in _warned:
        _warned.add(p)
        logger.warning('datasketch is not importable and no exported copy of its HLL++ tables is shipped: using REGENERATED '
                       'bias tables for p=%d (%s). Cardinalities / features on the bias-corrected branch will differ from a '
                       'reference run that uses datasketch; do not mix caches built with different tables.', p, table_id(tables))
    return tables
