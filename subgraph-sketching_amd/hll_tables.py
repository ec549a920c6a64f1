"""HyperLogLog++ constants for precision p: alpha, linear-counting threshold, bias tables.

The reference takes these from the third-party `datasketch` package (reference hashing.py:69-80):
`HyperLogLogPlusPlus(p).alpha / .max_rank`, `hyperloglog_const._thresholds/_bias/_raw_estimate`.
They are INPUTS of the engine:
  * if `datasketch` is importable (true on any machine that can run the reference's runners,
    reference README.md:45) its objects are used verbatim -> provenance "datasketch";
  * otherwise alpha comes from the standard HLL closed form, the thresholds are the ones published in
    the HLL++ paper, and the bias tables are the simulated ones in data/hllpp_tables_regenerated.npz
    (data/make_hllpp_tables.py) -> provenance "regenerated".  Outputs that went through the
    bias-corrected branch are then NOT pinned to the reference's values (parity unpinned).
"""
import os
from collections import namedtuple

import numpy as np

HllTables = namedtuple('HllTables', 'p alpha max_rank threshold raw_estimate bias provenance')

_HERE = os.path.dirname(os.path.abspath(__file__))
REGENERATED = os.path.join(_HERE, 'data', 'hllpp_tables_regenerated.npz')
_THRESHOLDS = (10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000)


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


def load(p, prefer='auto'):
    """prefer: 'auto' (datasketch if importable), 'datasketch', 'regenerated'"""
    if not 4 <= p <= 18:
        raise ValueError(f'hll_p must be in [4, 18], got {p}')
    if prefer in ('auto', 'datasketch'):
        try:
            from datasketch import HyperLogLogPlusPlus, hyperloglog_const
            tmp = HyperLogLogPlusPlus(p=p)
            return HllTables(p, float(tmp.alpha), int(tmp.max_rank), float(hyperloglog_const._thresholds[p - 4]),
                             np.asarray(hyperloglog_const._raw_estimate[p - 4], dtype=np.float64),
                             np.asarray(hyperloglog_const._bias[p - 4], dtype=np.float64), 'datasketch')
        except ImportError:
            if prefer == 'datasketch':
                raise
    with np.load(REGENERATED) as z:
        if not int(z['p_min']) <= p <= int(z['p_max']):
            raise ValueError(f'no regenerated HLL++ table for p={p}; install datasketch or extend '
                             f'data/make_hllpp_tables.py')
        return HllTables(p, hll_alpha(p), 64 - p, float(_THRESHOLDS[p - 4]), z[f'raw_p{p}'].astype(np.float64),
                         z[f'bias_p{p}'].astype(np.float64), 'regenerated')
