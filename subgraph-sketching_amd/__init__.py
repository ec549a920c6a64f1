"""subgraph-sketching_amd -- MI355X (gfx950) engine for the ELPH/BUDDY subgraph-sketching hot path.

Import name: `subgraph_sketching_amd` (see /subgraph_sketching_amd.py; the directory keeps the project's
hyphenated name).  Public surface = the reference's src/hashing.py surface.
"""
from .hashing import (LABEL_LOOKUP, ElphHashes, HllPropagation, HopSketch, MinhashPropagation, SketchTable,
                      build_csr, load_sketches, pack_minhash, save_sketches, unpack_minhash)
from .feature_store import DeviceFeatureStore
from . import _native, hll_tables, knobs, dist, heuristics, sign, roofline

__all__ = ['LABEL_LOOKUP', 'ElphHashes', 'HllPropagation', 'MinhashPropagation', 'SketchTable', 'HopSketch',
           'build_csr', 'DeviceFeatureStore', 'pack_minhash', 'unpack_minhash', 'save_sketches', 'load_sketches', 'hll_tables', 'knobs', 'dist', 'heuristics', 'sign', 'roofline']
