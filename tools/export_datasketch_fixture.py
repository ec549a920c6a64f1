#!/usr/bin/env python3
"""Export datasketch's HLL++ constants (and known answers computed with them) -- run this WHERE `datasketch` IS INSTALLED.

Why: the reference reads `datasketch.hyperloglog_const._thresholds/_bias/_raw_estimate` and
`HyperLogLogPlusPlus(p).alpha/.max_rank` (reference src/hashing.py:69-80).  The package is not vendored, not pinned
(reference README.md:45) and absent from the build / GPU image, so this repository ships simulated ("regenerated") tables
and flags every value on the bias-corrected branch as parity-unpinned.  This tool closes that gap from any machine that
can `pip install datasketch`:

    python tools/export_datasketch_fixture.py            # writes the two files below
    git add subgraph-sketching_amd/data/hllpp_tables_datasketch.npz tests/golden/g11_datasketch_tables.npz

  subgraph-sketching_amd/data/hllpp_tables_datasketch.npz
      alpha / max_rank / threshold / raw / bias for p = 4..16: loaded by hll_tables.load() when the package itself is not
      importable (provenance "datasketch-export").
  tests/golden/g11_datasketch_tables.npz
      sha256 of every table + known answers of the bias-corrected branch computed HERE with plain numpy in float64 from
      datasketch's own tables (6 nearest raw estimates by squared distance, mean bias -- the published HLL++ procedure
      that reference hashing.py:197-210 restates): tests/test_oracle_golden.py::test_datasketch_tables_fixture checks the
      oracle and (on the GPU) the kernels against them, and skips loudly while the file is absent.
Data only: no datasketch source text is copied.
"""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES_OUT = os.path.join(ROOT, 'subgraph-sketching_amd', 'data', 'hllpp_tables_datasketch.npz')
GOLDEN_OUT = os.path.join(ROOT, 'tests', 'golden', 'g11_datasketch_tables.npz')
P_RANGE = range(4, 17)


def write_tables(path, tables):
    """tables: {p: HllTables-like with alpha, max_rank, threshold, raw_estimate, bias}"""
    blob = {'p_list': np.asarray(sorted(tables), dtype=np.int64)}
    for p, t in tables.items():
        blob[f'alpha_p{p}'] = np.float64(t.alpha)
        blob[f'max_rank_p{p}'] = np.int64(t.max_rank)
        blob[f'threshold_p{p}'] = np.float64(t.threshold)
        blob[f'raw_p{p}'] = np.asarray(t.raw_estimate, dtype=np.float64)
        blob[f'bias_p{p}'] = np.asarray(t.bias, dtype=np.float64)
    np.savez_compressed(path, **blob)


def known_answers(t, n=64, seed=11):
    """raw estimates spread over the bias-corrected range of this precision and the corrected values in float64"""
    m = 1 << t.p
    raw, bias = np.asarray(t.raw_estimate, dtype=np.float64), np.asarray(t.bias, dtype=np.float64)
    rng = np.random.RandomState(seed + t.p)
    e = np.sort(rng.uniform(raw.min(), min(5.0 * m, raw.max()), size=n))
    out = np.empty_like(e)
    for i, x in enumerate(e):
        nearest = np.argsort((x - raw) ** 2, kind='stable')[:6]
        out[i] = x - bias[nearest].mean()
    return e, out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--tables-out', default=TABLES_OUT)
    ap.add_argument('--golden-out', default=GOLDEN_OUT)
    a = ap.parse_args()
    try:
        import datasketch
    except ImportError:
        sys.exit('datasketch is not importable here: run this tool on a machine with `pip install datasketch`')
    sys.path.insert(0, ROOT)
    import subgraph_sketching_amd as ssa
    tables, golden = {}, {'datasketch_version': np.asarray(getattr(datasketch, '__version__', 'unknown'))}
    for p in P_RANGE:
        t = ssa.hll_tables._from_datasketch(p)
        tables[p] = t
        golden[f'sha_raw_p{p}'] = np.asarray(hashlib.sha256(np.asarray(t.raw_estimate, dtype=np.float64).tobytes()).hexdigest())
        golden[f'sha_bias_p{p}'] = np.asarray(hashlib.sha256(np.asarray(t.bias, dtype=np.float64).tobytes()).hexdigest())
        golden[f'alpha_p{p}'], golden[f'threshold_p{p}'] = np.float64(t.alpha), np.float64(t.threshold)
        golden[f'table_id_p{p}'] = np.asarray(ssa.hll_tables.table_id(t._replace(provenance='datasketch-export')))
        e, corrected = known_answers(t)
        golden[f'estimate_p{p}'], golden[f'corrected_p{p}'] = e, corrected
    write_tables(a.tables_out, tables)
    np.savez_compressed(a.golden_out, **golden)
    print(f'wrote {a.tables_out} and {a.golden_out} (datasketch {golden["datasketch_version"]})')


if __name__ == '__main__':
    main()
