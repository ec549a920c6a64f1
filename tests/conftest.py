import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope='session')
def regenerated_tables():
    """the HLL++ tables the golden vectors were generated with (always the regenerated ones)"""
    import subgraph_sketching_amd as ssa
    return {p: ssa.hll_tables.load(p, prefer='regenerated') for p in (4, 6, 8, 16)}


def oracle_params(tables, with_lc=True):
    """oracle estimator constants for an HllTables tuple, in the table order the reference saw"""
    from oracle import oracle
    import subgraph_sketching_amd as ssa
    lc = ssa.hashing.linear_counting_table(1 << tables.p).numpy() if with_lc else None
    return oracle.HllParams(tables.p, tables.threshold, tables.raw_estimate, tables.bias, alpha=tables.alpha, lc_table=lc)
