"""The C-ABI library loads on a GPU-less host and exports every symbol include/subgraph_sketch.h (the drop-in boundary)
and include/subgraph_sketch_debug.h (measurement-only probes) declare (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols(headers=('subgraph_sketch.h', 'subgraph_sketch_debug.h')):
    names = set()
    for header in headers:
        text = open(os.path.join(REPO, 'include', header)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        names |= set(re.findall(r'\b(ss_[a-z0-9_]+)\s*\(', text))
    return sorted(names)


def test_header_declares_the_expected_boundary():
    names = _declared_symbols(('subgraph_sketch.h',))
    assert not [n for n in names if n.startswith(('ss_time_', 'ss_profile_'))], 'probes belong in subgraph_sketch_debug.h'
    for needed in ('ss_minhash_init', 'ss_hll_init', 'ss_csr_build', 'ss_propagate', 'ss_hll_count', 'ss_pair_features',
                   'ss_pack_minhash', 'ss_unpack_minhash', 'ss_estimate_bias', 'ss_version', 'ss_error_string'):
        assert needed in names


def test_library_exports_every_declared_symbol():
    import subgraph_sketching_amd as ssa
    path = ssa._native.LIB_PATH
    assert os.path.exists(path), 'run `python __graft_entry__.py` first (build())'
    handle = ctypes.CDLL(path)
    declared = _declared_symbols()
    missing = [n for n in declared if not hasattr(handle, n)]
    assert not missing, f'symbols declared in the header but not exported: {missing}'
    unbound = [n for n in declared if n not in ssa._native.SIGNATURES]
    assert not unbound, f'symbols declared in the header but not bound in _native.SIGNATURES: {unbound}'
    extra = [n for n in ssa._native.SIGNATURES if n not in declared]
    assert not extra, f'bound but not declared: {extra}'


def test_version_and_error_strings():
    import subgraph_sketching_amd as ssa
    lib = ssa._native.lib()
    assert lib.ss_version() == 129 == ssa._native.ABI_VERSION
    assert lib.ss_error_string(0) == b'ok'
    assert b'invalid' in lib.ss_error_string(-1)
    assert lib.ss_csr_workspace_bytes(1000, 5000) >= 8 * 1001


def test_profile_probe_bookkeeping_without_a_gpu():
    """the tagged span list: nothing recorded -> zero launches; bad tags are argument errors"""
    from ctypes import byref, c_float, c_int32
    import subgraph_sketching_amd as ssa
    lib = ssa._native.lib()
    ms, n = c_float(-1.0), c_int32(-1)
    assert lib.ss_profile_enable(0) == 0
    assert lib.ss_profile_read(ssa._native.PROF_MINHASH_HOP, byref(ms), byref(n)) == 0 and n.value == 0 and ms.value == 0.0
    assert lib.ss_profile_read(99, byref(ms), byref(n)) == -1


def test_argument_errors_are_reported_without_a_gpu():
    """argument validation happens on the host before any launch"""
    import subgraph_sketching_amd as ssa
    lib = ssa._native.lib()
    assert lib.ss_minhash_init(None, 0, 10, None, None, 128, None) == -1      # null pointers
    assert lib.ss_minhash_init(None, 0, 0, None, None, 128, None) == 0        # empty is fine
    assert lib.ss_hll_init(None, 0, 10, 3, None) == -1                        # p < 4
    assert lib.ss_pack_minhash(None, None, -1, None) == -1
    assert lib.ss_fused_hop_stage(None, None, None, 128, None, None, 8, None, None, None, None, 0, None, None) == -1   # no graph
    assert lib.ss_pair_features(None, 0, 0, 4, None, 128, None, None, 0, None, 0, None, None, None, None, None, None) == -4  # h = 4
    assert lib.ss_minhash_hop_rows(None, None, None, 128, None, 0, None) == -1                                                 # no graph
    assert lib.ss_first_hop(None, None, None, 128, None, 8, None, None, 0, None, None) == -1
    assert lib.ss_propagate(None, None, None, 128, None, None, 256, None, 0, None, None) == -1
    g = ssa._native.CsrGraphStruct(rowptr=8, col=8, num_nodes=4, n_self_loops=0, n_self_loops_dev=None, hub_threshold=512, reserved=0,
                                   hub_rows=None, hub_count=None, mega_rows=None, mega_count=None, mega_scratch=None, row_begin=0, row_end=0)
    from ctypes import byref, c_void_p
    fake = c_void_p(8)   # never dereferenced: every call below is rejected by the host-side checks
    assert lib.ss_minhash_hop_rows(byref(g), fake, fake, 130, fake, 4, None) == -1      # P not a multiple of 4
    assert lib.ss_minhash_hop_rows(byref(g), fake, fake, 128, None, 4, None) == -1      # rows missing
    assert lib.ss_minhash_hop_rows(byref(g), fake, fake, 128, fake, -1, None) == -1
    assert lib.ss_minhash_hop_rows(byref(g), fake, fake, 128, None, 0, None) == 0       # empty list: nothing to do
    g.row_begin, g.row_end = 1, 3
    assert lib.ss_minhash_hop_rows(byref(g), fake, fake, 128, fake, 4, None) == -1      # a row list and a row range exclude each other
    assert lib.ss_first_hop(byref(g), fake, fake, 100, fake, 8, None, None, 0, None, None) == -4     # no first-hop kernel for P = 100
    assert lib.ss_first_hop(byref(g), fake, fake, 128, None, 6, fake, None, 0, None, None) == -4     # HLL first hop is p = 8 only
    g.row_begin, g.row_end = 3, 1
    assert lib.ss_propagate(byref(g), fake, fake, 128, None, None, 256, None, 0, None, None) == -1   # empty / reversed row range


def test_missing_library_fails_loudly(monkeypatch):
    import subgraph_sketching_amd as ssa
    monkeypatch.setattr(ssa._native, '_lib', None)
    monkeypatch.setattr(ssa._native, 'LIB_PATH', '/nonexistent/libsubgraph_sketch.so')
    with pytest.raises(ssa._native.NativeLibraryMissing):
        ssa._native.lib()
