"""ctypes binding of the C-ABI library (include/subgraph_sketch.h).

The HIP library is the product: there is no CPU fallback.  `lib()` raises loudly when the shared
object has not been built (run `python __graft_entry__.py` or `subgraph-sketching_amd/csrc/build.sh`).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SS_LIB: measurement hook -- tools/ablate_fused.sh loads deliberately incomplete builds of the library to time what is left)
LIB_PATH = os.environ.get('SS_LIB') or os.path.join(_HERE, 'libsubgraph_sketch.so')

SS_MAX_HOPS = 3
SS_MAX_TABLE = 512
SS_FLAG_USE_ZERO_ONE = 1
SS_FLAG_FLOOR_SF = 2
SS_CSR_ERR_BOUNDS, SS_CSR_ERR_PROTOCOL = 1, 2  # bits of a CSR build's err_flag


class HllParams(ctypes.Structure):
    """mirror of `struct ss_hll_params`"""
    _fields_ = [('p', c_int32), ('n_tbl', c_int32), ('alpha_mm', c_float), ('threshold', c_float),
                ('lc_min_zeros', c_int32), ('reserved', c_int32),
                ('raw_est', c_void_p), ('bias', c_void_p), ('lc_table', c_void_p)]


class CsrGraphStruct(ctypes.Structure):
    """mirror of `struct ss_csr_graph`"""
    _fields_ = [('rowptr', c_void_p), ('col', c_void_p), ('num_nodes', c_int64), ('n_self_loops', c_int64),
                ('n_self_loops_dev', c_void_p), ('hub_threshold', c_int32), ('reserved', c_int32),
                ('hub_rows', c_void_p), ('hub_count', c_void_p), ('mega_rows', c_void_p), ('mega_count', c_void_p),
                ('mega_scratch', c_void_p), ('row_begin', c_int64), ('row_end', c_int64),
                ('n_mirrors', c_int32), ('reserved2', c_int32), ('mirror_mh', c_void_p * 7), ('mirror_hll', c_void_p * 7),
                ('mirror_cards', c_void_p * 7), ('hub_report', c_void_p), ('report_hub_count', c_void_p), ('report_mega_count', c_void_p)]


ABI_VERSION = 129  # ss_version() of the library this module's struct mirrors and signatures describe
PROF_MINHASH_HOP, PROF_HLL_HOP, PROF_FIRST_HOP_MH, PROF_FIRST_HOP_HLL, PROF_PAIRS, PROF_CSR, PROF_HUB, PROF_FUSED, PROF_MINHASH_ROWS = range(9)  # SS_PROF_* tags
MEGA_DESC_WORDS = 8  # SS_MEGA_DESC_WORDS
MEGA_SLICE, MEGA_SLOT_BYTES, CSR_FINGERPRINT_BYTES, MAX_MIRRORS = 1024, 1280, 8448, 7  # SS_MEGA_SLICE / SS_MEGA_SLOT_BYTES of include/subgraph_sketch.h


# name -> (restype, argtypes); must list every symbol declared in include/subgraph_sketch.h and include/subgraph_sketch_debug.h
SIGNATURES = {
    'ss_version': (c_int32, []),
    'ss_error_string': (c_char_p, [c_int32]),
    'ss_minhash_init': (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int32, c_void_p]),
    'ss_hll_init': (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    'ss_csr_workspace_bytes': (c_size_t, [c_int64, c_int64]),
    'ss_csr_build': (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ss_csr_build_cached': (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'ss_propagate': (c_int32, [POINTER(CsrGraphStruct), c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                               c_int32, c_void_p, c_int64, POINTER(HllParams), c_void_p]),
    'ss_minhash_hop_rows': (c_int32, [POINTER(CsrGraphStruct), c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    'ss_first_hop': (c_int32, [POINTER(CsrGraphStruct), c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                               c_void_p, c_void_p, c_int64, POINTER(HllParams), c_void_p]),
    'ss_fused_hop_stage': (c_int32, [POINTER(CsrGraphStruct), c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int64, POINTER(HllParams), c_void_p]),
    'ss_hll_count': (c_int32, [c_void_p, c_int64, POINTER(HllParams), c_void_p, c_int64, c_void_p]),
    'ss_estimate_bias': (c_int32, [c_void_p, c_int64, POINTER(HllParams), c_void_p, c_int32, c_void_p]),
    'ss_pair_features': (c_int32, [c_void_p, c_int64, c_int64, c_int32, POINTER(c_void_p), c_int32, POINTER(c_void_p),
                                   c_void_p, c_int64, POINTER(HllParams), c_uint32, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    'ss_pair_features_normalised': (c_int32, [c_void_p, c_int64, c_int64, c_int32, POINTER(c_void_p), c_int32,
                                              POINTER(c_void_p), c_void_p, c_int64, POINTER(HllParams), c_uint32, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    'ss_group_links_by_source': (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ss_gather_links': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'ss_scatter_feature_rows': (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    'ss_pair_features_grouped': (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_int32, POINTER(c_void_p), c_int32, POINTER(c_void_p),
                                           c_void_p, c_int64, POINTER(HllParams), c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ss_pair_features_grouped_kernel': (c_int32, [c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int32, POINTER(c_void_p), c_int32,
                                                  POINTER(c_void_p), c_void_p, c_int64, POINTER(HllParams), c_uint32, c_void_p, c_void_p,
                                                  c_void_p, c_void_p]),
    'ss_common_neighbour_scores': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                             c_void_p]),
    'ss_spmm_csr': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    'ss_csr_group_ids': (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ss_csr_sort_workspace_bytes': (c_size_t, [c_int64]),
    'ss_csr_sort_rows': (c_int32, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ss_gcn_scan_bytes': (c_size_t, [c_int64]),
    'ss_gcn_scan_edges': (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    'ss_gcn_degree': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ss_sign_spmm': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    'ss_csr_protocol_faults': (c_int32, []),
    'ss_table_digest': (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    'ss_pack_minhash': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    'ss_unpack_minhash': (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    'ss_profile_enable': (c_int32, [c_uint32]),
    'ss_profile_sample': (c_int32, [c_int32]),
    'ss_debug_hub_calls': (c_int64, [c_int32]),
    'ss_debug_csr_helpers': (c_int32, [c_void_p]),
    'ss_debug_csr_protocol_faults': (c_int32, []),
    'ss_profile_read': (c_int32, [c_int32, POINTER(c_float), POINTER(c_int32)]),
    'ss_time_propagate': (c_int32, [POINTER(CsrGraphStruct), c_void_p, c_void_p, c_int32, c_void_p,
                                    c_void_p, c_int32, c_void_p, c_int64, POINTER(HllParams), c_void_p, c_int32,
                                    POINTER(c_float)]),
    'ss_time_pair_features': (c_int32, [c_void_p, c_int64, c_int64, c_int32, POINTER(c_void_p), c_int32,
                                        POINTER(c_void_p), c_void_p, c_int64, POINTER(HllParams), c_uint32, c_void_p,
                                        c_void_p, c_int32, POINTER(c_float)]),
}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """load (once) and return the ctypes handle; raises NativeLibraryMissing if it was never built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f'{LIB_PATH} not found: the HIP engine is not built. Run `python __graft_entry__.py` '
                f'(or subgraph-sketching_amd/csrc/build.sh). There is no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.ss_version() != ABI_VERSION:  # a stale in-tree build with another struct layout would corrupt launches
            raise NativeLibraryMissing(f'{LIB_PATH} is version {handle.ss_version()}, this package needs {ABI_VERSION}: '
                                       f'rebuild with `python __graft_entry__.py`')
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().ss_error_string(code).decode()
        raise RuntimeError(f'{what} failed: {msg} ({code})')
