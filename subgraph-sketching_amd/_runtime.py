"""Plumbing shared by the host modules: device / stream / pointer helpers, the HIP-event span hook, the device error words and
their deferred reporting (reference hashing.py has no counterpart: its bounds errors are torch's own IndexError), the device copy
of the HLL++ estimator tables."""
import atexit
import logging
import os
import weakref
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from . import _native, hll_tables, knobs

logger = logging.getLogger('subgraph_sketching_amd.hashing')
logger.setLevel(logging.INFO)


class _Span(object):
    """optional HIP-event bracket around a launch, recorded on the launch stream"""

    def __init__(self, name, device):
        timer = knobs.KERNEL_TIMER
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


_FAULTS_REPORTED = [0]


class CsrProtocolFault(RuntimeError):
    """a CSR build (ss_csr_build*, ss_group_links_by_source, ss_csr_group_ids) gave up a cross-workgroup wait after its ~2 s bound:
    its outputs are incomplete.  Only a process descheduled for seconds can cause it (csrc/ss_csr.hip "who may wait for whom")."""


def raise_csr_protocol_faults():
    """non-waiting look at the library's process-wide count of given-up waits (pinned host memory, ss_csr_protocol_faults): raises
    once for every increase -- from the next call into the engine after the failed build completed, or from check_errors()"""
    n = _native.lib().ss_csr_protocol_faults()  # 0: none so far; a stamp that changes with every further one; -1: no device
    if n > 0 and n != _FAULTS_REPORTED[0]:
        _FAULTS_REPORTED[0] = n
        raise CsrProtocolFault(f'a cross-workgroup wait of a CSR build gave up after its time bound: the adjacency / link grouping '
                               f'that build produced is INCOMPLETE and everything computed from it since is wrong -- rebuild (this '
                               f'happens only when the process is descheduled for seconds: a debugger, a heavily oversubscribed GPU)')


def mark_csr_protocol_faults_reported():
    """the caller is about to raise for a fault it saw in a build's own err_flag: the library's word says nothing new"""
    _FAULTS_REPORTED[0] = _native.lib().ss_csr_protocol_faults()


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
                bits = int(flag[0])
                flag.zero_()
                calls, self._calls = ', '.join(self._calls), []
                if bits & _native.SS_CSR_ERR_PROTOCOL:  # (bit 1 of a CSR build's err_flag; bit 0 = ids out of range)
                    mark_csr_protocol_faults_reported()
                    raise CsrProtocolFault(f'a CSR build on this engine gave up a cross-workgroup wait (its adjacency is incomplete); calls '
                                           f'since the last clean check: {calls}')
                raise IndexError(f'an earlier call on this engine was given node ids outside its num_nodes (reported late: '
                                 f'strict_bounds="deferred"); calls since the last clean check: {calls}. Out-of-range edges '
                                 f'were dropped and out-of-range pairs returned NaN rows')
        if synchronize:
            self._calls = []
        raise_csr_protocol_faults()


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


