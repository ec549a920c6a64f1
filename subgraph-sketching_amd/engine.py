"""ElphHashes: the reference's class (hashing.py:48-323), same constructor, methods, attributes and errors, over the HIP engine."""
from collections import OrderedDict
import logging
import os
import weakref
from ctypes import byref, c_float, c_void_p

import numpy as np
import torch

from . import _native, hll_tables, knobs
from ._runtime import (CsrProtocolFault, raise_csr_protocol_faults, _DeferredErrors, _DeviceParams, _Span, _check_sizes, _compute_device, _error_flag, _ptr, _stream, _take_error,
                       linear_counting_table, logger)
from .containers import (HopSketch, LazyMinhash, SketchTable, _packed_hll_of, _packed_minhash_of, _stamp_tables, _tag, unpack_minhash)
from .csr import _CsrCache, build_csr, group_links_by_source
from .propagation import HllPropagation, MinhashPropagation, _propagate

# reference hashing.py:22-25 -- primary key = max hops, secondary key = feature index, value = (hops from u, hops from v)
LABEL_LOOKUP = {1: {0: (1, 1), 1: (0, 1), 2: (1, 0)},
                2: {0: (1, 1), 1: (2, 1), 2: (1, 2), 3: (2, 2), 4: (0, 1), 5: (1, 0), 6: (0, 2), 7: (2, 0)},
                3: {0: (1, 1), 1: (2, 1), 2: (1, 2), 3: (2, 2), 4: (3, 1), 5: (1, 3), 6: (3, 2), 7: (2, 3), 8: (3, 3),
                    9: (0, 1), 10: (1, 0), 11: (0, 2), 12: (2, 0), 13: (0, 3), 14: (3, 0)}}


class ElphHashes(object):
    """class to store hashes and retrieve subgraph features (mirror of reference hashing.py:48-323)"""
    HUB_HINT_SHAPES = 256  # graph shapes that get a hub hint word (one pinned int32 each, kept for the engine's lifetime)

    def __init__(self, args, fuse_hop_stage=None, defer_first_hop=None, defer_table_hop=None):
        """args: the reference's namespace (max_hash_hops, floor_sf, minhash_num_perm, hll_p, use_zero_one).  Extensions (keyword
        only in spirit; None = the engine's defaults): fuse_hop_stage -- hop-1 MinHash + hop-2 HLL in one launch inside
        build_hash_tables; defer_first_hop / defer_table_hop -- the deferred launches of the ELPH call sequence (minhash_prop
        records its hop, the next consumer decides how much of it runs).  The environment variables
        SS_FUSED_STAGE / SS_DEFER_TABLE_HOP only set the defaults (tests, A/B measurements)."""
        assert args.max_hash_hops in {1, 2, 3}, f'hashing is not implemented for {args.max_hash_hops} hops'
        self._defer_first_hop, self._defer_table_hop = defer_first_hop, defer_table_hop
        self.max_hops = args.max_hash_hops
        self.floor_sf = args.floor_sf  # if true set minimum sf to 0
        # minhash params (reference hashing.py:58-63)
        self._mersenne_prime = np.uint64((1 << 61) - 1)
        self._max_minhash = np.uint64((1 << 32) - 1)
        self._minhash_range = (1 << 32)
        self.minhash_seed = 1
        self.num_perm = args.minhash_num_perm
        self._csr_cache = _CsrCache(self._bounds, self._prop_hub_hint)
        self.minhash_prop = MinhashPropagation(self._csr_cache, self._report_after_host_copy, defer_first_hop, defer_table_hop)
        # hll params (reference hashing.py:65-81)
        self.p = args.hll_p
        self.m = 1 << self.p
        self.use_zero_one = args.use_zero_one
        self.label_lookup = LABEL_LOOKUP[self.max_hops]
        self.hll_tables = hll_tables.load(self.p)
        self.hll_hashfunc = None  # datasketch's sha1 hashfunc is never used on the path (reference :71)
        self.alpha = self.hll_tables.alpha
        self.max_rank = self.hll_tables.max_rank
        assert self.max_rank == 64 - self.p, 'not using 64 bits for hll++ hashing'
        self.hll_size = self.m
        self.hll_threshold = self.hll_tables.threshold
        self.bias_vector = torch.tensor(self.hll_tables.bias, dtype=torch.float)
        self.estimate_vector = torch.tensor(self.hll_tables.raw_estimate, dtype=torch.float)
        self.hll_prop = HllPropagation(self._csr_cache, self._params, self.m, self._report_after_host_copy)
        self._dev_params = {}
        self._dev_perms = {}
        self.fuse_first_hop = True  # compute hop 1 straight from node ids when the fused kernel supports (num_perm, p)
        # hop-1 MinHash + hop-2 HLL in one launch (ss_fused_hop_stage; num_perm == 128, hll_p == 8, max_hops >= 2, unsharded build)
        self.fuse_hop_stage = (os.environ.get('SS_FUSED_STAGE', '1') != '0') if fuse_hop_stage is None else bool(fuse_hop_stage)
        # node ids outside [0, num_nodes): 'deferred' (default) = IndexError at the NEXT call into this engine or at
        # check_errors(), no host synchronisation inside a step; True = IndexError from the offending call itself (one
        # synchronising 4-byte read per CSR build / query call); False = never reported (edges dropped, NaN feature rows)
        self.strict_bounds = 'deferred'
        self._deferred = _DeferredErrors()
        # link sets of >= knobs.GROUP_LINKS_MIN pairs: 'auto' = grouped by their first node unless the list already has its runs (one
        # host read per such call), True = always grouped, False = walked as listed (no host read)
        self.group_links = 'auto'
        # skip the hub units of build_hash_tables for shapes whose earlier builds listed no hub rows (see _hub_hint)
        self.hub_hints = os.environ.get('SS_HUB_HINTS', '1') != '0'
        self._hub_words, self._hub_arena = {}, None

    # no device handles in pickled state (SURVEY.md section 8(b) threading row)
    def __getstate__(self):
        state = dict(self.__dict__)
        state['_dev_params'], state['_dev_perms'] = {}, {}
        state['_csr_cache'], state['_deferred'] = None, None
        state['_hub_words'], state['_hub_arena'] = {}, None
        state.pop('_tables_id', None)
        state['minhash_prop'], state['hll_prop'] = None, None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault('hub_hints', os.environ.get('SS_HUB_HINTS', '1') != '0')
        self.__dict__.setdefault('group_links', 'auto')
        self._hub_words, self._hub_arena = {}, None
        self._deferred = _DeferredErrors()
        self._csr_cache = _CsrCache(self._bounds, self._prop_hub_hint)
        self.minhash_prop = MinhashPropagation(self._csr_cache, self._report_after_host_copy, self.__dict__.get('_defer_first_hop'),
                                               self.__dict__.get('_defer_table_hop'))
        self.hll_prop = HllPropagation(self._csr_cache, self._params, self.m, self._report_after_host_copy)

    def _report_after_host_copy(self):
        """a result has just been copied to a CPU caller (ELPH on CPU tensors): the launches behind it are complete, so a
        deferred bounds report is final -- raise it from the offending call, as the reference's CPU indexing would"""
        if self.strict_bounds == 'deferred':
            self._deferred.raise_if_set()

    # ---- host-side helpers -------------------------------------------------------------------------
    @property
    def tables_id(self):
        """identity of the HLL++ tables in use (hll_tables.table_id), recomputed if `hll_tables` is replaced"""
        cached = self.__dict__.get('_tables_id')
        if cached is None or cached[0] is not self.hll_tables:
            cached = (self.hll_tables, hll_tables.table_id(self.hll_tables))
            self.__dict__['_tables_id'] = cached
        return cached[1]

    def _params(self, device):
        key = str(device)
        if key not in self._dev_params:
            _check_sizes(self.num_perm, self.p)
            self._dev_params[key] = _DeviceParams(self.hll_tables, device)
        return self._dev_params[key]

    def _np_bit_length(self, bits):
        """number of bits needed to represent each (non-negative) int in `bits` (reference :83-89), computed
        exactly in integer arithmetic"""
        b = np.asarray(bits).astype(np.uint64)
        n = np.zeros(b.shape, dtype=np.int64)
        for s in (32, 16, 8, 4, 2, 1):
            big = b >= (np.uint64(1) << np.uint64(s))
            n = np.where(big, n + s, n)
            b = np.where(big, b >> np.uint64(s), b)
        return (n + (b > 0)).astype(int)

    def _get_hll_rank(self, bits):
        """rank = leading zeros of `bits` seen as a (64 - p)-bit word, plus one (reference :91-104)"""
        rank = self.max_rank - self._np_bit_length(bits) + 1
        if rank.size and rank.min() <= 0:
            raise ValueError("Hash value overflow, maximum size is %d bits" % self.max_rank)
        return rank

    def _init_permutations(self, num_perm):
        """universal-hash parameters (a_j, b_j), j < num_perm, from numpy's legacy RandomState(seed): the draws
        interleave a_0, b_0, a_1, b_1, ... (reference :106-116).  uint64 [2, num_perm]."""
        gen = np.random.RandomState(self.minhash_seed)
        ab = np.empty((2, num_perm), dtype=np.uint64)
        for j in range(num_perm):
            ab[0, j] = gen.randint(1, self._mersenne_prime, dtype=np.uint64)
            ab[1, j] = gen.randint(0, self._mersenne_prime, dtype=np.uint64)
        return ab

    def _perms(self, device):
        key = str(device)
        if key not in self._dev_perms:
            ab = self._init_permutations(self.num_perm).view(np.int64)
            self._dev_perms[key] = torch.from_numpy(ab.copy()).to(device)
        return self._dev_perms[key]

    # ---- hop-0 sketches ------------------------------------------------------------------------------
    def _init_minhash_u32(self, n_nodes, device):
        _check_sizes(self.num_perm, self.p)
        ab = self._perms(device)
        out = torch.empty((n_nodes, self.num_perm), dtype=torch.int32, device=device)
        _native.check(_native.lib().ss_minhash_init(_ptr(out), 0, n_nodes, _ptr(ab[0]), _ptr(ab[1]), self.num_perm,
                                                    _stream(device)), 'ss_minhash_init')
        return out

    def _init_hll_u8(self, n_nodes, device):
        _check_sizes(self.num_perm, self.p)
        out = torch.empty((n_nodes, self.m), dtype=torch.uint8, device=device)
        _native.check(_native.lib().ss_hll_init(_ptr(out), 0, n_nodes, self.p, _stream(device)), 'ss_hll_init')
        return out

    def initialise_minhash(self, n_nodes):
        """int64 [n_nodes, num_perm] hop-0 MinHash rows (reference :118-124); lives on the HIP device"""
        device = _compute_device()
        packed = self._init_minhash_u32(n_nodes, device)
        out = unpack_minhash(packed)
        _tag(out, '_ss_u32', packed)
        # hop-0 marker: lets minhash_prop compute the first hop straight from node ids (ss_first_hop) instead of
        # gathering this table; voided by any in-place edit (version counter)
        out._ss_hop0 = (out._version, self._perms(device), self.p)
        return out

    def initialise_hll(self, n_nodes):
        """int8 [n_nodes, m] hop-0 HLL rows, one non-zero register each (reference :126-137)"""
        device = _compute_device()
        packed = self._init_hll_u8(n_nodes, device)
        out = packed.view(torch.int8)
        _tag(out, '_ss_u8', packed)
        out._ss_hop0 = (out._version, None, self.p)
        return out

    # ---- build ---------------------------------------------------------------------------------------
    def _bounds(self, device, what):
        """-> (check, err_flag) for a launch that validates node ids, after raising what an earlier deferred launch reported"""
        if self.strict_bounds == 'deferred':
            self._deferred.raise_if_set()
            return False, self._deferred.flag(device, what)
        raise_csr_protocol_faults()  # (strict / unchecked modes pass no deferred word: the library's own count still reports)
        return bool(self.strict_bounds), None

    def _prop_hub_hint(self, device, num_nodes, edge_index):
        """the same hint for the CSR of hll_prop / minhash_prop (the ELPH call sequence)"""
        return self._hub_hint(device, num_nodes, edge_index) if getattr(self, 'hub_hints', False) else None

    def _hub_hint(self, device, num_nodes, edge_index):
        """-> (pinned host word a build of this shape reports its hub + mega row count into, whether an EARLIER build of
        the shape reported none).  The word is read without synchronising -- it holds whatever the most recent COMPLETED build of
        the shape left (-1: none yet) -- and is only a hint: with it, an unskewed graph is built without hub lists (no leading hub
        workgroups in its launches; rounds 1-3: without the two hub-pass launches per hop that found nothing to do, 9 us of a 0.455 ms
        step at ogbl-collab size); if the hint is stale (another graph of the
        same shape that does have hub rows) those rows are walked by single wavefronts once -- slow, never wrong -- and the next
        build of the shape has its hub units back.  `eh.hub_hints = False` keeps them unconditionally."""
        key = (str(device), int(num_nodes), tuple(edge_index.shape), knobs.HUB_THRESHOLD)
        word = self._hub_words.get(key)
        if word is None:
            # One pinned arena for the engine's lifetime, one word per shape, NEVER handed back while the engine lives: a first-hop
            # launch still in flight stores into its word (system-scope store, csrc report_hub_rows) -- a word returned to torch's
            # pinned-memory cache could be given to somebody else by then.  More shapes than words: no hint for them.
            if self._hub_arena is None:
                self._hub_arena = torch.full((self.HUB_HINT_SHAPES,), -1, dtype=torch.int32).pin_memory()
            if len(self._hub_words) >= self.HUB_HINT_SHAPES:
                return None, False
            word = self._hub_words[key] = self._hub_arena[len(self._hub_words):len(self._hub_words) + 1]
        return word, int(word[0]) == 0

    def check_errors(self):
        """strict_bounds = 'deferred': wait for the launches issued so far and raise IndexError if any of them met a node id
        outside its num_nodes (call once after preprocessing / at the end of an epoch; every call into the engine also
        performs the non-waiting form of this check)"""
        self._deferred.raise_if_set(synchronize=True)

    def build_hash_tables(self, num_nodes, edge_index):
        """k-hop sketches of every node, k = 0..max_hops, and their HLL cardinalities (reference :139-165).
        @return: (SketchTable {k: {'hll','minhash'}}, cards float32 [num_nodes, max_hops])"""
        return self._build(num_nodes, edge_index, None)

    def _build(self, num_nodes, edge_index, shard):
        """shard = None: this process computes every row.  Otherwise (dist.sharded_build_hash_tables) an object with
        `rows` = (begin, end) owned by this rank, `padded_rows` >= num_nodes (allocation size, a multiple of the world
        size) and `gather(tensor) -> handle` / `wait(handle)`: in-place all-gather of the owned row blocks.  The two
        sketches are launched separately so that the gather of one overlaps the kernel of the other."""
        home = edge_index.device
        device = _compute_device(edge_index)
        params = self._params(device)
        # add_self_loops without num_nodes (reference :148): loops for i < max(edge_index) + 1 only; the count is
        # produced on the device by ss_csr_build and read by the propagation kernel -- no host round trip
        check, err_flag = self._bounds(device, f'build_hash_tables(num_nodes={num_nodes})')
        report, no_hubs = (None, False) if (check or not self.hub_hints) else self._hub_hint(device, num_nodes, edge_index)
        csr = build_csr(edge_index, num_nodes, device, check=check, err_flag=err_flag)
        csr.hub_report = report  # (the first-hop kernels of this build leave its hub + mega row count there)
        if no_hubs:  # an earlier build of this shape listed no hub / mega rows: no hub units (the row kernels walk every row)
            csr.has_hub_rows = False
        csr.use_inferred_self_loops = True
        rows = None if shard is None else shard.rows
        n_alloc = num_nodes if shard is None else shard.padded_rows
        cards = torch.empty((n_alloc, self.max_hops), dtype=torch.float32, device=device)
        table = SketchTable()
        h = self.max_hops
        # hop 1 from node ids: MinHash for 64 / 128 / 192 / 256 permutations at any hll_p, HLL at hll_p == 8
        fused_mh = self.fuse_first_hop and self.num_perm % 64 == 0 and self.num_perm <= 256
        fused = fused_mh and self.p == 8
        peer = shard is not None and getattr(shard, 'peer_write', False)
        if peer:  # the shard's persistent, IPC-shared tables (every rank's launches store into every rank's copy)
            mh, hll, cards = shard.tables(h, self.num_perm, self.m, device)
        else:
            mh = [torch.empty((n_alloc, self.num_perm), dtype=torch.int32, device=device) for _ in range(h)]
            hll = [torch.empty((n_alloc, self.m), dtype=torch.uint8, device=device) for _ in range(h)]
        if fused:
            # hop 1 is computed straight from node ids (ss_first_hop); the hop-0 tables (pure functions of the node id,
            # never read by get_subgraph_features) are produced only if a caller actually looks at them
            table[0] = HopSketch(None, None, home, make_packed=lambda n=num_nodes, d=device: (self._init_minhash_u32(n, d),
                                                                                            self._init_hll_u8(n, d)))
            mh_prev = hll_prev = None
        elif fused_mh:
            hll_prev = self._init_hll_u8(num_nodes, device)
            table[0] = HopSketch(None, None, home, make_packed=lambda n=num_nodes, d=device, l=hll_prev: (self._init_minhash_u32(n, d), l))
            mh_prev = None
        else:
            mh_prev = self._init_minhash_u32(num_nodes, device)  # hop 0 is replicated: a pure function of the node id
            hll_prev = self._init_hll_u8(num_nodes, device)
            table[0] = HopSketch(mh_prev, hll_prev, home)
        # (collab-like -4 %, ppa-like -5 %, citation2-like -1.2 % per step against the unfused schedule)
        if (shard is None and fused and h >= 2 and self.num_perm == 128 and self.fuse_hop_stage
                and num_nodes * self.m <= knobs.FUSED_STAGE_MAX_TABLE_BYTES):
            # ONE call for hops 1 and 2: hop-1 HLL first (hop-2 HLL rows need that table complete), then hop-1 MinHash + hop-2 HLL
            # in one launch (the VALU-bound first hop and the memory-bound table hop interleaved inside every wavefront,
            # csrc/ss_fused_hop.hip), then hop-2 MinHash; further hops unfused
            ab = self._perms(device)
            graph = csr.struct()
            with _Span('fused_hop_stage', device):
                _native.check(_native.lib().ss_fused_hop_stage(byref(graph), _ptr(ab[0]), _ptr(ab[1]), self.num_perm, _ptr(mh[0]), _ptr(mh[1]),
                                                               self.p, _ptr(hll[0]), _ptr(cards), _ptr(hll[1]), _ptr(cards[:, 1]), h,
                                                               byref(params.struct), _stream(device)), 'ss_fused_hop_stage')
            for k in range(3, h + 1):
                _propagate(csr, mh[k - 2], hll[k - 2], device, cards_out=cards[:, k - 1], cards_stride=h, params=params,
                           mh_out=mh[k - 1], hll_out=hll[k - 1])
        elif shard is None:
            # (inside the library each of these calls is one launch per sketch, each hosting its hub units: measured faster than
            # two-sketch kernels -- first hop 37 + 134 us vs 184, table hop 111 + 192 us vs 326 on the bench graph)
            for k in range(1, h + 1):
                if k == 1 and fused:
                    self._first_hop(csr, device, mh[0], hll[0], cards, params)
                elif k == 1 and fused_mh:
                    self._first_hop(csr, device, mh[0], None, None, params)
                    _propagate(csr, None, hll_prev, device, cards_out=cards[:, 0], cards_stride=h, params=params, hll_out=hll[0])
                else:
                    logger.info(f"Calculating hop {k} hashes")
                    _propagate(csr, mh_prev, hll_prev, device, cards_out=cards[:, k - 1], cards_stride=h, params=params,
                               mh_out=mh[k - 1], hll_out=hll[k - 1])
                mh_prev, hll_prev = mh[k - 1], hll[k - 1]
        elif peer:
            # peer-write: no exchange step -- the kernels store every finished row into all ranks' tables while they run
            # (csrc: mirror_* stores); a hop may start once EVERY rank's launches of the previous hop are complete
            if not (fused_mh and fused):
                raise NotImplementedError('the peer-write build is built for the default sketch shape (128 permutations, hll_p = 8)')
            for k in range(1, h + 1):
                mir_mh, mir_hll = shard.mirrors('mh', k - 1), shard.mirrors('hll', k - 1)
                none = [0] * len(mir_mh)
                if k == 1:
                    self._first_hop(csr, device, mh[0], None, None, params, rows=rows, mirrors=(mir_mh, none, none))
                    self._first_hop(csr, device, None, hll[0], cards, params, rows=rows, mirrors=(none, mir_hll, shard.mirrors('cards', 0)))
                else:
                    _propagate(csr, mh[k - 2], None, device, mh_out=mh[k - 1], rows=rows, mirrors=(mir_mh, none, none))
                    _propagate(csr, None, hll[k - 2], device, cards_out=cards[:, k - 1], cards_stride=h, params=params, hll_out=hll[k - 1],
                               rows=rows, mirrors=(none, mir_hll, shard.mirrors('cards', k - 1)))
                shard.hop_barrier()
            cards = cards[:num_nodes]
        else:
            pending_mh = pending_hll = None
            for k in range(1, h + 1):
                shard.wait(pending_mh)  # hop k-1 MinHash rows of every rank have arrived
                if k == 1 and fused_mh:
                    self._first_hop(csr, device, mh[0], None, None, params, rows=rows)
                else:
                    _propagate(csr, mh_prev, None, device, mh_out=mh[k - 1], rows=rows)
                pending_mh = shard.gather(mh[k - 1])
                shard.wait(pending_hll)
                if k == 1 and fused:
                    self._first_hop(csr, device, None, hll[0], cards, params, rows=rows)
                else:
                    _propagate(csr, None, hll_prev, device, cards_out=cards[:, k - 1], cards_stride=h, params=params,
                               hll_out=hll[k - 1], rows=rows)
                pending_hll = shard.gather(hll[k - 1])
                mh_prev, hll_prev = mh[k - 1], hll[k - 1]
            shard.wait(pending_mh)
            shard.wait(pending_hll)
            shard.wait(shard.gather(cards))
            cards = cards[:num_nodes]
        for k in range(1, h + 1):
            table[k] = HopSketch(mh[k - 1][:num_nodes], hll[k - 1][:num_nodes], home)
        if home != device:
            cards = cards.to(home)
            if self.strict_bounds == 'deferred':  # (the copy has waited for the build: the report is final, see get_subgraph_features)
                self._deferred.raise_if_set()
        return table, _stamp_tables(cards, self.tables_id)

    def _first_hop(self, csr, device, mh_out, hll_out, cards, params, rows=None, mirrors=None):
        """fused hop-0 + hop-1 (ss_first_hop) for either or both sketches"""
        ab = self._perms(device)
        graph = csr.struct(rows, mirrors)
        with _Span('first_hop_mh' if hll_out is None else ('first_hop_hll' if mh_out is None else 'first_hop'), device):
            _native.check(_native.lib().ss_first_hop(byref(graph), _ptr(ab[0]), _ptr(ab[1]), self.num_perm, _ptr(mh_out), self.p,
                                                     _ptr(hll_out), _ptr(cards) if hll_out is not None else None, self.max_hops,
                                                     byref(params.struct), _stream(device)), 'ss_first_hop')

    # ---- query ---------------------------------------------------------------------------------------
    def _resolve_tables(self, hash_table, device, rows=None):
        """packed tables of hops 1 .. max_hops; rows: the node ids the caller is going to read (a deferred LAST hop -- nobody
        else's input -- is then computed for those rows only, knobs.DEFER_TABLE_HOP)"""
        mh, hll = [], []
        for k in range(1, self.max_hops + 1):
            entry = hash_table[k]
            if isinstance(entry, HopSketch) and entry.mh_u32.device == device:
                m, l = entry.packed(device)
                mh.append(m)
                hll.append(l)
            else:
                t = entry['minhash']
                tw = None
                if rows is not None and k == self.max_hops and isinstance(t, LazyMinhash) and t.device == device:
                    tw = t.packed_for_rows(rows)
                mh.append(tw if tw is not None else _packed_minhash_of(t, device))
                hll.append(_packed_hll_of(entry['hll'], device))
        N, P = mh[0].shape
        for a, b in zip(mh, hll):
            if a.shape != (N, P) or b.shape != (N, self.m):
                raise ValueError('hash tables of different hops must have the same shape')
        return mh, hll, N, P

    def _pair_kernel(self, links, hash_table, cards, want_debug=False, degrees=None, floor_sf=None, group_batch=None, out=None):
        """runs ss_pair_features for links [B,2]; returns (features [B,nf] (or [B,2nf] with degrees) on device, debug dict or None).
        group_batch: the links are first grouped by their first node and walked in that order, `group_batch` pairs per launch
        (knobs.GROUP_LINKS_MIN; same rows, in the caller's order).  out: the rows are written THERE (a contiguous float32
        [B, width] tensor on the compute device -- a slice of a larger result, dist.sharded_precompute) instead of a fresh tensor"""
        # where the links live, else where the packed tables already are, else cards, else the current device
        first = hash_table.get(1) if hasattr(hash_table, 'get') else None
        device = _compute_device(links, first.mh_u32 if isinstance(first, HopSketch) else None, cards)
        params = self._params(device)
        lk = links.to(device=device, dtype=torch.int64).contiguous()
        mh, hll, N, P = self._resolve_tables(hash_table, device, rows=lk)
        h = self.max_hops
        B = lk.size(0)
        if cards is None:
            cd = torch.zeros((N, h), dtype=torch.float32, device=device)
        else:
            made_with = getattr(cards, '_ss_tables', None)
            if made_with is not None and not hll_tables.same_tables(made_with, self.tables_id):
                raise ValueError(f'cards were estimated with HLL++ tables {made_with}, this engine uses {self.tables_id}: '
                                 f'a feature row would mix two bias tables (rebuild the cache or load the same tables)')
            # ELPH keeps `cards` on the CPU and the reference re-uploads it on every call (hashing.py:274): keep a device
            # twin on the tensor, invalidated by in-place edits, so repeated eval batches do not pay the copy again
            tag = getattr(cards, '_ss_cards', None)
            if cards.device == device and cards.dtype == torch.float32:
                cd = cards
            elif tag is not None and tag[0] == cards._version and tag[1].device == device:
                cd = tag[1]
            else:
                cd = cards.to(device=device, dtype=torch.float32)
                _tag(cards, '_ss_cards', cd)
            if cd.dim() != 2 or cd.size(0) != N or cd.size(1) < h:
                raise ValueError(f'cards must have shape [{N}, >= {h}], got {tuple(cd.shape)}')
            if cd.stride(1) != 1:
                cd = cd.contiguous()
        nf = h * (h + 2)
        width = 2 * nf if degrees is not None else nf
        if out is not None and (out.device != device or out.dtype != torch.float32 or tuple(out.shape) != (B, width) or not out.is_contiguous()):
            raise ValueError(f'out must be a contiguous float32 [{B}, {width}] tensor on {device}, got {out.dtype} {tuple(out.shape)} on {out.device}')
        mh_ptrs = (c_void_p * h)(*[t.data_ptr() for t in mh])
        hll_ptrs = (c_void_p * h)(*[t.data_ptr() for t in hll])
        floor = self.floor_sf if floor_sf is None else floor_sf  # DeviceFeatureStore records HashDataset's post-hoc floor
        flags = (_native.SS_FLAG_USE_ZERO_ONE if self.use_zero_one else 0) | (_native.SS_FLAG_FLOOR_SF if floor else 0)
        strict, err = self._bounds(device, f'get_subgraph_features({B} links, num_nodes={N})')
        if strict:
            err = _error_flag(device)  # (non-strict launches never touch the shared flag)
        dg = None
        if degrees is not None:
            dg = degrees.to(device=device, dtype=torch.float32).contiguous()
            if dg.dim() != 1 or dg.numel() != N:
                raise ValueError(f'degrees must have shape [{N}], got {tuple(dg.shape)}')
        if group_batch and not want_debug and 1 < B < (1 << 31) and N < (1 << 31):
            # a list that already has its runs (a coalesced edge list, an evaluation set listing every source's negatives together)
            # is walked as it is: grouping it again costs ~8 % and scatters the output rows.  One small reduction + ONE host read
            # per call of >= knobs.GROUP_LINKS_MIN links (0.5 ms of query and more); ElphHashes.group_links = True / False skips it.
            mode = getattr(self, 'group_links', 'auto')  # (instances pickled before the attribute existed)
            if mode == 'auto':
                mode = float((lk[1:, 0] == lk[:-1, 0]).sum().item()) < 0.5 * (B - 1)
            order = group_links_by_source(lk, N, device) if mode else None
            nf_out = width
            if out is None:
                out = torch.empty((B, nf_out), dtype=torch.float32, device=device)
            for s0 in range(0, B, group_batch):
                nb = min(group_batch, B - s0)
                lib = _native.lib()
                with _Span('pair_features', device):
                    if order is not None and B >= knobs.GROUP_GATHER_MIN:
                        # a set of gigabytes: gather the chunk's links, query the (now contiguous, grouped) chunk, scatter its rows
                        o = c_void_p(order.data_ptr() + 4 * s0)
                        lk_c = torch.empty((nb, 2), dtype=torch.int64, device=device)
                        rows_c = torch.empty((nb, nf_out), dtype=torch.float32, device=device)
                        _native.check(lib.ss_gather_links(_ptr(lk), o, nb, _ptr(lk_c), _stream(device)), 'ss_gather_links')
                        _native.check(lib.ss_pair_features_grouped(_ptr(lk_c), None, nb, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0),
                                                                   byref(params.struct), flags, _ptr(dg), _ptr(rows_c), _ptr(err),
                                                                   _stream(device)), 'ss_pair_features_grouped')
                        _native.check(lib.ss_scatter_feature_rows(_ptr(rows_c), o, nb, nf_out, _ptr(out), _stream(device)), 'ss_scatter_feature_rows')
                        continue
                    if order is not None:  # every launch writes rows out[order[s0 + t]] of the ONE output tensor
                        args = (_ptr(lk), c_void_p(order.data_ptr() + 4 * s0), nb, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0),
                                byref(params.struct), flags, _ptr(dg), _ptr(out), _ptr(err), _stream(device))
                    else:                  # as listed: a slice of the links and the matching slice of the output
                        args = (c_void_p(lk.data_ptr() + 16 * s0), None, nb, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0),
                                byref(params.struct), flags, _ptr(dg), c_void_p(out.data_ptr() + 4 * nf_out * s0), _ptr(err), _stream(device))
                    _native.check(lib.ss_pair_features_grouped(*args), 'ss_pair_features_grouped')
            if strict and _take_error(device):
                raise IndexError(f'links refer to nodes outside [-{N}, {N})')
            return out, None
        if degrees is not None:
            if out is None:
                out = torch.empty((B, 2 * nf), dtype=torch.float32, device=device)
            with _Span('pair_features', device):
                _native.check(_native.lib().ss_pair_features_normalised(
                    _ptr(lk), B, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0), byref(params.struct), flags, _ptr(dg),
                    _ptr(out), _ptr(err), _stream(device)), 'ss_pair_features_normalised')
            if strict and B > 0 and _take_error(device):
                raise IndexError(f'links refer to nodes outside [-{N}, {N})')
            return out, None
        if out is None:
            out = torch.empty((B, nf), dtype=torch.float32, device=device)
        dbg = None
        if want_debug:
            dbg = {'match': torch.empty((B, h, h), dtype=torch.int32, device=device),
                   'zeros': torch.empty((B, h, h), dtype=torch.int32, device=device),
                   'inter': torch.empty((B, h, h), dtype=torch.float32, device=device)}
        with _Span('pair_features', device):
            _native.check(_native.lib().ss_pair_features(
                _ptr(lk), B, N, h, mh_ptrs, P, hll_ptrs, _ptr(cd), cd.stride(0), byref(params.struct), flags, _ptr(out),
                _ptr(dbg['match']) if dbg else None, _ptr(dbg['zeros']) if dbg else None,
                _ptr(dbg['inter']) if dbg else None, _ptr(err), _stream(device)), 'ss_pair_features')
        if strict and B > 0 and _take_error(device):
            raise IndexError(f'links refer to nodes outside [-{N}, {N})')
        return out, dbg

    def _get_intersections(self, edge_list, hash_table):
        """set-intersection estimates jaccard * union for every (k1, k2) (reference :167-189).
        @return: {(k1, k2): float32 [n_edges]} on edge_list.device"""
        _, dbg = self._pair_kernel(edge_list, hash_table, None, want_debug=True)
        inter = dbg['inter'].to(edge_list.device)
        return {(k1, k2): inter[:, k1 - 1, k2 - 1].contiguous()
                for k1 in range(1, self.max_hops + 1) for k2 in range(1, self.max_hops + 1)}

    def get_hashval(self, x):
        return x.hashvals

    def _linearcounting(self, num_zero):
        return self.m * torch.log(self.m / num_zero)

    def _estimate_bias_or_refine(self, e, refine):
        device = _compute_device(e)
        params = self._params(device)
        x = e.to(device=device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        _native.check(_native.lib().ss_estimate_bias(_ptr(x), x.numel(), byref(params.struct), _ptr(out), int(refine),
                                                     _stream(device)), 'ss_estimate_bias')
        return out.to(e.device)

    def _estimate_bias(self, e):
        """mean bias of the 6 table entries nearest to each estimate (reference :197-204)"""
        return self._estimate_bias_or_refine(e, False)

    def _refine_hll_count_estimate(self, estimate):
        """subtract the bias from estimates <= 5m, in place like the reference (:206-210)"""
        refined = self._estimate_bias_or_refine(estimate, True)
        estimate.copy_(refined)
        return estimate

    def hll_count(self, regs):
        """HLL++ cardinality estimate of each register row (reference :212-232).
        @param regs: integer tensor [n, m] (or [m])  @return: float32 [n] on regs.device"""
        if regs.dim() == 1:
            regs = regs.unsqueeze(dim=0)
        if regs.size(1) != self.m:
            raise ValueError(f'expected rows of {self.m} registers, got {regs.size(1)}')
        device = _compute_device(regs)
        tag = getattr(regs, '_ss_count', None)
        if tag is not None:  # produced together with `regs` by hll_prop (same kernel arithmetic); handed out once
            regs._ss_count = None
            if tag[0] == regs._version and tag[1].device == device and tag[1].numel() == regs.size(0):
                return tag[1] if regs.device == device else tag[1].to(regs.device)
        params = self._params(device)
        packed = _packed_hll_of(regs, device)
        out = torch.empty(regs.size(0), dtype=torch.float32, device=device)
        _native.check(_native.lib().ss_hll_count(_ptr(packed), regs.size(0), byref(params.struct), _ptr(out), 1,
                                                 _stream(device)), 'ss_hll_count')
        return out if regs.device == device else out.to(regs.device)

    def _hll_merge(self, src, dst):
        if src.shape != dst.shape:
            raise ValueError('source and destination register shapes must be the same')
        return torch.maximum(src, dst)

    def hll_neighbour_merge(self, root, neighbours):
        all_regs = torch.cat([root.unsqueeze(dim=0), neighbours], dim=0)
        return torch.max(all_regs, dim=0)[0]

    def minhash_neighbour_merge(self, root, neighbours):
        all_regs = torch.cat([root.unsqueeze(dim=0), neighbours], dim=0)
        return torch.min(all_regs, dim=0)[0]

    def jaccard(self, src, dst):
        """minhash Jaccard estimate of [n_edges, num_perm] hash-value tensors (reference :247-256)"""
        if src.shape != dst.shape:
            raise ValueError('source and destination hash value shapes must be the same')
        return torch.count_nonzero(src == dst, dim=-1) / self.num_perm

    def get_subgraph_features(self, links, hash_table, cards, batch_size=11000000, degrees=None, lazy=False, out=None):
        """structural features of node pairs: approximations of the number of nodes at distance (d_u, d_v)
        from (u, v), for the (d_u, d_v) listed in LABEL_LOOKUP[max_hops] (reference :258-323).
        @param links: int tensor [n_edges, 2] (or [2])
        @param hash_table: {hop: {'hll': [N, m], 'minhash': [N, num_perm]}} (a SketchTable or plain tensors)
        @param cards: float tensor [N, max_hops] of neighbourhood cardinality estimates
        @param batch_size: pairs per kernel launch (results do not depend on it)
        @param degrees: optional float tensor [N] (HashDataset.degrees, datasets/elph.py:74).  Extension beyond the
               reference signature: when given, BUDDY's degree-normalised copy (models/elph.py:276-293: feature /
               sqrt(d_u * d_v), NaN / Inf -> 0) is appended in the same kernel and the result is [n_edges, 2 * F].
        @param lazy: extension: return a `DeviceFeatureStore` (feature_store.py) instead of the tensor -- rows are computed on
               the GPU when a batch indexes it (runners/train.py:58-60), nothing of size [n_edges, F] is materialised
        @param out: extension: a contiguous float32 [n_edges, F] tensor ON THE COMPUTE DEVICE the rows are written into (and which
               is returned): a slice of a larger result -- dist.sharded_precompute lets every launch store straight into the block
               of the output the collective then sends from
        @return: float32 [n_edges, max_hops * (max_hops + 2)] on links.device"""
        if self.max_hops not in (1, 2, 3):
            raise NotImplementedError("Only 1, 2 and 3 hop hashes are implemented")
        if links.dim() == 1:
            links = links.unsqueeze(0)
        if lazy:
            from .feature_store import DeviceFeatureStore
            return DeviceFeatureStore(self, links, hash_table, cards, degrees=degrees, batch_size=batch_size)
        n = links.size(0)
        if knobs.GROUP_LINKS_MIN and n >= knobs.GROUP_LINKS_MIN and n < (1 << 31):
            feats, _ = self._pair_kernel(links, hash_table, cards, degrees=degrees, group_batch=max(int(batch_size), 1), out=out)
        elif n <= batch_size:
            feats, _ = self._pair_kernel(links, hash_table, cards, degrees=degrees, out=out)
        elif out is not None:
            for s in range(0, n, batch_size):
                self._pair_kernel(links[s:s + batch_size], hash_table, cards, degrees=degrees, out=out[s:s + batch_size])
            feats = out
        else:
            chunks = [self._pair_kernel(links[s:s + batch_size], hash_table, cards, degrees=degrees)[0]
                      for s in range(0, n, batch_size)]
            feats = torch.cat(chunks, dim=0)
        if feats.device == links.device or out is not None:
            return feats
        out = feats.to(links.device)
        # links on another device (BUDDY keeps them on the CPU): the copy back has waited for the launches, so the deferred
        # bounds report is final and can be raised from the offending call itself -- as the reference's CPU indexing does
        if self.strict_bounds == 'deferred':
            self._deferred.raise_if_set()
        return out

