"""Multi-GPU paths.  Query: edge-pair batches shard across ranks, sketch tables replicated.
Build (optional, pays off once a hop costs more than moving the table: ogbl-ppa / citation2 sizes): destination rows
shard across ranks, every hop ends with an in-place all-gather of the freshly written row blocks.

One process per GPU (torchrun); `torch.distributed` backend "nccl" is RCCL on ROCm.  Pairs are
independent (the reference maps over rows, hashing.py:272-321) so the only exchange is the gather of
the per-rank feature rows -- one all_gather_into_tensor of [ceil(L/G), h(h+2)] fp32 per call
(2 MiB per rank at B=65536, h=2: latency-bound on xGMI, no ring-vs-direct concern).
min/max and integer counts are order independent, so every rank's replicated table is bit-identical
and no reduction collective exists on the path.
"""
import logging
import os

import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


def shard_bounds(n_items, world_size, rank):
    """contiguous, order-preserving partition: rank r owns [lo, hi); shards differ by at most one chunk"""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def sharded_subgraph_features(compute, links, group=None):
    """every rank passes the SAME links [L, 2]; rank r computes `compute(links[lo:hi])` -> [hi-lo, F] and all
    ranks return the full [L, F] tensor in the original order.

    compute: callable(links_shard) -> float tensor on the communication device (e.g.
             lambda lk: eh.get_subgraph_features(lk, table, cards))"""
    if not (dist.is_available() and dist.is_initialized()):
        return compute(links)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    L = links.size(0)
    lo, hi = shard_bounds(L, world, rank)
    local = compute(links[lo:hi])
    per = (L + world - 1) // world
    F = local.size(1)
    padded = local.new_zeros((per, F))
    padded[:hi - lo] = local
    gathered = local.new_empty((world * per, F))
    _all_gather_rows(gathered, padded.contiguous(), group)
    return gathered[:L]


def _all_gather_rows(dst, src, group=None, async_op=False):
    """all_gather_into_tensor; backends without device collectives (gloo in the one-GPU tests) go through the host"""
    if not dst.is_cuda or dist.get_backend(group) == 'nccl':
        return dist.all_gather_into_tensor(dst, src, group=group, async_op=async_op)
    staged = torch.empty(dst.shape, dtype=dst.dtype, device='cpu')
    dist.all_gather_into_tensor(staged, src.cpu(), group=group)
    dst.copy_(staged)
    return None


def exchange_blocks_p2p(full, rank, world, per, group=None):
    """the all-gather of equally sized blocks (`per` leading rows / elements each, block r owned by rank r, in place in `full`)
    as world - 1 concurrent point-to-point transfers per rank: one send of the own block to every peer, one receive from
    every peer, all posted at once (batch_isend_irecv) -- on xGMI each lands on its own link.  Returns the request list
    (wait on all of them); synchronous backends (gloo on CPU tensors) complete before returning."""
    ops = []
    mine = full[rank * per:(rank + 1) * per]
    # `rank` / `world` are ranks WITHIN `group`; P2POp's positional peer is a GLOBAL rank: translate, so that a
    # sub-group (one node of a multi-node job) exchanges with its own members
    to_global = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    for step in range(1, world):
        dst, src = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, mine, to_global(dst), group))
        ops.append(dist.P2POp(dist.irecv, full[src * per:(src + 1) * per], to_global(src), group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    for r in reqs:
        r.wait()
    return reqs


_EXCHANGE_CHOICE = {}


def choose_exchange(device, group=None, block_bytes=32 << 20, reps=3):
    """which form of the per-hop block exchange is faster on THIS node, measured once per (process, group): the collective
    (`all_gather_into_tensor`) or world - 1 concurrent point-to-point transfers per rank (one per xGMI link; a ring all-gather is
    bound by ONE link).  Every rank times both on a block of `block_bytes` per rank, the MAX over ranks decides (all ranks agree by
    construction).  SS_EXCHANGE=all_gather|p2p forces a form (tests, A/B runs).  Returns 'all_gather' or 'p2p'."""
    forced = os.environ.get('SS_EXCHANGE')
    if forced in ('all_gather', 'p2p'):
        return forced
    key = (str(device), id(group))
    if key in _EXCHANGE_CHOICE:
        return _EXCHANGE_CHOICE[key]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    choice = 'all_gather'
    if world > 1:
        import time
        per = max(block_bytes // 4, 1)
        full = torch.zeros(world * per, dtype=torch.int32, device=device)
        cuda = torch.device(device).type == 'cuda'

        def sync():
            if cuda:
                torch.cuda.synchronize(device)

        def run(form):
            mine = full[rank * per:(rank + 1) * per]
            if form == 'p2p':
                exchange_blocks_p2p(full, rank, world, per, group)
            else:
                dist.all_gather_into_tensor(full, mine if cuda else mine.clone(), group=group)
            sync()
        times = {}
        for form in ('all_gather', 'p2p'):
            run(form)  # warm-up (connection set-up)
            dist.barrier(group=group)
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                run(form)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            times[form] = float(t.item())
        choice = 'p2p' if times['p2p'] < 0.9 * times['all_gather'] else 'all_gather'
        _EXCHANGE_CHOICE[(key, 'times')] = {k: v / reps for k, v in times.items()}
    _EXCHANGE_CHOICE[key] = choice
    return choice


def exchange_probe_times(device, group=None):
    """seconds per exchange of the probe block for both forms (None before choose_exchange ran or when a form was forced)"""
    return _EXCHANGE_CHOICE.get(((str(device), id(group)), 'times'))


class RowShard(object):
    """destination-row partition of one build across the ranks of `group` + the exchange that follows every hop.

    Rank r owns rows [r * per, min((r + 1) * per, num_nodes)); tables are allocated with `padded_rows` = per * world
    rows so that every rank contributes an equally sized block and `all_gather_into_tensor` can run IN PLACE (input =
    the owned block of the output).  xGMI is point-to-point: RCCL's all-gather of N*R/G bytes per rank uses all 7 links
    of a GPU at once; no ring over a single link is forced by this call pattern.  With the nccl (RCCL) backend the
    gather is asynchronous (its own stream): `ElphHashes._build` overlaps the MinHash exchange with the HLL kernel and
    vice versa.  Other backends (gloo in the tests) stage device tensors through the host."""

    def __init__(self, num_nodes, group=None, exchange=None):
        """exchange: 'all_gather' | 'p2p' | None (= SS_EXCHANGE if set, else the collective; sharded_build_hash_tables passes the
        form choose_exchange measured)"""
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.per = max((num_nodes + self.world - 1) // self.world, 1)
        self.padded_rows = self.per * self.world
        lo = min(self.rank * self.per, num_nodes)
        self.rows = (lo, min(lo + self.per, num_nodes))
        self.native = dist.get_backend(group) == 'nccl'
        # SS_EXCHANGE=p2p: G - 1 concurrent point-to-point transfers per rank instead of one all_gather (tools/probe_allgather.py
        # decides which is faster on a given node: a ring all-gather is bound by one xGMI link)
        self.p2p = (exchange or os.environ.get('SS_EXCHANGE', 'all_gather')) == 'p2p'

    def block(self, full):
        return full[self.rank * self.per:(self.rank + 1) * self.per]

    def gather(self, full):
        """every rank's owned block of `full` ([padded_rows, ...], contiguous) reaches every other rank"""
        assert full.size(0) == self.padded_rows and full.is_contiguous()
        if self.p2p and (self.native or not full.is_cuda):
            exchange_blocks_p2p(full, self.rank, self.world, self.per, self.group)  # stream-ordered on RCCL; blocking on gloo
            return None
        if self.native or not full.is_cuda:
            mine = self.block(full) if self.native else self.block(full).clone()  # RCCL gathers in place
            return dist.all_gather_into_tensor(full, mine, group=self.group, async_op=self.native)
        staged = torch.empty(full.shape, dtype=full.dtype, device='cpu')
        dist.all_gather_into_tensor(staged, self.block(full).cpu(), group=self.group)
        full.copy_(staged)
        return None

    @staticmethod
    def wait(handle):
        if handle is not None:
            handle.wait()  # the current stream waits for the collective; the host does not block


class PeerShard(RowShard):
    """the row-sharded build WITHOUT an exchange step (DESIGN 6 'rows written straight into the peers' tables'):
    every rank's kernels store each row they finish into ALL ranks' tables while they run -- the same bytes over xGMI as the
    per-hop all-gather, spread over the whole kernel, no collective launch -- and a hop boundary needs only a cross-rank barrier.

    The tables are ONE slab per shard, shared through a CUDA-IPC handle (hipIpcGetMemHandle under torch's reductions;
    HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver) and pooled per shape for the life of the process: a build through this shard
    returns VIEWS of that slab, which the shard's next build overwrites (BUDDY builds once; a caller that wants to keep several
    builds alive makes several shards).
    At most 8 ranks (SS_MAX_MIRRORS + 1: one node)."""
    peer_write = True

    def __init__(self, num_nodes, max_hops, num_perm, m, device, group=None):
        super().__init__(num_nodes, group)
        from . import _native
        if self.world - 1 > _native.MAX_MIRRORS:
            raise ValueError(f'a peer-write build spans at most {_native.MAX_MIRRORS + 1} ranks, got {self.world}')
        self.device = torch.device(device)
        self.shape = (max_hops, num_perm, m)
        rows = self.padded_rows
        self._token = torch.zeros(1, dtype=torch.float32, device=device)
        # ONE slab per shard, carved into the 2 h + 1 tables: one allocation, one IPC handle to export, one mapping per peer (round 4
        # exported every table by itself: seven handles per shard and rank at h = 3)
        self._carve = _SlabLayout(rows, max_hops, num_perm, m)
        # Slabs are POOLED per (group, size, device) and never handed back to torch's allocator: a shard that is dropped leaves its
        # slab -- exported once, mapped once by every peer -- to the next shard of the same shape.  Why: tools/stress_multiproc.sh
        # (eight processes on one GPU, 20 passes of two shards each) fails in 1-2 of 10 launches, never before the 13th pass, always
        # inside the IPC layer under torch's reductions and only on RECYCLED allocations -- `hipIpcGetMemHandle: invalid argument`
        # on export, or (once the export was a single slab) a handle that opens on the peers but does not reach this rank's memory,
        # which the store-and-read-back probe below catches (profiles/round5_mp_stress_*.txt).  That is THE intermittent failure of
        # test_sharded_build_two_ranks_one_gpu[8] of rounds 3-4; no kernel of the library is involved.  Not recycling is the cure
        # that is ours to apply; what still fails is retried as a whole on fresh memory and, failing that, raised on every rank
        # (fallback=True of peer_write_build_hash_tables then takes the exchange form).
        key = (id(group), self._carve.bytes, str(self.device))
        entry, errors = None, []
        for attempt in range(3):
            try:
                entry = self._pooled_or_new(key, max_hops)
                break
            except RuntimeError as exc:  # (agreed: every rank is here)
                errors.append(str(exc))
                logger.warning('PeerShard rank %d, attempt %d: %s', self.rank, attempt + 1, exc)
                import gc
                import time
                gc.collect()
                torch.cuda.ipc_collect()
                time.sleep(0.2 * (attempt + 1))
        if entry is None:
            raise RuntimeError('peer-write build unavailable after 3 attempts: ' + ' | '.join(errors))
        self._entry = entry
        self._slab, self._peer_tensors = entry['slab'], entry['peers']
        self.mh, self.hll, self.cards = self._carve.views(self._slab)
        self._order = [r for r in range(self.world) if r != self.rank]
        import weakref
        # (the finaliser runs BEFORE the shard's own attributes are cleared: its views of the slab are still alive then, so "no tensor
        # of a build is left" means "the storage has as many users as right now")
        weakref.finalize(self, _release_entry, key, entry, _storage_users(self._slab))  # the slab and its mappings outlive the shard
        self.generation = 0
        self.hop_barrier()  # nobody starts storing into a table before everybody has mapped (and probed) it

    def _pooled_or_new(self, key, max_hops):
        """a pool entry {id, slab, peers: {rank: (peer slab, [views])}} every rank agrees on -- a free one of an earlier shard when
        ALL ranks hold the same one free, else a new one (allocate, export, exchange, map) -- probed before it is returned.
        Raises RuntimeError on every rank or on none."""
        free = _POOL.setdefault(key, [])
        everyone = [None] * self.world
        dist.all_gather_object(everyone, sorted(e['id'] for e in free), group=self.group)
        common = set(everyone[0]).intersection(*map(set, everyone[1:]))
        if common:
            pick = min(common)
            entry = next(e for e in free if e['id'] == pick)
            free.remove(entry)
        else:
            entry = self._new_entry(key)
        try:
            self._probe(entry, max_hops)
        except RuntimeError:
            entry['bad'] = True  # (kept referenced by _QUARANTINE: its memory must not come back through the allocator either)
            _QUARANTINE.append(entry)
            raise
        return entry

    def _new_entry(self, key):
        slab, mine, failure = None, None, None
        try:
            slab = torch.empty(self._carve.bytes, dtype=torch.uint8, device=self.device)
            mine = _export_tables([slab])
        except Exception as exc:  # noqa: BLE001
            failure = RuntimeError(f'IPC export failed: {type(exc).__name__}: {str(exc).splitlines()[0]} [{_fd_state()}; {_ipc_probe()}]')
            if slab is not None:
                _QUARANTINE.append({'slab': slab})
        self._agree(failure, 'a rank could not export its tables')
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        peers, failure = {}, None
        try:
            for r in range(self.world):
                if r == self.rank:
                    continue
                (rebuild, args), = everyone[r]
                peer_slab = rebuild(*args)
                if peer_slab.device != self.device:  # another GPU of the node: this GPU must be allowed to address its memory
                    # asked BEFORE anything touches the mapping: a kernel (or copy) that stores through a mapping its GPU
                    # cannot address faults the whole process (hipDeviceCanAccessPeer)
                    if not torch.cuda.can_device_access_peer(self.device.index or 0, peer_slab.device.index or 0):
                        raise RuntimeError(f'{self.device} cannot address the memory of {peer_slab.device} (no peer access)')
                    _enable_peer_access(self.device, peer_slab.device)
                pm, ph, pc = self._carve.views(peer_slab)
                peers[r] = (peer_slab, pm + ph + [pc])
        except Exception as exc:  # (mapping a peer's memory can fail on ONE rank only: agree before anybody waits for anybody)
            failure = exc
            _QUARANTINE.append({'slab': slab, 'peers': peers})
        self._agree(failure, 'a rank could not map its peers\' tables')
        _POOL_IDS[key] = _POOL_IDS.get(key, 0) + 1  # (creations are collective: the counter agrees across ranks)
        return {'id': _POOL_IDS[key], 'slab': slab, 'peers': peers}

    def _probe(self, entry, max_hops):
        """store-and-read-back through every mirror: rank r writes a fresh token into row r of every PEER's cards table (column 0), a
        barrier, then every rank checks the rows its peers wrote -- a mapping that opened but does not reach the peer's memory is
        found here, on every rank or on none, not inside a kernel of the first build"""
        self.hop_barrier()  # (a pooled slab: whatever any rank still has queued on the tables of the shard that held it before completes first)
        entry['probes'] = entry.get('probes', 0) + 1
        token = float(1000 * entry['probes'])  # (a reused slab still holds the tokens of its last probe)
        _, _, cards = self._carve.views(entry['slab'])
        failure = None
        try:
            for r, (_, views) in entry['peers'].items():
                views[2 * max_hops][self.rank, 0] = token + self.rank + 1
            cards[self.rank, 0] = token + self.rank + 1
            self.hop_barrier()
            got = cards[:self.world, 0].cpu()
            want = torch.arange(1, self.world + 1, dtype=torch.float32) + token
            if not torch.equal(got, want):
                raise RuntimeError(f'probe stores of the peers did not arrive: rows {got.tolist()} (want {want.tolist()})')
        except Exception as exc:  # noqa: BLE001
            failure = exc
        self._agree(failure, 'the store-and-read-back probe through the mirrors failed')

    def _agree(self, failure, what):
        """all ranks raise, or none (a rank that failed alone would leave the others waiting in a collective)"""
        ok = torch.tensor([0.0 if failure else 1.0], device=self.device if self.native else 'cpu')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if float(ok.item()) < 1.0:
            raise RuntimeError(f'peer-write build: {what} (this rank: {failure!r})')

    def tables(self, max_hops, num_perm, m, device):
        """the shard's buffers for ONE build.  They are shared with every rank, and every rank's kernels of the new build store into
        them: a cross-rank barrier first, in stream order -- whatever this rank (or a lagging peer) still has queued on the tables
        of the PREVIOUS build through this shard (a query, a copy) completes before anybody's first store of the new one
        (a rank still reading the old rows would otherwise see them change under it).  `generation` counts the builds:
        views handed out by an earlier build are stale once it moves."""
        if (max_hops, num_perm, m) != self.shape or torch.device(device) != self.device:
            raise ValueError(f'this shard was made for {self.shape} on {self.device}')
        self.hop_barrier()
        self.generation += 1
        return self.mh, self.hll, self.cards

    def mirrors(self, kind, k):
        """device addresses, inside the OTHER ranks' memory, of the table this rank's launch for (`kind`, hop index k) also writes"""
        h = self.shape[0]
        out = []
        for r in self._order:
            t = self._peer_tensors[r][1]
            if kind == 'mh':
                out.append(t[k].data_ptr())
            elif kind == 'hll':
                out.append(t[h + k].data_ptr())
            else:
                out.append(t[2 * h].data_ptr() + 4 * k)  # column k of the peer's cards [rows, h]
        return out

    def hop_barrier(self):
        """every rank's launches issued so far are complete (their stores into this rank's tables included) before this rank's
        NEXT launch starts: RCCL -- a one-element all-reduce on its own stream, the compute stream waits for it, the host does not
        block; other backends (gloo in the one-GPU tests) -- device synchronisation + a host barrier"""
        if self.native:
            dist.all_reduce(self._token, group=self.group, async_op=True).wait()
        else:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)


_POOL = {}        # (group, slab bytes, device) -> free pool entries (see PeerShard.__init__)
_POOL_IDS = {}    # same key -> entries created so far
_QUARANTINE = []  # slabs / mappings of constructions that failed: never reused, never freed (their memory must not be recycled)


def _storage_users(t):
    """tensors / storage handles that share t's memory right now (the temporary handle of this call included)"""
    return torch._C._storage_Use_Count(t.untyped_storage()._cdata)


def _release_entry(key, entry, users_of_the_shard_itself):
    """a shard has been dropped: its slab goes back to the pool -- unless tensors handed out by its builds are still alive (the
    caller kept a table but not the shard): that slab is then kept aside for good, a later shard must not overwrite the table.
    users_of_the_shard_itself: _storage_users(slab) when the shard stood complete (the slab + the shard's own views)"""
    try:
        in_use = _storage_users(entry['slab']) > users_of_the_shard_itself
    except Exception:  # noqa: BLE001  (interpreter shutdown, a torch without the hook)
        in_use = True
    (_QUARANTINE if in_use else _POOL.setdefault(key, [])).append(entry)


class _SlabLayout(object):
    """where the 2 h + 1 tables of a shard lie inside its one allocation (every table 256-byte aligned)"""

    def __init__(self, rows, max_hops, num_perm, m):
        self.rows, self.h, self.P, self.m = rows, max_hops, num_perm, m
        al = lambda x: (x + 255) & ~255
        self.mh_bytes, self.hll_bytes, self.cards_bytes = al(rows * num_perm * 4), al(rows * m), al(rows * max_hops * 4)
        self.bytes = max_hops * (self.mh_bytes + self.hll_bytes) + self.cards_bytes

    def views(self, slab):
        off, mh, hll = 0, [], []
        for _ in range(self.h):
            mh.append(slab[off:off + self.rows * self.P * 4].view(torch.int32).view(self.rows, self.P))
            off += self.mh_bytes
        for _ in range(self.h):
            hll.append(slab[off:off + self.rows * self.m].view(self.rows, self.m))
            off += self.hll_bytes
        cards = slab[off:off + self.rows * self.h * 4].view(torch.float32).view(self.rows, self.h)
        return mh, hll, cards


def _ipc_probe():
    """does hipIpcGetMemHandle work at all in this process right now?  (diagnostics of a failed export: a raw hipMalloc, outside
    torch's allocator)"""
    try:
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        ptr, handle = ctypes.c_void_p(), (ctypes.c_char * 64)()
        rc_m = hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(1 << 20))
        rc_h = hip.hipIpcGetMemHandle(handle, ptr) if rc_m == 0 else -1
        if rc_m == 0:
            hip.hipFree(ptr)
        return f'raw hipMalloc rc {rc_m}, hipIpcGetMemHandle on it rc {rc_h}'
    except Exception as exc:  # noqa: BLE001
        return f'raw IPC probe failed: {exc!r}'


def _export_tables(tensors):
    """CUDA-IPC handles (torch's picklable rebuild recipes) of device tensors"""
    from torch.multiprocessing.reductions import reduce_tensor
    return [reduce_tensor(t) for t in tensors]


def _fd_state():
    """open file descriptors of this process against its limit (dmabuf IPC hands memory around as file descriptors)"""
    try:
        import resource
        return f'{len(os.listdir("/proc/self/fd"))} open fds, RLIMIT_NOFILE {resource.getrlimit(resource.RLIMIT_NOFILE)}'
    except Exception:  # noqa: BLE001
        return 'fd state unknown'


def _enable_peer_access(device, other):
    """make `other`'s memory addressable from kernels running on `device` (and the reverse): torch enables peer access between two
    devices the first time it copies between them"""
    a = torch.zeros(1, device=device)
    b = a.to(other)
    a.copy_(b)
    torch.cuda.synchronize(device)


def peer_write_build_hash_tables(eh, num_nodes, edge_index, shard=None, group=None, fallback=False):
    """`eh.build_hash_tables` with the rows of every hop computed once across the group and written straight into every rank's
    tables (PeerShard).  Pass the shard of an earlier call to reuse its IPC-shared buffers (the tables of that earlier build are
    overwritten: they are the same memory); returns (table, cards, shard).  fallback=True: if the ranks cannot map or reach each
    other's tables (PeerShard's constructor fails on every rank or on none), the exchange form (RowShard) builds instead and
    `shard` comes back as None."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        table, cards = eh.build_hash_tables(num_nodes, edge_index)
        return table, cards, None
    if shard is None:
        device = edge_index.device if edge_index.is_cuda else torch.device('cuda', torch.cuda.current_device())
        try:
            shard = PeerShard(num_nodes, eh.max_hops, eh.num_perm, eh.m, device, group)
        except RuntimeError:
            if not fallback:
                raise
            table, cards = sharded_build_hash_tables(eh, num_nodes, edge_index, group)
            return table, cards, None
    table, cards = eh._build(num_nodes, edge_index, shard)
    return table, cards, shard


def sharded_build_hash_tables(eh, num_nodes, edge_index, group=None):
    """`eh.build_hash_tables(num_nodes, edge_index)` with the rows of every hop computed once across the group instead
    of once per rank; every rank passes the SAME edge_index and returns the SAME full (table, cards) -- bit-identical to
    the unsharded result (min / max / integer counts are order independent, the cardinality of a row is computed by
    the rank that owns it with the same arithmetic)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return eh.build_hash_tables(num_nodes, edge_index)
    device = edge_index.device if edge_index.is_cuda else (torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else edge_index.device)
    exchange = choose_exchange(device, group) if dist.get_backend(group) == 'nccl' else None  # (a startup micro-probe, not an env var)
    return eh._build(num_nodes, edge_index, RowShard(num_nodes, group, exchange))


# ---- workload bookkeeping shared by bench.py and the gloo tests ------------------------------------------------------
class BatchPlan(object):
    """which pairs of a step a rank owns, and what a step counts for.

    weak   : per-GPU work fixed -- every rank has its OWN `batch` pairs (the caller seeds them per rank); a step processes
             world * batch pairs.  With a replicated build the N-GPU rate is ~N x the 1-GPU rate BY CONSTRUCTION (every rank
             repeats the build); it measures the collective machinery, not the sharding.
    strong : total work fixed -- ONE global batch of `batch` pairs (BASELINE configs[3] / [4]: "edge-batch sharded across
             8"), rank r owns the contiguous slice shard_bounds(batch, world, r); a step processes `batch` pairs whatever
             the world size, so value(N) / value(1) is the speed-up of the same job."""

    def __init__(self, scaling, world, rank, batch):
        if scaling not in ('weak', 'strong'):
            raise ValueError(f'scaling must be weak or strong, got {scaling}')
        self.scaling, self.world, self.rank, self.batch = scaling, world, rank, batch
        if scaling == 'weak':
            self.lo, self.hi = 0, batch
            self.rows_per_rank = batch
            self.pairs_per_step = world * batch
            self.links_seed = 2 + rank  # every rank draws its own pairs
        else:
            self.lo, self.hi = shard_bounds(batch, world, rank)
            self.rows_per_rank = (batch + world - 1) // world  # gather block (the last ranks pad)
            self.pairs_per_step = batch
            self.links_seed = 2         # every rank draws the SAME global batch and slices it
        self.local_pairs = self.hi - self.lo

    def local(self, links):
        """this rank's slice of the batch tensor it generated with `links_seed`"""
        return links if self.scaling == 'weak' else links[self.lo:self.hi]

    def unpad(self, gathered):
        """[world * rows_per_rank, F] as gathered -> the rows of the step in order (strong: drops the padding rows)"""
        if self.scaling == 'weak' or self.world * self.rows_per_rank == self.batch:
            return gathered
        per, keep = self.rows_per_rank, []
        for r in range(self.world):
            lo, hi = shard_bounds(self.batch, self.world, r)
            keep.append(gathered[r * per:r * per + (hi - lo)])
        return torch.cat(keep, dim=0)


class AsyncFeatureGather(object):
    """per-batch all_gather of the feature rows, issued asynchronously (RCCL: on its own stream, under the next step's
    kernels) into `depth` rotating output buffers; collectives of one group run in issue order, so the buffer written
    `depth` gathers ago is free again without the compute stream ever waiting.  drain() completes everything (bench.py
    calls it inside the timed region, before the closing fence).  Works unchanged on gloo / CPU tensors (tests)."""

    def __init__(self, plan, num_features, device, dtype=torch.float32, depth=2, group=None):
        self.plan, self.group, self.depth = plan, group, depth
        self.active = dist.is_available() and dist.is_initialized()
        self.out = [torch.empty((plan.world * plan.rows_per_rank, num_features), dtype=dtype, device=device)
                    for _ in range(depth)] if self.active else None
        self.pad = ([torch.zeros((plan.rows_per_rank, num_features), dtype=dtype, device=device) for _ in range(depth)]
                    if self.active and plan.scaling == 'strong' else None)
        self.slot_work = [None] * depth  # the gather that last used slot i (its pad as input, its buffer as output)
        self.works, self.issued = [], 0

    def __call__(self, feats):
        """feats: this rank's [local_pairs, F]; returns the buffer the gather lands in (valid after drain / wait)"""
        if not self.active:
            return feats
        slot = self.issued % self.depth
        if self.slot_work[slot] is not None:
            self.slot_work[slot].wait()  # stream-ordered: the current stream waits for that gather, the host does not block
        src = feats
        if feats.size(0) != self.plan.rows_per_rank:  # strong scaling, ragged last shard: pad to the common block size
            self.pad[slot][:feats.size(0)].copy_(feats)
            src = self.pad[slot]
        dst = self.out[slot]
        work = _all_gather_rows(dst, src.contiguous(), self.group, async_op=True)
        if work is None:  # (host-staged: already complete)
            self.issued += 1
            self.slot_work[slot] = None
            return dst
        self.works.append((work, src))
        self.slot_work[slot] = work
        self.issued += 1
        if len(self.works) > 2 * self.depth:  # host-side bookkeeping only: the oldest ones finished steps ago
            self.works.pop(0)[0].wait()
        return dst

    def drain(self):
        for work, _ in self.works:
            work.wait()
        self.works.clear()
