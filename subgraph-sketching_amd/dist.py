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


class LinkRounds(object):
    """which links of a set of L a rank computes when the rows are to be GATHERED, and where their rows lie.

    The set is cut into ROUNDS of `world * block` consecutive links; inside a round rank r owns the r-th block.  The rows of a round
    are then one contiguous piece of the output IN THE CALLER'S ORDER with rank r's rows as its r-th equal part -- exactly the layout
    `all_gather_into_tensor` produces, so every round is gathered IN PLACE into its final position (input = the owned block of the
    output) as soon as its launches are queued, while the next round's launches run: no staging buffer, no padding copy, no
    permutation afterwards.  The last, shorter round takes blocks of ceil(tail / world) links (its last blocks may be short or
    empty; the output is allocated with the few padding rows that needs and returned without them).
    (reference: get_subgraph_features is a map over rows, hashing.py:258-323; datasets/elph.py:200-213 computes every link of a
    split in one call and keeps the rows in one tensor)"""

    def __init__(self, n_links, world, rank, block):
        if block < 1:
            raise ValueError('block must be positive')
        self.n_links, self.world, self.rank, self.block = n_links, world, rank, block
        self.rounds = []  # (first link of the round, links per block)
        span = world * block
        full = n_links // span
        self.rounds = [(c * span, block) for c in range(full)]
        tail = n_links - full * span
        if tail:
            self.rounds.append((full * span, (tail + world - 1) // world))
        self.padded_links = (self.rounds[-1][0] + world * self.rounds[-1][1]) if self.rounds else 0

    def owned(self, c, rank=None):
        """[lo, hi) of the links rank `rank` (default: this one) computes in round c -- hi - lo < links-per-block only in the last round"""
        base, per = self.rounds[c]
        r = self.rank if rank is None else rank
        lo = min(base + r * per, self.n_links)
        return lo, min(lo + per, self.n_links)

    def owned_count(self, rank=None):
        return sum(hi - lo for lo, hi in (self.owned(c, rank) for c in range(len(self.rounds))))

    def owned_index(self, device=None, rank=None):
        """int64 [owned_count]: the positions in the caller's link set of the rows this rank computes, round by round"""
        parts = [torch.arange(lo, hi, dtype=torch.int64, device=device) for lo, hi in (self.owned(c, rank) for c in range(len(self.rounds)))]
        return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=device)


def default_link_block(n_links, world):
    """links per block of LinkRounds when the caller names none: a QUARTER of a rank's share (three quarters of the gather then run
    under later rounds' launches), at least 2^21 and at most 2^24 links.  Why not smaller: the query groups every block by first
    node on its own (knobs.GROUP_LINKS_MIN), and a block holds a source far fewer times than the share does -- measured on one
    rank's share of a BUDDY link set at 8 ranks, random pairs (tools/probe_link_blocks.py, profiles/round6_link_blocks.txt):
    ogbl-citation2 size (44.5 M links) whole share 29.9 ms, blocks of 11 M 32.3, 5.6 M 35.8, 1 M 46.7; ogbl-ppa size (7.2 M) 2.75 /
    4 M 2.82 / 2 M 3.13 / 1 M 3.62 ms.  Lists that already have their runs (ogbl-citation2's evaluation sets: 1 000 negatives per
    source, listed together) lose 1-2 % at any of these sizes."""
    share = (n_links + world - 1) // max(world, 1)
    return int(min(max((share + 3) // 4, 1 << 21), 1 << 24))


class ShardedFeatures(object):
    """what sharded_precompute returns.  rows: the [L, F] tensor in the caller's order where the gather mode puts it (every rank /
    group rank 0 / nowhere), else None; local: this rank's own rows [owned, F] (`index[i]` = position of row i in the caller's
    link set) -- what a data-parallel training loop that keeps its shard of the links reads (runners/train.py:58-60 indexes the
    feature tensor per batch); for a rank that holds `rows`, `local` is gathered from it on first use."""

    def __init__(self, rows, local, index_fn):
        self.rows, self._local, self._index_fn, self._index = rows, local, index_fn, None

    @property
    def index(self):
        if self._index is None:
            self._index = self._index_fn()
        return self._index

    @property
    def local(self):
        if self._local is None and self.rows is not None:
            self._local = self.rows[self.index.to(self.rows.device)]
        return self._local


GATHER_MODES = ('all', 'rank0', 'none')


def _sharded_rows(fill, links, width, device, group, gather, block, dtype=torch.float32):
    """the machinery of sharded_precompute / sharded_subgraph_features.  fill(link_slice, out_view): compute the rows of the
    slice into out_view ([len, width], contiguous, on `device`; launches are queued on the current stream, nothing waits)."""
    if gather not in GATHER_MODES:
        raise ValueError(f'gather must be one of {GATHER_MODES}, got {gather!r}')
    L = links.size(0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if gather == 'none':
        # nothing is exchanged: a contiguous share per rank -- the query then groups the WHOLE share by first node at once (every
        # repeat of a source inside the share is found, not only those inside a block)
        lo, hi = shard_bounds(L, world, rank)
        local = torch.empty((hi - lo, width), dtype=dtype, device=device)
        fill(links[lo:hi], local)
        return ShardedFeatures(None, local, lambda: torch.arange(lo, hi, dtype=torch.int64, device=device))
    plan = LinkRounds(L, world, rank, block or default_link_block(L, world))
    native = dist.get_backend(group) == 'nccl'
    on_device = torch.device(device).type == 'cuda'
    holds_all = gather == 'all' or rank == 0
    # ranks that end up with every row compute theirs straight into the final tensor; the others (gather = 'rank0') into a
    # buffer of their own rows only
    full = torch.empty((plan.padded_links, width), dtype=dtype, device=device) if holds_all else None
    own = None if holds_all else torch.empty((plan.owned_count() + (plan.rounds[-1][1] if plan.rounds else 0), width), dtype=dtype, device=device)
    root = 0 if group is None else dist.get_global_rank(group, 0)
    works, own_at = [], 0
    for c, (base, per) in enumerate(plan.rounds):
        lo, hi = plan.owned(c)
        if holds_all:
            block_view = full[base + rank * per:base + (rank + 1) * per]
        else:
            block_view = own[own_at:own_at + per]
            own_at += hi - lo
        if hi > lo:
            fill(links[lo:hi], block_view[:hi - lo])
        piece = full[base:base + world * per] if holds_all else None
        # the round's exchange, issued at once: over RCCL it runs on RCCL's stream behind THIS round's launches (the collective
        # waits for what the current stream holds at this point) and under the next rounds' -- the host does not block
        # (host tensors on gloo take the SAME deferred form -- issued now, awaited after the last round -- so that the CPU tests walk the
        # control flow RCCL runs; only device tensors on a backend without device collectives are staged through the host, synchronously)
        deferred = native or not on_device
        if gather == 'all':
            if deferred:
                src = block_view if native else block_view.clone()  # (RCCL gathers in place: the input IS the owned block of the output)
                works.append((dist.all_gather_into_tensor(piece, src, group=group, async_op=True), src))
            else:
                staged = torch.empty(piece.shape, dtype=dtype, device='cpu')
                dist.all_gather_into_tensor(staged, block_view.cpu(), group=group)
                piece.copy_(staged)
        else:  # 'rank0'
            if deferred:
                dests = [piece[q * per:(q + 1) * per] for q in range(world)] if rank == 0 else None
                # (the root's own block is also dests[0]: it sends from a copy, so that no backend is handed one buffer as input AND output)
                src = block_view if (native and rank != 0) else block_view.clone()
                works.append((dist.gather(src, dests, dst=root, group=group, async_op=True), src))
            else:
                dests = [torch.empty((per, width), dtype=dtype, device='cpu') for _ in range(world)] if rank == 0 else None
                dist.gather(block_view.cpu(), dests, dst=root, group=group)
                if rank == 0:
                    piece.copy_(torch.cat(dests, dim=0))
    for w, _keep_alive in works:
        w.wait()  # RCCL: the current stream waits for the collectives, the host does not block
    rows = full[:L] if holds_all else None
    local = None if holds_all else own[:plan.owned_count()]
    return ShardedFeatures(rows, local, lambda: plan.owned_index(device))


def sharded_precompute(eh, links, hash_table, cards, group=None, gather='all', block=None, batch_size=11000000, degrees=None):
    """BUDDY's feature precompute (reference datasets/elph.py:200-213: `get_subgraph_features(links, hashes, cards, batch_size)` over
    every link of a split) across the ranks of `group`: every rank passes the SAME links and holds the same (replicated) tables.

    gather: who ends up with the rows --
      'none'  nobody exchanges anything: rank r computes a contiguous share and keeps it (`.local`, `.index`).  What a data-parallel
              training loop needs (each rank trains on its shard of the links); the only mode whose cost is compute / world.
      'all'   every rank ends with the full [L, F] tensor in the caller's order (`.rows`) -- the reference's single-process
              semantics.  (world - 1) / world of L * F * 4 bytes arrive at every rank (ogbl-citation2: 18.7 of 21.4 GB), so the
              exchange is NOT one collective after the job: the link set is walked in rounds (LinkRounds), each round's rows are
              all-gathered in place while the next round's launches run.
      'rank0' only group rank 0 ends with the tensor (the rank that writes the feature cache, datasets/elph.py:212-213); the same
              bytes arrive there, the other ranks receive nothing.
    block: links per rank and round (default_link_block).  Rows are bit-identical to the unsharded call (a pair's features depend
    on its own sketch rows only).  Returns a ShardedFeatures."""
    if links.dim() == 1:
        links = links.unsqueeze(0)
    nf = eh.max_hops * (eh.max_hops + 2)
    width = 2 * nf if degrees is not None else nf
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        rows = eh.get_subgraph_features(links, hash_table, cards, batch_size=batch_size, degrees=degrees)
        n = links.size(0)
        return ShardedFeatures(rows if gather != 'none' else None, rows, lambda: torch.arange(n, dtype=torch.int64, device=rows.device))
    from ._runtime import _compute_device
    first = hash_table.get(1) if hasattr(hash_table, 'get') else None
    device = _compute_device(links, getattr(first, 'mh_u32', None), cards)
    lk = links.to(device=device, dtype=torch.int64).contiguous()

    def fill(link_slice, out_view):
        eh.get_subgraph_features(link_slice, hash_table, cards, batch_size=batch_size, degrees=degrees, out=out_view)
    return _sharded_rows(fill, lk, width, device, group, gather, block)


def sharded_subgraph_features(compute, links, group=None, gather='all', block=None):
    """the same sharding around ANY row function: every rank passes the SAME links [L, 2]; `compute(link_slice) -> [len, F]` is
    called for the slices this rank owns.  gather = 'all' (default): every rank returns the full [L, F] tensor in the original
    order; 'rank0': group rank 0 does, the others return None; 'none': every rank returns its own contiguous share
    (links[shard_bounds(L, world, rank)]).  See sharded_precompute for the engine's own form (rows stored in place)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or links.size(0) == 0:
        return compute(links)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    L = links.size(0)
    block = block or max((L + world - 1) // world, 1)  # (one round: the contiguous shares of rounds 1-5, one compute call per rank)
    # width / dtype / device of the rows are only known from a result: every rank computes the first slice it owns (a rank that
    # owns none learns them from the others)
    lo, hi = shard_bounds(L, world, rank) if gather == 'none' else LinkRounds(L, world, rank, block).owned(0)
    first = compute(links[lo:hi]) if hi > lo else None
    seen = [None] * world
    dist.all_gather_object(seen, None if first is None else (first.size(1), first.dtype, str(first.device)), group=group)
    width, dtype, device = next(x for x in seen if x is not None)
    state = {'first': first}

    def fill(link_slice, out_view):
        cached, state['first'] = state['first'], None
        out_view.copy_(cached if cached is not None else compute(link_slice))
    res = _sharded_rows(fill, links, width, torch.device(device) if first is None else first.device, group, gather, block, dtype=dtype)
    return res.local if gather == 'none' else res.rows


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
        # keyed on the LAYOUT, not only on the slab size: a pooled entry caches the peers' views carved with the creating shard's layout
        # (two shapes of equal bytes would aim the mirrors at the old offsets: ADVICE r5); the group object is pinned by the pool so
        # that its id cannot be handed to another group after a destroy / re-init
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        _POOL_GROUPS[id(pg)] = pg
        key = (id(pg), rows, max_hops, num_perm, m, str(self.device))
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
        self._slabs, self._peer_tensors = entry['slabs'], entry['peers']
        self.mh, self.hll, self.cards = self._carve.views(self._slabs)
        self._order = [r for r in range(self.world) if r != self.rank]
        import weakref
        # (the finaliser runs BEFORE the shard's own attributes are cleared: its views of the slab are still alive then, so "no tensor
        # of a build is left" means "the storage has as many users as right now")
        try:
            users = _storage_users(self._slabs)
        except Exception as exc:  # noqa: BLE001  (a torch without the private hook: the constructor fails the way fallback=True expects)
            _QUARANTINE.append(entry)
            raise RuntimeError(f'peer-write build unavailable: cannot count the users of a slab ({exc!r})')
        weakref.finalize(self, _release_entry, key, entry, users)  # the slab and its mappings outlive the shard
        self.generation = 0
        self.verified, self.digests, self.failed = False, None, None  # (peer_write_build_hash_tables: replicas compared after the first build)
        self.hop_barrier()  # nobody starts storing into a table before everybody has mapped (and probed) it

    def _pooled_or_new(self, key, max_hops):
        """a pool entry {id, slab, peers: {rank: (peer slab, [views])}} every rank agrees on -- a free one of an earlier shard when
        ALL ranks hold the same one free, else a new one (allocate, export, exchange, map) -- probed before it is returned.
        Raises RuntimeError on every rank or on none."""
        free = _POOL.setdefault(key, [])
        # slabs set aside because a caller still held a table of theirs come back once that table is gone
        for q in [q for q in _QUARANTINE if q.get('key') == key and not q.get('bad') and 'idle_users' in q]:
            try:
                idle = _storage_users(q['slabs']) <= q['idle_users']
            except Exception:  # noqa: BLE001
                idle = False
            if idle:
                _QUARANTINE.remove(q)
                free.append(q)
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
        slabs, mine, failure = [], None, None
        try:
            for nbytes in self._carve.slab_bytes:
                slabs.append(torch.empty(nbytes, dtype=torch.uint8, device=self.device))
            mine = _export_tables(slabs)
        except Exception as exc:  # noqa: BLE001
            failure = RuntimeError(f'IPC export failed: {type(exc).__name__}: {str(exc).splitlines()[0]} [{_fd_state()}; {_ipc_probe()}]')
        try:
            self._agree(failure, 'a rank could not export its tables')
        except RuntimeError:
            if slabs:  # on EVERY rank: memory that was (or may have been) exported is never handed back to the allocator
                _QUARANTINE.append({'slabs': slabs, 'bad': True})
            raise
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        peers, failure = {}, None
        try:
            for r in range(self.world):
                if r == self.rank:
                    continue
                peer_slabs = [rebuild(*args) for rebuild, args in everyone[r]]
                other = peer_slabs[0].device
                if other != self.device:  # another GPU of the node: this GPU must be allowed to address its memory
                    # asked BEFORE anything touches the mapping: a kernel (or copy) that stores through a mapping its GPU
                    # cannot address faults the whole process (hipDeviceCanAccessPeer)
                    if not torch.cuda.can_device_access_peer(self.device.index or 0, other.index or 0):
                        raise RuntimeError(f'{self.device} cannot address the memory of {other} (no peer access)')
                    _enable_peer_access(self.device, other)
                pm, ph, pc = self._carve.views(peer_slabs)
                peers[r] = (peer_slabs, pm + ph + [pc])
        except Exception as exc:  # (mapping a peer's memory can fail on ONE rank only: agree before anybody waits for anybody)
            failure = exc
        try:
            self._agree(failure, 'a rank could not map its peers\' tables')
        except RuntimeError:
            _QUARANTINE.append({'slabs': slabs, 'peers': peers, 'bad': True})  # (the ranks that did map keep their slab out of the allocator too)
            raise
        _POOL_IDS[key] = _POOL_IDS.get(key, 0) + 1  # (creations are collective: the counter agrees across ranks)
        entry = {'id': _POOL_IDS[key], 'slabs': slabs, 'peers': peers, 'key': key}
        try:
            entry['idle_users'] = _storage_users(slabs)  # (the slab with nobody's views on it: what "no table of a build is left" looks like)
        except Exception:  # noqa: BLE001
            pass
        return entry

    def _probe(self, entry, max_hops):
        """store-and-read-back through every mirror: rank r writes a fresh token into row r of every PEER's cards table (column 0), a
        barrier, then every rank checks the rows its peers wrote -- a mapping that opened but does not reach the peer's memory is
        found here, on every rank or on none, not inside a kernel of the first build"""
        self.hop_barrier()  # (a pooled slab: whatever any rank still has queued on the tables of the shard that held it before completes first)
        entry['probes'] = entry.get('probes', 0) + 1
        token = float(1000 * entry['probes'])  # (a reused slab still holds the tokens of its last probe)
        _, _, cards = self._carve.views(entry['slabs'])
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


_POOL = {}        # (group, rows, hops, permutations, registers, device) -> free pool entries (see PeerShard.__init__)
_POOL_GROUPS = {}  # id(process group) -> the group object (pinned: a destroyed group's id must not be reused by another one)
_POOL_IDS = {}    # same key -> entries created so far
_QUARANTINE = []  # slabs / mappings of constructions that failed: never reused, never freed (their memory must not be recycled)


def pool_stats():
    """{'pooled_bytes', 'pooled_slabs', 'set_aside_bytes', 'set_aside_slabs'} of this process: PeerShard slabs are never handed back to
    torch's allocator (recycled allocations are what broke IPC exports in rounds 3-4) -- a dropped shard's slabs wait in the
    pool for the next shard of the same layout, slabs whose tables a caller still holds (or whose construction failed) are set aside
    until those tables are gone.  A process that builds through many different layouts keeps all of them: this is the number to watch."""
    def total(entries):
        slabs = [t for e in entries for t in (e.get('slabs') or ([e['slab']] if 'slab' in e else []))]
        return sum(t.numel() * t.element_size() for t in slabs), len(slabs)
    pooled = total([e for free in _POOL.values() for e in free])
    aside = total(_QUARANTINE)
    return {'pooled_bytes': pooled[0], 'pooled_slabs': pooled[1], 'set_aside_bytes': aside[0], 'set_aside_slabs': aside[1]}


def _storage_users(slabs):
    """tensors / storage handles that share the memory of a slab (or of a list of slabs: the sum) right now, the temporary handle of
    this call included"""
    if isinstance(slabs, torch.Tensor):
        slabs = [slabs]
    return sum(torch._C._storage_Use_Count(t.untyped_storage()._cdata) for t in slabs)


def _release_entry(key, entry, users_of_the_shard_itself):
    """a shard has been dropped: its slab goes back to the pool -- unless tensors handed out by its builds are still alive (the
    caller kept a table but not the shard): that slab is then kept aside for good, a later shard must not overwrite the table.
    users_of_the_shard_itself: _storage_users(slab) when the shard stood complete (the slab + the shard's own views)"""
    try:
        in_use = _storage_users(entry.get('slabs') or [entry['slab']]) > users_of_the_shard_itself
    except Exception:  # noqa: BLE001  (interpreter shutdown, a torch without the hook)
        in_use = True
    (_QUARANTINE if in_use else _POOL.setdefault(key, [])).append(entry)


PEER_SLAB_MAX_BYTES = int(os.environ.get('SS_PEER_SLAB_MAX_MB', '2048')) << 20


class _SlabLayout(object):
    """where the 2 h + 1 tables of a shard lie inside its allocation(s) (every table 256-byte aligned).  ONE slab while the tables fit
    PEER_SLAB_MAX_BYTES (2 GiB; ogbl-ppa size: 0.9 GB), otherwise as few slabs of at most that size as a greedy packing in table order
    gives (a table above the limit gets a slab to itself): opening the IPC handle of a single 6.8 GB slab -- ogbl-citation2 size, h = 3
    -- never returned (two processes on one MI355X, hipIpcOpenMemHandle under torch's rebuild_cuda_tensor, round 6), five slabs of
    <= 2 GiB map at once."""

    def __init__(self, rows, max_hops, num_perm, m, max_slab_bytes=None):
        self.rows, self.h, self.P, self.m = rows, max_hops, num_perm, m
        al = lambda x: (x + 255) & ~255
        self.mh_bytes, self.hll_bytes, self.cards_bytes = al(rows * num_perm * 4), al(rows * m), al(rows * max_hops * 4)
        sizes = [self.mh_bytes] * max_hops + [self.hll_bytes] * max_hops + [self.cards_bytes]
        limit = PEER_SLAB_MAX_BYTES if max_slab_bytes is None else max_slab_bytes
        self.place, self.slab_bytes = [], []  # table -> (slab, offset); bytes of every slab
        for size in sizes:
            if not self.slab_bytes or (self.slab_bytes[-1] + size > limit and self.slab_bytes[-1] > 0):
                self.slab_bytes.append(0)
            self.place.append((len(self.slab_bytes) - 1, self.slab_bytes[-1]))
            self.slab_bytes[-1] += size
        self.bytes = sum(self.slab_bytes)

    def views(self, slabs):
        """slabs: the list of allocations (a single tensor is accepted while the layout has one slab)"""
        if isinstance(slabs, torch.Tensor):
            slabs = [slabs]
        if len(slabs) != len(self.slab_bytes):
            raise ValueError(f'this layout has {len(self.slab_bytes)} slab(s), got {len(slabs)}')
        piece = lambda t, n: slabs[self.place[t][0]][self.place[t][1]:self.place[t][1] + n]
        mh = [piece(k, self.rows * self.P * 4).view(torch.int32).view(self.rows, self.P) for k in range(self.h)]
        hll = [piece(self.h + k, self.rows * self.m).view(self.rows, self.m) for k in range(self.h)]
        cards = piece(2 * self.h, self.rows * self.h * 4).view(torch.float32).view(self.rows, self.h)
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


class ReplicaMismatch(RuntimeError):
    """the sketch tables of the ranks of a group differ after a build that must leave identical replicas"""


def table_digests(tensors):
    """int64 [len(tensors), 2]: ss_table_digest (128-bit content digest, one streaming pass) of every tensor (contiguous, on one device)"""
    from . import _native
    from ._runtime import _ptr, _stream
    lib = _native.lib()
    device = tensors[0].device
    out = torch.empty((len(tensors), 2), dtype=torch.int64, device=device)
    for i, t in enumerate(tensors):
        if not t.is_contiguous():
            raise ValueError('table_digests needs contiguous tensors')
        nbytes = t.numel() * t.element_size()
        if nbytes % 16:  # (cards [N, h] with N * h odd: digest a zero-padded copy -- the same on every rank)
            padded = torch.zeros((nbytes + 15) // 16 * 16, dtype=torch.uint8, device=device)
            padded[:nbytes] = t.reshape(-1).view(torch.uint8)
            t, nbytes = padded, padded.numel()
        _native.check(lib.ss_table_digest(_ptr(t), nbytes, _ptr(out[i]), _stream(device)), 'ss_table_digest')
    return out


def verify_replicas(table, cards, max_hops, num_nodes, group=None, what='build'):
    """every rank digests its hop 1 .. max_hops MinHash / HLL tables and its cards (rows [0, num_nodes)) on the device, the digests
    are exchanged, and EVERY rank raises ReplicaMismatch if any two replicas differ -- or returns the digests.  One streaming read
    of the tables (ogbl-ppa size: 0.9 GB, ~0.15 ms) plus one small host-side exchange; the reference has one table
    (hashing.py:139-165), a multi-GPU build must leave the same one everywhere."""
    tensors = []
    for k in range(1, max_hops + 1):
        tensors += [table[k].mh_u32[:num_nodes], table[k].hll_u8[:num_nodes]]
    tensors.append(cards[:num_nodes].contiguous())
    mine = table_digests(tensors).cpu().tolist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    names = [f'hop {k} {kind}' for k in range(1, max_hops + 1) for kind in ('minhash', 'hll')] + ['cards']
    bad = sorted({names[i] for r in range(1, world) for i in range(len(names)) if everyone[r][i] != everyone[0][i]})
    if bad:
        differing = sorted({r for r in range(1, world) for i in range(len(names)) if everyone[r][i] != everyone[0][i]})
        raise ReplicaMismatch(f'{what}: the replicas of {bad} differ between group rank 0 and rank(s) {differing} (this is rank {rank}): '
                              f'rows written by one GPU were not (all) seen by another')
    return mine


def _peer_verify_mode():
    mode = os.environ.get('SS_PEER_VERIFY', 'first')
    if mode not in ('first', 'always', 'never'):
        raise ValueError(f'SS_PEER_VERIFY must be first, always or never, got {mode!r}')
    return mode


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
    # PeerShard's constructor has probed every mapping with ONE store; what a build relies on is that EVERY row its peers' kernels stored
    # is visible to this rank's next kernel after the hop barrier (coarse-grained device memory written from another GPU, lines of
    # the previous build possibly still in this GPU's L2).  Checked, not assumed: after the FIRST build through a shard (SS_PEER_VERIFY =
    # first, the default; always: every build; never) the replicas' digests are compared; a mismatch raises on every rank -- or, with
    # fallback=True, rebuilds through the exchange form and retires the shard.
    mode = _peer_verify_mode()
    if mode == 'always' or (mode == 'first' and not shard.verified):
        try:
            shard.digests = verify_replicas(table, cards, eh.max_hops, num_nodes, group, 'peer-write build')
            shard.verified = True
        except ReplicaMismatch as exc:
            shard.failed = str(exc)
            logger.error('%s', exc)
            if not fallback:
                raise
            table, cards = sharded_build_hash_tables(eh, num_nodes, edge_index, group)
            return table, cards, None
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
