"""N > 1 query path on CPU: world_size-2 gloo processes shard the pairs, compute their shard (with the oracle
standing in for the GPU kernel -- the sharding / gather logic is what is under test) and all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, L, out_dir):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    from conftest import oracle_params
    from oracle import oracle
    prm = oracle_params(ssa.hll_tables.load(8, prefer='regenerated'))
    n = 500
    rng = np.random.RandomState(0)
    e = rng.randint(0, n, size=(2, 1500)).astype(np.int64)
    ei = np.concatenate([e, e[::-1]], axis=1)
    tables, cards = oracle.build_hash_tables(n, ei, 2, 128, prm)   # replicated table: identical on every rank
    links = torch.from_numpy(np.random.RandomState(1).randint(0, n, size=(L, 2)).astype(np.int64))
    calls = []

    def compute(shard):
        calls.append(shard.shape[0])
        return torch.from_numpy(oracle.pair_features(shard.numpy(), tables, cards, 2, prm)) if shard.shape[0] else torch.zeros((0, 8))

    full = ssa.dist.sharded_subgraph_features(compute, links)
    lo, hi = ssa.dist.shard_bounds(L, world, rank)
    assert calls == ([hi - lo] if hi > lo else [])  # default: one round = the contiguous shares, one compute call per rank
    ref = torch.from_numpy(oracle.pair_features(links.numpy(), tables, cards, 2, prm)) if L else torch.zeros((0, 8))
    assert full.shape == (L, 8) and torch.equal(full, ref), f'rank {rank}: gathered features differ'
    torch.save(full, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('L', [1001, 64, 1])
def test_sharded_query_world2(tmp_path, L):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), L, str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(a, b)  # every rank ends with the same, complete, ordered result


# ---- the BUDDY precompute across ranks: rounds of blocks gathered in place, three gather modes -------------------
def test_link_rounds_partition():
    """LinkRounds: every link is owned by exactly one rank; a round's rows are one contiguous piece with rank r's block as its r-th
    equal part (the layout all_gather_into_tensor writes); the padding stays below one block per rank"""
    import subgraph_sketching_amd as ssa
    for L in (0, 1, 7, 1001, 65536):
        for world in (1, 2, 3, 8):
            for block in (1, 5, 300, 10 ** 6):
                if L // (world * block) > 5000:
                    continue
                plans = [ssa.dist.LinkRounds(L, world, r, block) for r in range(world)]
                owner = np.full(L, -1)
                for r, plan in enumerate(plans):
                    assert plan.rounds == plans[0].rounds and plan.padded_links == plans[0].padded_links
                    for c, (base, per) in enumerate(plan.rounds):
                        lo, hi = plan.owned(c)
                        assert 0 <= hi - lo <= per and (lo == hi or lo == base + r * per)
                        assert (owner[lo:hi] == -1).all()
                        owner[lo:hi] = r
                    idx = plan.owned_index()
                    assert idx.numel() == plan.owned_count() and (owner[idx.numpy()] == r).all()
                assert (owner >= 0).all() and L <= plans[0].padded_links < L + world * min(block, max(L, 1)) + 1
                for (base, per), nxt in zip(plans[0].rounds, plans[0].rounds[1:] + [(plans[0].padded_links, 0)]):
                    assert base + world * per == nxt[0]  # rounds tile the (padded) output
    assert ssa.dist.default_link_block(356_000_000, 8) == 11_125_000 and ssa.dist.default_link_block(65536, 8) == 1 << 21
    with pytest.raises(ValueError):
        ssa.dist.LinkRounds(10, 2, 0, 0)


def _rounds_worker(rank, world, port, L, block, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    F = 15
    links = torch.from_numpy(np.random.RandomState(4).randint(0, 1000, size=(L, 2)).astype(np.int64))
    rows_of = lambda lk: (lk[:, :1] * 1000 + lk[:, 1:]).float() + torch.arange(F, dtype=torch.float32)[None, :]
    ref = rows_of(links)
    for gather in ssa.dist.GATHER_MODES:
        calls = []

        def fill(link_slice, out_view):
            calls.append(link_slice.size(0))
            assert out_view.shape == (link_slice.size(0), F) and out_view.is_contiguous()
            out_view.copy_(rows_of(link_slice))
        res = ssa.dist._sharded_rows(fill, links, F, torch.device('cpu'), None, gather, block)
        if gather == 'none':
            lo, hi = ssa.dist.shard_bounds(L, world, rank)
            assert res.rows is None and calls == [hi - lo] and torch.equal(res.index, torch.arange(lo, hi))
        else:
            plan = ssa.dist.LinkRounds(L, world, rank, block)
            assert calls == [hi - lo for lo, hi in (plan.owned(c) for c in range(len(plan.rounds))) if hi > lo]
            if gather == 'all' or rank == 0:
                assert res.rows.shape == (L, F) and torch.equal(res.rows, ref), f'rank {rank} {gather}: gathered rows differ'
            else:
                assert res.rows is None
        assert torch.equal(res.local, ref[res.index]), f'rank {rank} {gather}: local rows differ'
        # the callable form returns the tensor of the mode
        got = ssa.dist.sharded_subgraph_features(rows_of, links, gather=gather, block=block)
        if gather == 'none':
            lo, hi = ssa.dist.shard_bounds(L, world, rank)
            assert torch.equal(got, ref[lo:hi])
        elif gather == 'all' or rank == 0:
            assert torch.equal(got, ref)
        else:
            assert got is None
    owned = [None] * world
    dist.all_gather_object(owned, res.index.tolist())
    assert sorted(i for part in owned for i in part) == list(range(L))  # the shares of the ranks partition the link set
    dist.destroy_process_group()


@pytest.mark.parametrize('world,L,block', [(2, 1001, 100), (3, 1000, 64), (2, 5, 100), (3, 2, 1), (2, 600, 300)])
def test_precompute_rounds_and_gather_modes(tmp_path, world, L, block):
    """sharded_precompute's machinery on gloo: several rounds + a ragged tail, gather = all / rank0 / none"""
    mp.spawn(_rounds_worker, args=(world, _free_port(), L, block, str(tmp_path)), nprocs=world, join=True)


# ---- sharded build: destination rows partitioned, in-place all-gather after every hop ------------------
def _build_worker(rank, world, port, n, out_dir, exchange='all_gather'):
    import sys
    os.environ['SS_EXCHANGE'] = exchange  # all_gather_into_tensor, or G - 1 concurrent point-to-point transfers per rank
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    from conftest import oracle_params
    from oracle import oracle
    prm = oracle_params(ssa.hll_tables.load(8, prefer='regenerated'))
    rng = np.random.RandomState(3)
    e = rng.randint(0, n, size=(2, 3 * n)).astype(np.int64)
    ei = np.concatenate([e, e[::-1]], axis=1)
    ei_loops = oracle.add_self_loops(ei)
    shard = ssa.dist.RowShard(n, None)
    assert shard.p2p == (exchange == 'p2p')
    lo, hi = shard.rows
    assert shard.padded_rows % world == 0 and shard.padded_rows >= n and 0 <= lo <= hi <= n
    if world == 2 and n > 2:
        assert (lo, hi) == ((0, (n + 1) // 2) if rank == 0 else ((n + 1) // 2, n))
    # the protocol of ElphHashes._build with the oracle standing in for the row-range kernels: each rank fills ONLY the
    # rows it owns (everything else poisoned), gathers, and must then hold the complete hop to compute the next one
    mh = torch.from_numpy(oracle.minhash_init(n, 128).view(np.int32))  # packed uint32 carried as int32, like the engine
    hll = torch.from_numpy(oracle.hll_init(n, 8))
    cards = torch.full((shard.padded_rows, 2), float('nan'))
    for k in (1, 2):
        full_mh, full_hll = oracle.propagate(n, ei_loops, mh=mh.numpy().view(np.uint32), hll=hll.numpy())
        full_mh = full_mh.view(np.int32)
        mine_mh = torch.full((shard.padded_rows, 128), -7, dtype=torch.int32)
        mine_hll = torch.full((shard.padded_rows, 256), 77, dtype=torch.uint8)
        mine_mh[lo:hi] = torch.from_numpy(full_mh)[lo:hi]
        mine_hll[lo:hi] = torch.from_numpy(full_hll)[lo:hi]
        cards[lo:hi, k - 1] = torch.from_numpy(oracle.hll_count(full_hll, prm))[lo:hi]
        shard.wait(shard.gather(mine_mh))
        shard.wait(shard.gather(mine_hll))
        mh, hll = mine_mh[:n].clone(), mine_hll[:n].clone()
        assert torch.equal(mh, torch.from_numpy(full_mh)), f'rank {rank} hop {k}: MinHash rows differ after the gather'
        assert torch.equal(hll, torch.from_numpy(full_hll)), f'rank {rank} hop {k}: HLL rows differ after the gather'
    shard.wait(shard.gather(cards))
    ref_tables, ref_cards = oracle.build_hash_tables(n, ei, 2, 128, prm)
    assert np.array_equal(mh.numpy().view(np.uint32), ref_tables[2]['minhash'])
    assert np.array_equal(hll.numpy(), ref_tables[2]['hll'])
    assert np.array_equal(cards[:n].numpy(), ref_cards)
    torch.save((mh, hll, cards[:n]), os.path.join(out_dir, f'build_rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('exchange', ['all_gather', 'p2p'])
@pytest.mark.parametrize('n', [501, 64, 1])
def test_sharded_build_protocol_world2(tmp_path, n, exchange):
    world = 2
    mp.spawn(_build_worker, args=(world, _free_port(), n, str(tmp_path), exchange), nprocs=world, join=True)
    a, b = torch.load(tmp_path / 'build_rank0.pt'), torch.load(tmp_path / 'build_rank1.pt')
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_row_shard_bounds():
    """ownership arithmetic without a process group (RowShard reads world / rank from one)"""
    import subgraph_sketching_amd as ssa
    for n in (0, 1, 7, 8, 9, 235868):
        for world in (1, 2, 3, 8):
            per = max((n + world - 1) // world, 1)
            covered = []
            for rank in range(world):
                lo = min(rank * per, n)
                covered.append((lo, min(lo + per, n)))
            assert covered[0][0] == 0 and covered[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            assert per * world >= n
    assert ssa.dist.RowShard.__init__.__code__.co_argcount == 4  # (self, num_nodes, group, exchange)


# ---- bench.py's workload bookkeeping: BatchPlan + AsyncFeatureGather under both scaling modes -------------------------
def _plan_worker(rank, world, port, batch, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    F = 8
    results = {}
    for scaling in ('strong', 'weak'):
        plan = ssa.dist.BatchPlan(scaling, world, rank, batch)
        links = torch.from_numpy(np.random.RandomState(plan.links_seed).randint(0, 1000, size=(batch, 2)).astype(np.int64))
        mine = plan.local(links)
        assert mine.size(0) == plan.local_pairs
        gather = ssa.dist.AsyncFeatureGather(plan, F, torch.device('cpu'))
        outs = []
        for step in range(5):  # more gathers than buffers: slots are reused, the ragged shard is re-padded every time
            feats = (mine[:, :1] * 10 + mine[:, 1:] + step).float().repeat(1, F)
            outs.append((step, gather(feats)))
        gather.drain()
        step, buf = outs[-1]
        got = plan.unpad(buf)
        if scaling == 'strong':  # every rank generated the same global batch: the gathered rows are the whole batch in order
            want = (links[:, :1] * 10 + links[:, 1:] + step).float().repeat(1, F)
            assert got.shape == (batch, F) and torch.equal(got, want), f'rank {rank} strong: gathered rows differ'
        else:                    # rank r's block holds rank r's own pairs
            assert got.shape == (world * batch, F)
            for r in range(world):
                lr = torch.from_numpy(np.random.RandomState(2 + r).randint(0, 1000, size=(batch, 2)).astype(np.int64))
                assert torch.equal(got[r * batch:(r + 1) * batch], (lr[:, :1] * 10 + lr[:, 1:] + step).float().repeat(1, F))
        results[scaling] = got
    torch.save(results, os.path.join(out_dir, f'plan_rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('batch', [1001, 64, 3])
def test_batch_plan_and_async_gather_world2(tmp_path, batch):
    world = 2
    mp.spawn(_plan_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / 'plan_rank0.pt'), torch.load(tmp_path / 'plan_rank1.pt')
    assert torch.equal(a['strong'], b['strong']) and torch.equal(a['weak'], b['weak'])


def _p2p_worker(rank, world, port, per):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    full = torch.full((world * per, 3), -1, dtype=torch.int32)
    full[rank * per:(rank + 1) * per] = rank + 10
    ssa.dist.exchange_blocks_p2p(full, rank, world, per)
    want = torch.arange(world, dtype=torch.int32).repeat_interleave(per)[:, None].expand(-1, 3) + 10
    assert torch.equal(full, want), f'rank {rank}: blocks after the point-to-point exchange differ'
    dist.destroy_process_group()


@pytest.mark.parametrize('world,per', [(3, 7), (4, 1), (2, 100)])
def test_point_to_point_block_exchange(world, per):
    """the direct (link-parallel) form of the per-hop exchange: every rank ends with every block, any world size"""
    mp.spawn(_p2p_worker, args=(world, _free_port(), per), nprocs=world, join=True)


def _p2p_subgroup_worker(rank, world, port, per):
    """ranks {1, 3} and {0, 2} form two sub-groups (group rank != global rank for rank 3 / 2): the exchange must stay inside each"""
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    groups = [dist.new_group([0, 2]), dist.new_group([1, 3])]  # every rank creates every group (torch requirement)
    group = groups[rank % 2]
    g_rank, g_world = dist.get_rank(group), dist.get_world_size(group)
    assert g_world == 2 and g_rank == rank // 2
    full = torch.full((g_world * per, 2), -1, dtype=torch.int32)
    full[g_rank * per:(g_rank + 1) * per] = 100 * (rank % 2) + g_rank
    ssa.dist.exchange_blocks_p2p(full, g_rank, g_world, per, group)
    want = torch.arange(g_world, dtype=torch.int32).repeat_interleave(per)[:, None].expand(-1, 2) + 100 * (rank % 2)
    assert torch.equal(full, want), f'rank {rank}: sub-group exchange reached the wrong peers'
    # RowShard on the sub-group: owned rows and the gathered table follow the GROUP rank
    shard = ssa.dist.RowShard(10, group)
    assert (shard.rank, shard.world) == (g_rank, g_world)
    dist.barrier()
    dist.destroy_process_group()


def test_point_to_point_block_exchange_in_a_sub_group():
    """P2POp peers are global ranks; a per-node sub-group must exchange with its own members"""
    mp.spawn(_p2p_subgroup_worker, args=(4, _free_port(), 5), nprocs=4, join=True)


def _probe_worker(rank, world, port):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.pop('SS_EXCHANGE', None)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import subgraph_sketching_amd as ssa
    choice = ssa.dist.choose_exchange(torch.device('cpu'), None, block_bytes=1 << 16, reps=2)
    times = ssa.dist.exchange_probe_times(torch.device('cpu'))
    assert choice in ('all_gather', 'p2p') and set(times) == {'all_gather', 'p2p'} and all(t > 0 for t in times.values())
    # every rank reaches the same decision (it is taken on the max over ranks) and the decision is cached
    picks = [None] * world
    dist.all_gather_object(picks, choice)
    assert len(set(picks)) == 1 and ssa.dist.choose_exchange(torch.device('cpu')) == choice
    shard = ssa.dist.RowShard(100, None, choice)
    assert shard.p2p == (choice == 'p2p')
    os.environ['SS_EXCHANGE'] = 'p2p'   # the environment variable is an override for tests / A-B runs
    assert ssa.dist.choose_exchange(torch.device('cpu')) == 'p2p'
    dist.destroy_process_group()


def test_exchange_form_is_chosen_by_a_micro_probe():
    """the form of the per-hop block exchange (collective or world - 1 point-to-point transfers) is measured at
    start-up, the same on every rank; SS_EXCHANGE only overrides"""
    mp.spawn(_probe_worker, args=(3, _free_port()), nprocs=3, join=True)
