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
    assert calls == [hi - lo]
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
