"""Multi-GPU query path: edge-pair batches shard across ranks, sketch tables replicated.

One process per GPU (torchrun); `torch.distributed` backend "nccl" is RCCL on ROCm.  Pairs are
independent (the reference maps over rows, hashing.py:272-321) so the only exchange is the gather of
the per-rank feature rows -- one all_gather_into_tensor of [ceil(L/G), h(h+2)] fp32 per call
(2 MiB per rank at B=65536, h=2: latency-bound on xGMI, no ring-vs-direct concern).
min/max and integer counts are order independent, so every rank's replicated table is bit-identical
and no reduction collective exists on the path.
"""
import torch
import torch.distributed as dist


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
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)
    return gathered[:L]
