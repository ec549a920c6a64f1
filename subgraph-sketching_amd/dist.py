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


class RowShard(object):
    """destination-row partition of one build across the ranks of `group` + the exchange that follows every hop.

    Rank r owns rows [r * per, min((r + 1) * per, num_nodes)); tables are allocated with `padded_rows` = per * world
    rows so that every rank contributes an equally sized block and `all_gather_into_tensor` can run IN PLACE (input =
    the owned block of the output).  xGMI is point-to-point: RCCL's all-gather of N*R/G bytes per rank uses all 7 links
    of a GPU at once; no ring over a single link is forced by this call pattern.  With the nccl (RCCL) backend the
    gather is asynchronous (its own stream): `ElphHashes._build` overlaps the MinHash exchange with the HLL kernel and
    vice versa.  Other backends (gloo in the tests) stage device tensors through the host."""

    def __init__(self, num_nodes, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.per = max((num_nodes + self.world - 1) // self.world, 1)
        self.padded_rows = self.per * self.world
        lo = min(self.rank * self.per, num_nodes)
        self.rows = (lo, min(lo + self.per, num_nodes))
        self.native = dist.get_backend(group) == 'nccl'

    def block(self, full):
        return full[self.rank * self.per:(self.rank + 1) * self.per]

    def gather(self, full):
        """every rank's owned block of `full` ([padded_rows, ...], contiguous) reaches every other rank"""
        assert full.size(0) == self.padded_rows and full.is_contiguous()
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


def sharded_build_hash_tables(eh, num_nodes, edge_index, group=None):
    """`eh.build_hash_tables(num_nodes, edge_index)` with the rows of every hop computed once across the group instead
    of once per rank; every rank passes the SAME edge_index and returns the SAME full (table, cards) -- bit-identical to
    the unsharded result (min / max / integer counts are order independent, the cardinality of a row is computed by
    the rank that owns it with the same arithmetic)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return eh.build_hash_tables(num_nodes, edge_index)
    return eh._build(num_nodes, edge_index, RowShard(num_nodes, group))
