#!/usr/bin/env python3
"""probe (needs >= 2 GPUs of one node; launch with torchrun): how the per-hop exchange of the row-sharded build moves over xGMI.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/probe_allgather.py

For the block sizes of the two large configs (ogbl-ppa: N*768/8 = 55 MB per rank and hop, ogbl-citation2: 281 MB) it times
  (a) dist.all_gather_into_tensor, in place (what dist.RowShard.gather issues), and
  (b) the same exchange as G-1 concurrent point-to-point transfers per rank (batch_isend_irecv: every rank sends its block
      to every peer at once -- one transfer per xGMI link, the "direct" exchange SURVEY section 5 asks for),
and prints the effective per-rank receive bandwidth.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): if (a) is a ring it
is bound by ONE link (~(G-1)/G * bytes / 153 GB/s), if it spreads over all links it approaches (b).  RowShard.gather switches to
the point-to-point form when SS_EXCHANGE=p2p is set, so whichever wins here can be selected without touching the engine.
SURVEY section 8(e): the row-sharded build only pays off if this exchange stays well below the per-hop compute it saves
(ppa: 2.2 ms, citation2: 5.6 ms per hop and rank at 8 ranks)."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    import subgraph_sketching_amd as ssa
    rows = []
    for label, block_bytes in (('collab-like hop (N*768/G at G=8)', 235868 * 768 // 8), ('ppa-like hop', 576289 * 768 // 8),
                               ('citation2-like hop', 2927963 * 768 // 8)):
        per = block_bytes // 4
        full = torch.zeros(world * per, dtype=torch.int32, device=dev)
        mine = full[rank * per:(rank + 1) * per]
        mine.fill_(rank + 1)
        res = {'block': label, 'bytes_per_rank': per * 4, 'world': world}
        for mode in ('all_gather', 'p2p'):
            def once():
                if mode == 'all_gather':
                    dist.all_gather_into_tensor(full, mine)
                else:
                    ssa.dist.exchange_blocks_p2p(full, rank, world, per)
            for _ in range(3):
                once()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                once()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res[mode + '_ms'] = float(t.item()) * 1e3
            res[mode + '_recv_GBps_per_rank'] = (world - 1) * per * 4 / float(t.item()) / 1e9
            ok = all(int(full[r * per]) == r + 1 and int(full[(r + 1) * per - 1]) == r + 1 for r in range(world))
            assert ok, f'{mode}: blocks did not arrive'
        rows.append(res)
        if rank == 0:
            print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
