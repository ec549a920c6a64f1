"""Stress test of the cross-workgroup hand-off of mega rows (rows with more than SS_MEGA_SLICE in-neighbours).

Round 1 published a row's scratch slots behind a WORKGROUP-scope fence, which emits no `s_waitcnt` on gfx950: the ticket
could reach the memory side before the slot stores and the last finisher combine a stale slot.  The
fix drains the storing waves (`s_waitcnt vmcnt(0)`) before the barrier that precedes the ticket
(csrc/ss_common.hpp publish_drain; ISA excerpt in profiles/round2_handoff_isa.txt).  This test tries to hit the window:
64 mega rows of >= 100 000 in-neighbours (about 1 600 slices dealt over the 256 hub workgroups), both hub kernels
(table hop and first hop from node ids), 500 repetitions each while a second stream saturates the memory system with
large copies -- the case in which the stores of a slot queue behind other traffic.  Every repetition compares every mega
row (MinHash + HLL + cardinality) with the UNSLICED path (the same rows handled as ordinary hub rows by one workgroup
each, no scratch, no ticket); the unsliced result itself is checked once against the CPU oracle.
"""
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import oracle_params

pytestmark = pytest.mark.gpu

N_NODES, N_MEGA, MEGA_DEG, REPS = 150_000, 64, 100_000, 500


def _graph(seed=71):
    rng = np.random.RandomState(seed)
    src = [rng.randint(0, N_NODES, size=300_000)]
    dst = [rng.randint(0, N_NODES, size=300_000)]
    mega = rng.choice(N_NODES, size=N_MEGA, replace=False)
    for i, node in enumerate(mega):
        d = MEGA_DEG + 37 * i  # different slice counts and ragged last slices
        src.append(rng.choice(N_NODES, size=d, replace=False))
        dst.append(np.full(d, node))
    ei = np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
    return ei, np.sort(mega)


class _Hog(object):
    """keeps a side stream busy with 256 MiB device copies (read + write traffic through every channel)"""

    def __init__(self, dev):
        self.stream = torch.cuda.Stream(device=dev)
        self.a = torch.empty(64 * 1024 * 1024, dtype=torch.int32, device=dev).random_()
        self.b = torch.empty_like(self.a)

    def kick(self, copies=2):
        with torch.cuda.stream(self.stream):
            for _ in range(copies):
                self.b.copy_(self.a, non_blocking=True)


def test_mega_row_handoff_under_memory_pressure(regenerated_tables):
    import subgraph_sketching_amd as ssa
    from oracle import oracle
    H = ssa.hashing
    dev = torch.device('cuda:0')
    ei_np, mega_nodes = _graph()
    ei = torch.from_numpy(ei_np).to(dev)
    eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
    eh.hll_tables = regenerated_tables[8]
    params = eh._params(dev)
    csr = ssa.build_csr(ei, N_NODES, dev)
    csr.use_inferred_self_loops = True
    assert int(csr.mega_count[0].item()) == N_MEGA
    n_slices = int(csr.mega_count[1].item())
    assert n_slices >= N_MEGA * (MEGA_DEG // ssa._native.MEGA_SLICE)
    rows = torch.from_numpy(mega_nodes).to(dev)

    def run(graph_csr, mh_in, hll_in, mh_out, hll_out, cards):
        if mh_in is None:  # hop 1 from node ids: first_hop_hub_kernel
            eh._first_hop(graph_csr, dev, mh_out, hll_out, cards, params)
        else:              # table hop: propagate_hub_kernel
            H._propagate(graph_csr, mh_in, hll_in, dev, cards_out=cards[:, 1], cards_stride=2, params=params, mh_out=mh_out,
                         hll_out=hll_out)

    # ---- reference results by the unsliced path: the same CSR with the mega list withheld -> every long row is an ordinary
    # hub row (one workgroup walks all of it; nothing crosses workgroups)
    unsliced = H.CsrGraph(csr.rowptr, csr.col, N_NODES, csr.n_self_dev, csr.err, torch.cat([csr.hub_rows[:int(csr.hub_count.item())],
                          csr.mega_rows[:N_MEGA, 0].contiguous()]).contiguous(), torch.tensor([int(csr.hub_count.item()) + N_MEGA],
                          dtype=torch.int32, device=dev), csr.hub_threshold, mega=None)
    unsliced.use_inferred_self_loops = True
    mh1 = torch.empty((N_NODES, 128), dtype=torch.int32, device=dev)
    hl1 = torch.empty((N_NODES, 256), dtype=torch.uint8, device=dev)
    mh2, hl2 = torch.empty_like(mh1), torch.empty_like(hl1)
    cards_ref = torch.zeros((N_NODES, 2), dtype=torch.float32, device=dev)
    run(unsliced, None, None, mh1, hl1, cards_ref)
    run(unsliced, mh1, hl1, mh2, hl2, cards_ref)
    prm = oracle_params(regenerated_tables[8])
    otab, ocards = oracle.build_hash_tables(N_NODES, ei_np, 2, 128, prm)
    for k, (m, l) in enumerate([(mh1, hl1), (mh2, hl2)], start=1):
        assert np.array_equal(m.cpu().numpy().view(np.uint32), otab[k]['minhash']), f'unsliced minhash hop {k}'
        assert np.array_equal(l.cpu().numpy(), otab[k]['hll']), f'unsliced hll hop {k}'
    np.testing.assert_allclose(cards_ref.cpu().numpy(), ocards, rtol=1e-5, atol=1e-4)
    want = {1: (mh1[rows].clone(), hl1[rows].clone(), cards_ref[rows, 0].clone()),
            2: (mh2[rows].clone(), hl2[rows].clone(), cards_ref[rows, 1].clone())}

    # ---- the sliced path, REPS times per kernel, under memory pressure; mismatches are counted on the device
    hog = _Hog(dev)
    bad = torch.zeros(6, dtype=torch.int64, device=dev)
    out_mh, out_hl = torch.empty_like(mh1), torch.empty_like(hl1)
    cards = torch.zeros((N_NODES, 2), dtype=torch.float32, device=dev)
    for rep in range(REPS):
        hog.kick()
        out_mh.fill_(-1)  # a row nobody wrote must not look right
        out_hl.fill_(255)
        run(csr, None, None, out_mh, out_hl, cards)
        bad[0] += (out_mh[rows] != want[1][0]).any(dim=1).sum()
        bad[1] += (out_hl[rows] != want[1][1]).any(dim=1).sum()
        bad[2] += (cards[rows, 0] != want[1][2]).sum()
        hog.kick()
        out_mh.fill_(-1)
        out_hl.fill_(255)
        run(csr, mh1, hl1, out_mh, out_hl, cards)
        bad[3] += (out_mh[rows] != want[2][0]).any(dim=1).sum()
        bad[4] += (out_hl[rows] != want[2][1]).any(dim=1).sum()
        bad[5] += (cards[rows, 1] != want[2][2]).sum()
    torch.cuda.synchronize(dev)
    counts = bad.cpu().tolist()
    assert counts == [0] * 6, (f'stale mega rows over {REPS} repetitions x {N_MEGA} rows: first hop minhash/hll/cards = {counts[:3]}, '
                               f'table hop minhash/hll/cards = {counts[3:]}')
    assert not csr.mega_rows[:N_MEGA, 3].any()  # every ticket counter is back to zero
    # the whole table of the last repetition, not only the mega rows
    assert torch.equal(out_mh, mh2) and torch.equal(out_hl, hl2)


def test_dense_bucket_helpers_under_memory_pressure():
    """the CSR build's helper workgroups (ss_csr.hip dense_helper): bucket workgroups REGISTER dense buckets and publish their
    descriptors + zeroed counters (release -> arrival counter), helpers wait for every bucket, count their shares into global per-node
    counters, meet at a counter barrier and place the sources.  300 builds of a graph whose first buckets hold most of the edges,
    while a second stream saturates the memory system; every build's CSR is compared on the device with the first one's row
    multisets (which is checked against a host counting sort); both plans: the two-launch gather plan and the partition plan."""
    import os
    import subgraph_sketching_amd as ssa
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(3)
    for n, e_und, force_partition in ((120_000, 900_000, False), (400_000, 3_000_000, True)):
        w = np.arange(1, n + 1, dtype=np.float64) ** -0.9
        cdf = np.cumsum(w / w.sum())
        e = np.stack([np.minimum(np.searchsorted(cdf, rng.random_sample(e_und)), n - 1), rng.randint(0, n, size=e_und)]).astype(np.int64)
        ei_np = np.concatenate([e, e[::-1]], axis=1)
        ei = torch.from_numpy(ei_np).to(dev)
        E = ei.size(1)
        deg = torch.bincount(ei[1], minlength=n)
        want_rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(deg, 0)])
        want_keys = torch.sort(ei[1] * n + ei[0])[0]                      # (row, source) pairs, sorted: the row multisets
        rows_of = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        hog = _Hog(dev)
        bad = torch.zeros(2, dtype=torch.int64, device=dev)
        n_dense = None
        for rep in range(300 if not force_partition else 60):
            hog.kick()
            csr = ssa.build_csr(ei, n, dev, check=False)
            bad[0] += (csr.rowptr != want_rowptr).sum()
            got = torch.sort(rows_of * n + csr.col[:E].to(torch.int64))[0]
            bad[1] += (got != want_keys).sum()
        torch.cuda.synchronize(dev)
        assert bad.cpu().tolist() == [0, 0], f'n={n}: rowptr / col mismatches over the repetitions: {bad.cpu().tolist()}'
        assert int(deg.max()) > 4 * ssa._native.MEGA_SLICE
        # the graph really takes the dense path: some 1024-node (or smaller) bucket holds far more than 32 768 edges
        assert int(deg[:64].sum()) > 32768
