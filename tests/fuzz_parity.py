"""fuzz: random graph shapes, hop counts, hub thresholds and batch sizes through BOTH call styles -- build_hash_tables +
get_subgraph_features (BUDDY) and the ELPH call sequence with its deferred launches (fused stage, row-list table hop) --
against the C oracle: tables bit-exact, features within the stated tolerance, the two call styles bit-identical to each other.
usage (GPU box): python tests/fuzz_parity.py [seconds = 120]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

import numpy as np
import torch

import subgraph_sketching_amd as ssa
from oracle import oracle

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device('cuda:0')
H = ssa.hashing
K = ssa.knobs  # (the switches live there; hashing.<NAME> only reads them)
t8 = ssa.hll_tables.load(8, prefer='regenerated')
prm8 = oracle.HllParams(t8.p, t8.threshold, t8.raw_estimate, t8.bias, alpha=t8.alpha, lc_table=H.linear_counting_table(256).numpy())
rng = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '12345')))
t0, trials, bad = time.time(), 0, 0
while time.time() - t0 < seconds:
    n = int(rng.choice([60, 900, 7000, 40000]))
    e_und = int(n * rng.choice([0.5, 3, 10, 40]))
    h = int(rng.choice([1, 2, 3]))
    if rng.randint(3) == 0:  # skewed endpoints: hub rows, with a low threshold also mega-ish rows
        w = np.arange(1, n + 1, dtype=np.float64) ** -float(rng.choice([0.5, 0.9, 1.2]))
        cdf = np.cumsum(w / w.sum())
        e = np.stack([np.minimum(np.searchsorted(cdf, rng.random_sample(e_und)), n - 1), rng.randint(0, n, size=e_und)])
    else:
        e = rng.randint(0, n, size=(2, e_und))
    if rng.randint(4) == 0:
        e = e[:, e.max(axis=0) < n - 5] if e.shape[1] else e  # trailing nodes without a self loop
    ei = np.concatenate([e, e[::-1]], axis=1).astype(np.int64)
    K.HUB_THRESHOLD = int(rng.choice([0, 0, 8, 40, 300])) or None
    B = int(rng.choice([1, 17, 400, 5000]))
    links = rng.randint(-n, n, size=(B, 2)).astype(np.int64)
    style = int(rng.randint(4))  # link lists with locality: the grouped / run-aware query paths (round 3)
    if style == 1:
        links = links[np.argsort(links[:, 0], kind='stable')]
    elif style == 2:  # evaluation style: every source lists its pairs together
        links[:, 0] = np.repeat(rng.randint(-n, n, size=B // 7 + 1), 7)[:B]
    K.GROUP_LINKS_MIN = int(rng.choice([1, 1, 1 << 20]))          # 1: every query of this trial takes the grouped path
    K.GROUP_GATHER_MIN = int(rng.choice([1, 1 << 24]))            # 1: ... through the gather / scatter passes
    # a quarter of the trials: other sketch sizes (the run-time-sized kernels, the other specialised permutation counts)
    P, hp = (int(rng.choice([4, 20, 64, 100, 192, 256, 260])), int(rng.choice([4, 5, 6, 8, 10, 12]))) if rng.randint(4) == 0 else (128, 8)
    tp = t8 if hp == 8 else ssa.hll_tables.load(hp, prefer='regenerated')
    prm = prm8 if hp == 8 else oracle.HllParams(tp.p, tp.threshold, tp.raw_estimate, tp.bias, alpha=tp.alpha,
                                                 lc_table=H.linear_counting_table(1 << hp).numpy())
    eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=hp, minhash_num_perm=P, floor_sf=bool(rng.randint(2)), use_zero_one=bool(rng.randint(2))))
    eh.hll_tables = tp
    eh.group_links = [True, False, 'auto'][rng.randint(3)]
    tag = f'P={P} p={hp} n={n} e_dir={ei.shape[1]} h={h} B={B} hub={H.HUB_THRESHOLD} links={style} group={eh.group_links}/{H.GROUP_LINKS_MIN}/{H.GROUP_GATHER_MIN}'
    otab, ocards = oracle.build_hash_tables(n, ei, h, P, prm)
    lk_pos = np.where(links < 0, links + n, links)
    ofeat = oracle.pair_features(lk_pos, otab, ocards, h, prm, use_zero_one=eh.use_zero_one, floor_sf=eh.floor_sf)
    ok = True
    # BUDDY style
    tei, tlk = torch.from_numpy(ei).to(dev), torch.from_numpy(links).to(dev)
    table, cards = eh.build_hash_tables(n, tei)
    for k in range(1, h + 1):
        ok &= np.array_equal(table[k].mh_u32.cpu().numpy().view(np.uint32), otab[k]['minhash'])
        ok &= np.array_equal(table[k].hll_u8.cpu().numpy(), otab[k]['hll'])
    f1 = eh.get_subgraph_features(tlk, table, cards, batch_size=int(rng.choice([11000000, 1000, 64])))
    scale = max(1.0, float(np.abs(ofeat).max()) / 100) if ofeat.size else 1.0
    ok &= bool(np.allclose(f1.cpu().numpy(), ofeat, rtol=1e-4, atol=1e-4 * scale))
    # ELPH style (explicit self loops over max(edge_index) + 1 nodes, per-hop modules, deferred launches)
    if ei.shape[1]:
        n_loops = int(ei.max()) + 1
        hei = torch.cat([tei, torch.arange(n_loops, device=dev).repeat(2, 1)], dim=1)
        tb = {0: {'minhash': eh.initialise_minhash(n), 'hll': eh.initialise_hll(n)}}
        cd = torch.zeros((n, h), device=dev)
        for k in range(1, h + 1):
            tb[k] = {'hll': eh.hll_prop(tb[k - 1]['hll'], hei), 'minhash': eh.minhash_prop(tb[k - 1]['minhash'], hei)}
            cd[:, k - 1] = eh.hll_count(tb[k]['hll'])
        f2 = eh.get_subgraph_features(tlk, tb, cd, batch_size=int(rng.choice([11000000, 7, 333])))
        ok &= torch.equal(f2, f1)
        for k in range(1, h + 1):  # whoever asks afterwards gets the whole tables
            ok &= np.array_equal(H._packed_minhash_of(tb[k]['minhash'], dev).cpu().numpy().view(np.uint32), otab[k]['minhash'])
    # the SAME engine on another graph of the same shape: hub hints and the content-keyed CSR must not carry anything over
    if ei.shape[1] and rng.randint(3) == 0:
        top = int(ei.max())
        e2 = rng.randint(0, top + 1, size=e.shape)
        if rng.randint(2):  # this one skewed whatever the first was (a stale "no hub rows" hint)
            e2[0] = np.minimum(((top + 1) * rng.random_sample(e.shape[1]) ** 3).astype(np.int64), top)
        e2[:, 0] = top  # (same max id: the ELPH-style edge_index keeps its shape)
        ei2 = np.concatenate([e2, e2[::-1]], axis=1).astype(np.int64)
        otab2, ocards2 = oracle.build_hash_tables(n, ei2, h, P, prm)
        tei2 = torch.from_numpy(ei2).to(dev)
        for rep in range(2):  # (the second build runs on the first one's hint)
            table2, cards2 = eh.build_hash_tables(n, tei2)
            for k in range(1, h + 1):
                ok &= np.array_equal(table2[k].mh_u32.cpu().numpy().view(np.uint32), otab2[k]['minhash'])
                ok &= np.array_equal(table2[k].hll_u8.cpu().numpy(), otab2[k]['hll'])
            ok &= bool(np.allclose(cards2.cpu().numpy(), ocards2, rtol=1e-4, atol=1e-4))
        # ELPH style: the tensor of the first graph edited in place (same allocation, same shape, new contents)
        hei[:, :tei2.size(1)] = tei2
        tb = {0: {'minhash': eh.initialise_minhash(n), 'hll': eh.initialise_hll(n)}}
        for k in range(1, h + 1):
            tb[k] = {'hll': eh.hll_prop(tb[k - 1]['hll'], hei), 'minhash': eh.minhash_prop(tb[k - 1]['minhash'], hei)}
            ok &= np.array_equal(tb[k]['hll'].cpu().numpy().view(np.uint8), otab2[k]['hll'])
            ok &= np.array_equal(H._packed_minhash_of(tb[k]['minhash'], dev).cpu().numpy().view(np.uint32), otab2[k]['minhash'])
    eh.check_errors()
    trials += 1
    if not ok:
        bad += 1
        print('MISMATCH', tag, flush=True)
print(f'{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s')
sys.exit(1 if bad else 0)
