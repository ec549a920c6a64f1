"""stress: GPU engine vs the C oracle on many seeded collab-sized graphs (rare paths of the two-phase first hop: keys that
share a 64-wide bucket, low words within 8 of 2^32 -- about one per 3*10^8 hashes --, duplicate neighbours).
usage: python tests/stress_parity.py [n_seeds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argparse import Namespace
import subgraph_sketching_amd as ssa
from oracle import oracle

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
t = ssa.hll_tables.load(8, prefer='regenerated')
prm = oracle.HllParams(t.p, t.threshold, t.raw_estimate, t.bias, alpha=t.alpha,
                       lc_table=ssa.hashing.linear_counting_table(256).numpy())
bad = 0
hashes = 0
t0 = time.time()
for seed in range(n_seeds):
    rng = np.random.RandomState(10_000 + seed)
    n = int(rng.choice([235868, 150001, 300007]))
    e_und = int(rng.choice([1179052, 600000, 2000000]))
    if seed % 3 == 2:
        w = np.arange(1, n + 1, dtype=np.float64) ** -0.6
        cdf = np.cumsum(w / w.sum())
        e = np.stack([np.minimum(np.searchsorted(cdf, rng.random_sample(e_und)), n - 1), rng.randint(0, n, size=e_und)])
    else:
        e = rng.randint(0, n, size=(2, e_und))
    if seed % 4 == 1:
        e[:, : e_und // 20] = e[:, e_und // 20: 2 * (e_und // 20)]  # 5 % duplicate edges
    ei = np.concatenate([e, e[::-1]], axis=1).astype(np.int64)
    eh = ssa.ElphHashes(Namespace(max_hash_hops=2, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
    eh.hll_tables = t
    table, cards = eh.build_hash_tables(n, torch.from_numpy(ei).to(dev))
    otab, ocards = oracle.build_hash_tables(n, ei, 2, 128, prm)
    ok = True
    for k in (1, 2):
        ok &= np.array_equal(table[k].mh_u32.cpu().numpy().view(np.uint32), otab[k]['minhash'])
        ok &= np.array_equal(table[k].hll_u8.cpu().numpy(), otab[k]['hll'])
    ok &= bool(np.allclose(cards.cpu().numpy(), ocards, rtol=1e-5, atol=1e-4))
    hashes += (ei.shape[1] + n) * 128
    bad += not ok
    print(f'seed {seed}: n={n} e_dir={ei.shape[1]} {"ok" if ok else "MISMATCH"}', flush=True)
print(f'{n_seeds} graphs, {hashes / 1e9:.1f} G first-hop hash evaluations, {bad} mismatches, {time.time() - t0:.0f} s')
sys.exit(1 if bad else 0)
