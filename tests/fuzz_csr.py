"""fuzz of the CSR builder at sizes that take multi-level plans (more than 256 fine buckets: the finish launch's image-by-image path for
buckets of up to three LDS images, the share protocol above that, the fallback when ONE row exceeds the image): random node / edge
counts, endpoint skews (power law over ids, dense id windows, single hub rows of random size, uniform), every build checked against
torch (rowptr = cumsum(bincount), every row the same multiset of sources), n_self and the protocol-fault stamp.
usage (GPU box): python tests/fuzz_csr.py [seconds = 120]      env FUZZ_SEED, SS_CSR_WALK_IMAGES (1 .. 3 forces the limit)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import subgraph_sketching_amd as ssa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device('cuda:0')
lib = ssa._native.lib()
rng = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '2024')))
g = torch.Generator(device=dev).manual_seed(int(os.environ.get('FUZZ_SEED', '2024')))
t0, trials, bad = time.time(), 0, 0
while time.time() - t0 < seconds:
    n = int(rng.choice([270_000, 400_000, 900_000, 2_000_000, 3_100_000]))
    e = int(rng.choice([1_000_000, 4_000_000, 12_000_000, 30_000_000]))
    kind = rng.choice(['uniform', 'powerlaw', 'windows', 'hubs', 'mixed'])
    dst = torch.randint(0, n, (e,), device=dev, generator=g)
    if kind in ('powerlaw', 'mixed'):
        alpha = float(rng.choice([0.3, 0.5, 0.9, 1.2]))
        w = torch.arange(1, n + 1, device=dev, dtype=torch.float64) ** -alpha
        k = e // 2 if kind == 'mixed' else e
        dst[:k] = torch.multinomial(w / w.sum(), k, replacement=True, generator=g)
    if kind in ('windows', 'mixed'):  # dense id windows: buckets of one to a few images next to empty ones
        for _ in range(int(rng.randint(1, 6))):
            lo, width, cnt = int(rng.randint(0, n - 5000)), int(rng.choice([40, 300, 2000])), int(rng.choice([20_000, 45_000, 120_000]))
            at = int(rng.randint(0, e - cnt))
            dst[at:at + cnt] = torch.randint(lo, lo + width, (cnt,), device=dev, generator=g)
    if kind in ('hubs', 'mixed'):  # single rows around the image size (16 384): inside a walkable bucket they force the share protocol
        for _ in range(int(rng.randint(1, 5))):
            node, cnt = int(rng.randint(0, n)), int(rng.choice([9_000, 17_000, 40_000, 200_000]))
            at = int(rng.randint(0, e - cnt))
            dst[at:at + cnt] = node
    src = torch.randint(0, n, (e,), device=dev, generator=g)
    csr = ssa.build_csr(torch.stack([src, dst]), n, dev, check=False)
    deg = torch.bincount(dst, minlength=n)
    want = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    want[1:] = torch.cumsum(deg, 0)
    ok = torch.equal(csr.rowptr, want)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    ok = ok and torch.equal(torch.sort(rows * n + csr.col[:e].long())[0], torch.sort(dst * n + src)[0])
    ok = ok and int(csr.n_self_dev.item()) == int(max(int(src.max()), int(dst.max()))) + 1 and lib.ss_csr_protocol_faults() == 0
    trials += 1
    if not ok:
        bad += 1
        print(f'MISMATCH: n={n} e={e} kind={kind}', flush=True)
    del csr, rows, want, deg, src, dst
print(f'{trials} builds, {bad} mismatches, {time.time() - t0:.0f} s')
sys.exit(1 if bad else 0)
