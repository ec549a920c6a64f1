#!/usr/bin/env python3
"""Regenerate HyperLogLog++ empirical bias-correction tables by simulation.

Why this exists
---------------
The reference reads `datasketch.hyperloglog_const._thresholds/_bias/_raw_estimate`
(/root/reference/src/hashing.py:78-80).  `datasketch` is an un-vendored, un-pinned
third-party dependency (/root/reference/README.md:45) that is absent from the build
image and from the GPU boxes, and there is no network.  The tables are *inputs* of
our engine (see hll_tables.py): when `datasketch` is importable its tables are used
verbatim; otherwise the tables produced by this script are used and every result that
went through the bias-corrected branch is labelled "regenerated tables".

Procedure (HLL++ paper, Heule/Nunkesser/Hall 2013, section 5.2 / appendix): for each
precision p pick 200 cardinalities in [0, ~5m]; for each, average the *raw* HLL
estimate alpha*m^2/sum(2^-M[j]) over many random sketches; record the mean raw
estimate and the bias (mean raw estimate - true cardinality).

The sketch distribution is simulated exactly: bucket occupancy ~ Multinomial(n, 1/m),
register value = max of k iid Geometric(1/2) ranks, drawn by inverse CDF
(P[max <= r] = (1 - 2^-r)^k).

Output: hllpp_tables_regenerated.npz with arrays raw_p{p}, bias_p{p} (float64) and
`thresholds` (the HLL++ paper's published linear-counting thresholds for p=4..18) and
meta fields.  Deterministic for a fixed seed.
"""
import sys
import time
import numpy as np

THRESHOLDS = [10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000]
NPTS = 200


def alpha(p):
    m = 1 << p
    if p == 4:
        return 0.673
    if p == 5:
        return 0.697
    if p == 6:
        return 0.709
    return 0.7213 / (1.0 + 1.079 / m)


def simulate(p, n, trials, rng):
    """mean raw estimate over `trials` random sketches holding n distinct items"""
    m = 1 << p
    if n == 0:
        return alpha(p) * m  # all registers zero: alpha*m^2/m
    tot = 0.0
    done = 0
    chunk = max(1, min(trials, (1 << 22) // m))
    pv = np.full(m, 1.0 / m)
    while done < trials:
        t = min(chunk, trials - done)
        k = rng.multinomial(n, pv, size=t).astype(np.float64)  # [t, m] bucket loads
        u = rng.random((t, m))
        # r = smallest integer with (1-2^-r)^k >= u  ->  r = ceil(-log2(1 - u^(1/k)))
        with np.errstate(divide='ignore', invalid='ignore'):
            x = 1.0 - np.power(u, 1.0 / np.maximum(k, 1.0))
            r = np.ceil(-np.log2(np.maximum(x, 2.0 ** -80)))
        r = np.clip(r, 1, 64 - p + 1)
        r = np.where(k > 0, r, 0.0)
        s = np.sum(np.exp2(-r), axis=1)
        tot += np.sum(alpha(p) * m * m / s)
        done += t
    return tot / trials


def one_precision(p):
    m = 1 << p
    t0 = time.time()
    rng = np.random.default_rng([20260928, p])   # independent, reproducible stream per precision
    trials = max(48, (1 << 23) // m)
    ns = np.unique(np.round(np.linspace(0, 5.15 * m, NPTS)).astype(np.int64))
    raw = np.array([simulate(p, int(n), trials, rng) for n in ns])
    order = np.argsort(raw, kind='stable')
    print(f'p={p} m={m} pts={len(ns)} trials={trials} {time.time() - t0:.1f}s '
          f'raw[0]={raw[0]:.3f} raw[-1]={raw[-1]:.3f} bias[0]={raw[0]-ns[0]:.3f} bias[-1]={raw[-1]-ns[-1]:.3f}',
          flush=True)
    return p, raw[order], (raw - ns)[order]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else 'hllpp_tables_regenerated.npz'
    pmin, pmax = 4, 16
    arrays = {'thresholds': np.asarray(THRESHOLDS, dtype=np.float64),
              'p_min': np.asarray(pmin), 'p_max': np.asarray(pmax)}
    from multiprocessing import Pool
    with Pool(min(8, pmax - pmin + 1)) as pool:
        for p, raw, bias in pool.imap_unordered(one_precision, range(pmin, pmax + 1)):
            arrays[f'raw_p{p}'] = raw
            arrays[f'bias_p{p}'] = bias
    np.savez_compressed(out, **arrays)
    print('wrote', out)


if __name__ == '__main__':
    main()
