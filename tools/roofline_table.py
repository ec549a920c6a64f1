#!/usr/bin/env python3
"""Per-kernel roofline table of one profiled run, recomputed from TRACKED files only:
   profiles/<tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats: average duration per kernel)
   profiles/<tag>_pmc.json          (separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB per dispatch)
   subgraph-sketching_amd/roofline.py (algorithmic bytes per launch)
usage: python tools/roofline_table.py <tag> <config> [--graph powerlaw --alpha a]     -> profiles/<tag>_roofline.md
HBM bytes from PMC = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies the 128-B requests of dwordx4 streams
at 64 B, MI355X_MICROARCH.md "HBM"); the counters sit on the fabric side of the L2 and INCLUDE Infinity-Cache hits, so for a
table that fits the 256 MiB cache they show "no L2 re-reads", not HBM bytes -- the `resident` column says which case applies."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (CONFIGS only)
import subgraph_sketching_amd as ssa  # noqa: E402

FAMILIES = [('propagate_hub_kernel', 'hub_table_hop'), ('first_hop_hub_kernel', 'hub_first_hop'),
            ('propagate_kernel<128, 256>', 'minhash_hop'), ('hll_propagate_row16_kernel', 'hll_hop'),
            ('first_hop_kernel<2, true, false>', 'first_hop_minhash'), ('first_hop_rows_kernel<2', 'first_hop_minhash'),
            ('hll_first_hop_kernel', 'first_hop_hll'), ('fused_hop_persistent_kernel', 'fused_first_hop_hll_hop'),
            ('pair_features_kernel', 'pair_features')]


def main():
    tag, config = sys.argv[1], sys.argv[2]
    cfg = bench.CONFIGS[config]
    n, e, h, b = cfg['n'], 2 * cfg['e_und'], cfg['h'], cfg['batch']
    rf = ssa.roofline
    hub_e = hub_n = 0
    if '--graph' in sys.argv:  # skewed shapes: the rows above the hub threshold belong to the hub passes, not to the row kernels
        import numpy as np
        kind = sys.argv[sys.argv.index('--graph') + 1]
        alpha = float(sys.argv[sys.argv.index('--alpha') + 1]) if '--alpha' in sys.argv else 0.5
        hub_e, hub_n = bench.hub_stats(ssa, bench.synthetic_graph(n, cfg['e_und'], kind, alpha), n)
    model = rf.kernel_bytes(n, e, 128, 8, h, b, hub_e, hub_n, bench.HUB_UNITS_HOSTED)
    stats = {r['Name']: r for r in csv.DictReader(l for l in open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv')) if not l.startswith('#'))}
    pmc_path = os.path.join(ROOT, 'profiles', f'{tag}_pmc.json')
    pmc = json.load(open(pmc_path))['raw'] if os.path.exists(pmc_path) else {}
    lines = [f'# {tag}: per-kernel roofline ({config}-like, N={n}, E_dir={e}, h={h}, B={b}; peak {rf.HBM_PEAK_GBS:.0f} GB/s)', '',
             '| kernel | calls | avg us | algorithmic bytes / launch | achieved GB/s | frac of peak | PMC bytes / launch | PMC / algorithmic | resident |',
             '|---|---|---|---|---|---|---|---|---|']
    total_ns, ours_ns = 0.0, 0.0
    for name, r in stats.items():
        if not (name.startswith('void ss::') or name.startswith('ss::')):
            continue
        short = name.split('(')[0].replace('void ', '')
        avg = float(r['AverageNs'])
        ours_ns += float(r['TotalDurationNs'])
        fam = next((f for key, f in FAMILIES if key in name), None)
        raw = next((v for k, v in pmc.items() if k.replace('void ', '') == short), None)
        hbm = (2 * raw.get('FETCH_SIZE_KB_avg', 0) + raw.get('WRITE_SIZE_KB_avg', 0)) * 1024 if raw else None
        if fam and model[fam] > 0:
            alg = model[fam]
            res = (f'{rf.residency(n, fam, 128, 8, h)} ({rf.cache_resident_fraction(rf.gathered_table_bytes(n, fam, 128, 8, h)):.2f} cached)'
                   if fam in ('minhash_hop', 'hll_hop', 'fused_first_hop_hll_hop', 'pair_features') else '-')
            # the fraction is taken on the algorithmic bytes CAPPED at the bytes the PMC passes saw crossing the fabric when those are
            # fewer (a row kernel on a skewed graph re-reads source rows out of the L2: repeats are not traffic) -- the rule of
            # bench.py's roofline_numbers, so that no tracked table prints a fraction above 1
            basis = hbm if (hbm and hbm < 0.97 * alg) else alg
            mark = ' (PMC bytes)' if basis is not alg else ''
            lines.append(f'| `{short}` | {r["Calls"]} | {avg / 1e3:.1f} | {alg / 1e9:.4f} GB | {basis / avg:.0f} | {basis / avg / rf.HBM_PEAK_GBS:.3f}{mark} | '
                         f'{hbm / 1e9:.4f} GB | {hbm / alg:.3f} | {res} |' if hbm else
                         f'| `{short}` | {r["Calls"]} | {avg / 1e3:.1f} | {alg / 1e9:.4f} GB | {alg / avg:.0f} | {alg / avg / rf.HBM_PEAK_GBS:.3f} | - | - | {res} |')
        else:
            lines.append(f'| `{short}` | {r["Calls"]} | {avg / 1e3:.1f} | - | - | - | {hbm / 1e6:.2f} MB | - | - |' if hbm else
                         f'| `{short}` | {r["Calls"]} | {avg / 1e3:.1f} | - | - | - | - | - | - |')
    # the CSR build as one family: its kernels summed (one launch of each per build) against the algorithmic 20E + 8N
    csr_keys = ('tile_sort_kernel', 'regroup_sort_kernel', 'finish_runs_kernel', 'level_scan_kernel', 'level_tiles_kernel', 'level_fill_kernel',
                'dense_count_runs_kernel', 'dense_place_runs_kernel')
    csr_us = sum(float(r['AverageNs']) for nme, r in stats.items() if any(k in nme for k in csr_keys)) / 1e3
    csr_pmc = sum((2 * v.get('FETCH_SIZE_KB_avg', 0) + v.get('WRITE_SIZE_KB_avg', 0)) * 1024 for k, v in pmc.items() if any(c in k for c in csr_keys))
    if csr_us:
        alg = model['csr_build']
        lines.append(f'| **CSR build: the kernels above it is made of, summed** | 1 each | {csr_us:.1f} | {alg / 1e9:.4f} GB (20E + 8N) | {alg / csr_us / 1e3:.0f} | '
                     f'{alg / csr_us / 1e3 / rf.HBM_PEAK_GBS:.3f} | ' + (f'{csr_pmc / 1e9:.4f} GB | {csr_pmc / alg:.2f} | - |' if csr_pmc else '- | - | - |'))
    calls = max(int(r['Calls']) for nme, r in stats.items() if 'propagate_kernel<128' in nme) // max(h - 1, 1)
    step_us = ours_ns / 1e3 / calls
    lines += ['', f'Sum of the engine\'s kernels per step (kernel time only, {calls} steps traced): **{step_us:.1f} us**; bytes of the implemented '
                  f'schedule per step {rf.step_bytes_implemented(n, e, 128, 8, h, b, hub_e, hub_n) / 1e9:.3f} GB '
                  f'-> {rf.step_bytes_implemented(n, e, 128, 8, h, b, hub_e, hub_n) / step_us / 1e3:.0f} GB/s = '
                  f'{rf.step_bytes_implemented(n, e, 128, 8, h, b, hub_e, hub_n) / step_us / 1e3 / rf.HBM_PEAK_GBS:.3f} of peak over kernel time.']
    out = os.path.join(ROOT, 'profiles', f'{tag}_roofline.md')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
