"""turn gpurun_out/prof_<tag>/ (rocprofv3 csv) into profiles/<tag>_kernel_stats.csv and profiles/pmc_traffic.json
usage: python tools/summarise_prof.py <tag> [--set-traffic] [--shape KEY]
--shape KEY: also record the per-family fabric bytes of this run under pmc_traffic.json["shapes"][KEY] (bench.py reads them: the
`traffic` of its roofline objects, and the cap on the bytes of a row kernel whose sources repeat)"""
import csv, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'prof_{tag}')
os.makedirs(os.path.join(root, 'profiles'), exist_ok=True)
rows = list(csv.DictReader(open(os.path.join(src, f'{tag}_kernel_stats.csv'))))
with open(os.path.join(root, 'profiles', f'{tag}_kernel_stats.csv'), 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (tag {tag}); see also {tag}_bench.log\n')
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
        w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
log = os.path.join(root, 'gpurun_out', f'prof_{tag}_bench.log')
if os.path.exists(log):
    open(os.path.join(root, 'profiles', f'{tag}_bench.log'), 'w').write(open(log).read())
pmc = {}
for kind in ('fetch', 'write'):
    path = os.path.join(src, f'{tag}_{kind}_counter_collection.csv')
    if not os.path.exists(path):
        continue
    acc = {}
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0]
        acc.setdefault(k, []).append(float(r['Counter_Value']))
    for k, v in acc.items():
        pmc.setdefault(k, {})[f"{'FETCH' if kind == 'fetch' else 'WRITE'}_SIZE_KB_avg"] = sum(v) / len(v)
if pmc:
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB per dispatch). gfx950 correction per '
                   'MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of the bytes of dwordx4 streams -> x2; WRITE_SIZE calibrates '
                   'exactly on the store-only init kernels (minhash_init: N*512 B).', 'raw': pmc}
    for k, v in pmc.items():
        if 'propagate_kernel<128, 256>' in k and 'hub' not in k:
            hbm = 2 * v.get('FETCH_SIZE_KB_avg', 0) * 1024 + v.get('WRITE_SIZE_KB_avg', 0) * 1024
            out['propagate_kernel_hbm_bytes_per_launch'] = hbm
        if 'pair_features_kernel<2, 128, 256' in k:
            out['pair_features_kernel_hbm_bytes_per_launch'] = 2 * v.get('FETCH_SIZE_KB_avg', 0) * 1024 + v.get('WRITE_SIZE_KB_avg', 0) * 1024
    # per kernel family: bytes per launch that crossed the fabric (2 x FETCH_SIZE + WRITE_SIZE, see the note)
    def fabric(v):
        return 2 * v.get('FETCH_SIZE_KB_avg', 0) * 1024 + v.get('WRITE_SIZE_KB_avg', 0) * 1024
    fam = {}
    csr_kernels = ('tile_sort_kernel', 'regroup_sort_kernel', 'finish_runs_kernel', 'level_scan_kernel', 'level_tiles_kernel',
                   'level_fill_kernel', 'dense_count_runs_kernel', 'dense_place_runs_kernel')
    for k, v in pmc.items():
        if 'propagate_kernel<128, 256>' in k and 'hub' not in k:
            fam['minhash_hop'] = fabric(v)
        elif 'fused_hop_persistent_kernel' in k:
            fam['fused_first_hop_hll_hop'] = fabric(v)
        elif 'pair_features_kernel<' in k or 'pair_features_runs_kernel<' in k:
            fam['pair_features'] = fam.get('pair_features', 0) + fabric(v)
        elif 'hll_propagate_row16_kernel' in k:
            fam['hll_hop'] = fabric(v)
        elif 'hll_first_hop_kernel' in k:
            fam['first_hop_hll'] = fabric(v)
        elif any(c in k for c in csr_kernels):
            fam['csr_build'] = fam.get('csr_build', 0) + fabric(v)  # (one launch of each per build)
    out['families'] = fam
    json.dump(out, open(os.path.join(root, 'profiles', f'{tag}_pmc.json'), 'w'), indent=1)
    traffic_path = os.path.join(root, 'profiles', 'pmc_traffic.json')
    if '--set-traffic' in sys.argv:
        keep = json.load(open(traffic_path)).get('shapes', {}) if os.path.exists(traffic_path) else {}
        json.dump(dict(out, shapes=keep), open(traffic_path, 'w'), indent=1)
    if '--shape' in sys.argv:
        key = sys.argv[sys.argv.index('--shape') + 1]
        blob = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
        blob.setdefault('shapes', {})[key] = dict(fam, file=f'profiles/{tag}_pmc.json')
        json.dump(blob, open(traffic_path, 'w'), indent=1)
for r in rows[:14]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
print({k: v for k, v in (pmc or {}).items() if 'propagate' in k or 'pair' in k or 'first' in k})
