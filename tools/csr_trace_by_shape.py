"""per-shape kernel times of ss_csr_build from a rocprofv3 kernel trace of tools/probe_csr_large.py (which builds each shape 12 times:
1 checked + 1 warm-up + 10 timed): python tools/csr_trace_by_shape.py <kernel_trace.csv> <shape> [<shape> ...]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'ss::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
builds, cur = [], None
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').split('(')[0].split('::')[1].split('<')[0]
    if name == 'tile_sort_kernel':
        cur = {}
        builds.append(cur)
    if cur is not None:
        cur[name] = cur.get(name, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
shapes = sys.argv[2:]
per = len(builds) // max(len(shapes), 1)
for i, s in enumerate(shapes):
    bs = builds[i * per + 2:(i + 1) * per]
    keys = sorted({k for b in bs for k in b})
    avg = {k: sum(b.get(k, 0.0) for b in bs) / len(bs) for k in keys}
    print(f'{s:22s} sum {sum(avg.values()):8.1f} us   ' + '  '.join(f'{k.replace("_kernel", "")} {v:.1f}' for k, v in avg.items()))
