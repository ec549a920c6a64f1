#!/bin/bash
# usage: bash tools/import_external_fixtures.sh [--dry-run] <dir>
#   --dry-run: validate and report exactly as below but install NOTHING (tests/test_host_logic.py runs it on a copy of the committed
#   regenerated tables written in the exporter's format, so that this five-minute path cannot rot unnoticed)
# <dir> holds what the two exporters wrote on a machine that has the third-party packages this image lacks:
#   hllpp_tables_datasketch.npz  (+ g11_datasketch_tables.npz)   tools/export_datasketch_fixture.py   (needs datasketch)
#   g13_pyg_sign.npz                                             tools/export_pyg_fixture.py          (needs torch_geometric + torch_sparse)
# Either may be missing.  The files are validated, installed where the engine / the tests look for them, the golden vectors are
# regenerated from the reference WITH THE REAL TABLES into a scratch directory and compared with the committed ones: which arrays
# changed, on how many entries and by how much (only values on the bias-corrected branch -- the `*_uses_tables` masks -- may move).
# Nothing is committed: the script ends with the git commands to run after a look at the report.
set -e
DRY=0
if [ "$1" = "--dry-run" ]; then DRY=1; shift; fi
SRC=${1:?usage: bash tools/import_external_fixtures.sh [--dry-run] <dir>}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
python - "$SRC" "$DRY" <<'PY'
import os, shutil, subprocess, sys, tempfile
import numpy as np
src, root, dry = sys.argv[1], os.getcwd(), sys.argv[2] == '1'
installed = []
_copy = shutil.copyfile
if dry:
    shutil.copyfile = lambda a, b: print(f'(dry run: would install {os.path.relpath(b, root)})')

def fail(msg):
    sys.exit(f'import_external_fixtures: {msg}')

tables = os.path.join(src, 'hllpp_tables_datasketch.npz')
if os.path.exists(tables):
    z = np.load(tables)
    if 'p_list' not in z.files:
        fail(f'{tables}: no p_list -- not written by tools/export_datasketch_fixture.py')
    ps = [int(p) for p in z['p_list']]
    if 8 not in ps:
        fail(f'{tables}: precision 8 (the reference default) is missing')
    for p in ps:
        for key in (f'alpha_p{p}', f'max_rank_p{p}', f'threshold_p{p}', f'raw_p{p}', f'bias_p{p}'):
            if key not in z.files:
                fail(f'{tables}: {key} missing')
        raw, bias = z[f'raw_p{p}'], z[f'bias_p{p}']
        if raw.shape != bias.shape or raw.ndim != 1 or raw.size < 6 or raw.size > 512:
            fail(f'{tables}: p={p}: raw / bias shapes {raw.shape} / {bias.shape} (want equal 1-D, 6..512 entries)')
        if not (np.isfinite(raw).all() and np.isfinite(bias).all()):
            fail(f'{tables}: p={p}: non-finite entries')
        if int(z[f'max_rank_p{p}']) != 64 - p:
            fail(f'{tables}: p={p}: max_rank {int(z[f"max_rank_p{p}"])} != 64 - p (reference hashing.py:76 asserts it)')
    dst = os.path.join(root, 'subgraph-sketching_amd', 'data', 'hllpp_tables_datasketch.npz')
    shutil.copyfile(tables, dst)
    installed.append(dst)
    g11 = os.path.join(src, 'g11_datasketch_tables.npz')
    if os.path.exists(g11):
        dst11 = os.path.join(root, 'tests', 'golden', 'g11_datasketch_tables.npz')
        shutil.copyfile(g11, dst11)
        installed.append(dst11)
    print(f'datasketch tables for p in {ps} installed')
else:
    print(f'(no {tables}: the HLL++ bias branch stays on regenerated tables)')

g13 = os.path.join(src, 'g13_pyg_sign.npz')
if os.path.exists(g13):
    z = np.load(g13)
    need = [k for k in ('edge_index', 'x') if not any(f.startswith(k) for f in z.files)]
    if need or not any(f.startswith('sign_k0') for f in z.files) or not any(f.startswith('sign_k2') for f in z.files):
        fail(f'{g13}: not written by tools/export_pyg_fixture.py (missing {need or "sign_k0 / sign_k2 outputs"})')
    dst = os.path.join(root, 'tests', 'golden', 'g13_pyg_sign.npz')
    shutil.copyfile(g13, dst)
    installed.append(dst)
    print('PyG / torch_sparse SIGN fixture installed')
else:
    print(f'(no {g13}: gcn_norm / spmm stay "PyG semantics restated")')

if os.path.exists(tables):
    # the golden vectors again, from the reference, with the real tables -- what moves?
    if not os.path.isdir('/root/reference'):
        print('(/root/reference is not here: the golden vectors cannot be regenerated on this machine -- run the report where it is)')
    else:
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, SS_GOLDEN_TABLES=tables, SS_GOLDEN_OUT=tmp)
            subprocess.run([sys.executable, os.path.join('tests', 'golden', 'make_golden.py')], check=True, env=env, stdout=subprocess.DEVNULL)
            changed_arrays = 0
            print('\ngolden vectors regenerated with datasketch\'s tables vs the committed ones (regenerated tables):')
            for name in sorted(f for f in os.listdir(tmp) if f.endswith('.npz')):
                new, old = np.load(os.path.join(tmp, name)), np.load(os.path.join('tests', 'golden', name))
                for key in sorted(new.files):
                    if key not in old.files or new[key].dtype.kind not in 'fiu' or new[key].shape != old[key].shape:
                        continue
                    a, b = new[key].astype(np.float64), old[key].astype(np.float64)
                    moved = ~np.isclose(a, b, rtol=0, atol=0, equal_nan=True)
                    if moved.any():
                        changed_arrays += 1
                        print(f'  {name}:{key}: {int(moved.sum())} of {moved.size} entries changed, max |diff| = {np.nanmax(np.abs(a - b)[moved]):.6g}')
            print(f'arrays that moved: {changed_arrays}')
            print('(arrays not listed are unchanged.  To adopt the real tables as the pinned ones: re-run tests/golden/make_golden.py with\n'
                  ' SS_GOLDEN_TABLES set and commit its output together with the installed files)')

print('\ninstalled:', *installed, sep='\n  ')
print('\nnext: python -m pytest tests -q -m "not gpu"   (test_datasketch_tables_fixture / test_pyg_sign_fixture no longer skip)\n'
      '      gpurun -- python -m pytest tests -q -m gpu\n      git add ' + ' '.join(os.path.relpath(p, root) for p in installed))
PY
