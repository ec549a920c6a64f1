"""probe: ss_csr_build at ogbl-ppa / ogbl-citation2 size (uniform and rank^-0.5 endpoints): checked against torch
(rowptr from bincount, every row the same multiset of sources) and timed (wall clock per build including the host side: use tools/kstats_cmd.sh for kernel times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import subgraph_sketching_amd as ssa
dev = torch.device('cuda:0')


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def check(csr, ei, n):
    src, dst = ei[0], ei[1]
    e = src.numel()
    deg = torch.bincount(dst, minlength=n)
    want = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    want[1:] = torch.cumsum(deg, 0)
    ok = torch.equal(csr.rowptr, want)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    got = torch.sort(rows * n + csr.col[:e].to(torch.int64))[0]
    del rows
    ref = torch.sort(dst * n + src)[0]
    return ok and torch.equal(got, ref) and int(csr.n_self_dev.item()) == int(ei.max()) + 1


names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['ppa', 'citation2', 'collab']
KINDS = ({},) if '--uniform' in sys.argv else (dict(kind='powerlaw', alpha=0.5),) if '--pl05' in sys.argv else ({}, dict(kind='powerlaw', alpha=0.5), dict(kind='powerlaw', alpha=0.9))
for name in names:
    cfg = bench.CONFIGS[name]
    for kw in KINDS:
        n = cfg['n']
        ei = torch.from_numpy(bench.synthetic_graph(n=n, e_und=cfg['e_und'], **kw)).to(dev)
        csr = ssa.build_csr(ei, n, dev, check=False)
        good = check(csr, ei, n)
        us = timeit(lambda: ssa.build_csr(ei, n, dev, check=False))
        E = ei.size(1)
        print(f'{name:10s} {str(kw):40s} ok={good} {us:9.1f} us  {(20 * E + 8 * n) / us / 1e3:7.1f} GB/s on 20E+8N  hubs={int(csr.hub_count)}', flush=True)
        del csr, ei
        torch.cuda.empty_cache()
