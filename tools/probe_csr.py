"""probe: ss_csr_build on the bench graph (and with explicit self loops, the ELPH shape): the two-launch gather plan
(tile_sort + finish_gather) against the five-launch partition plan (SS_CSR_NO_GATHER=1); also a power-law graph"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import subgraph_sketching_amd as ssa
dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for name, kw in (('uniform', {}), ('powerlaw 0.5', dict(kind='powerlaw', alpha=0.5)), ('powerlaw 0.9', dict(kind='powerlaw', alpha=0.9))):
    n = bench.N_NODES
    ei = torch.from_numpy(bench.synthetic_graph(**kw)).to(dev)
    for plan in ('gather', 'partition'):
        os.environ.pop('SS_CSR_NO_GATHER', None)
        if plan == 'partition':
            os.environ['SS_CSR_NO_GATHER'] = '1'
        us = timeit(lambda: ssa.build_csr(ei, n, dev, check=False))
        csr = ssa.build_csr(ei, n, dev, check=False)
        print(f'{name:14s} {plan:10s} {us:8.1f} us   rowptr[-1]={int(csr.rowptr[-1])} hubs={int(csr.hub_count)} n_self={int(csr.n_self_dev)}', flush=True)
os.environ.pop('SS_CSR_NO_GATHER', None)
