#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

metric   : edge-pairs/sec subgraph-feature extraction (build+query)
workload : configs[1] "ogbl-collab, BUDDY, max_hash_hops=2, batch_size=65536" as a synthetic graph of the same
           shape: N=235 868 nodes, E_und=1 179 052 uniform-random undirected edges
           (seed 1, both directions -> E_dir=2 358 104), P=128, p=8, h=2; one batch of B=65 536 node pairs per
           step and per GPU (seed 2+rank).
step     : one pass of the whole hot path over one batch: ElphHashes.build_hash_tables (CSR build, hop-0
           sketches, h propagation hops with fused cardinalities) + ElphHashes.get_subgraph_features(B pairs),
           i.e. what ELPH does per training step (reference runners/train.py:198,204) and what BUDDY does once
           per edge set.  Nothing is cached between steps.  Inputs (edge_index, links) are resident in HBM.
N > 1    : one process per GPU (torchrun), sketch table replicated (every rank builds it), edge batches
           sharded -- each rank owns its own B pairs -- and the per-batch feature rows all-gathered over
           RCCL (async all_gather_into_tensor on RCCL's stream, overlapped with the next step's build, double-buffered,
           every gather completed inside the timed region).  Weak scaling.
roofline : dominant kernel = ss::propagate_kernel (one launch per hop).  achieved = algorithmic bytes per
           launch ((E'+N)*768 + 4E' + 8(N+1) + 4N, E' = E_dir + N) / mean launch
           duration measured live in the timed region with HIP events on the launch stream.
cpu_baseline : the oracle's C port (oracle/sketch_oracle.c, OpenMP on all host cores) timed on ONE full step
           of the same workload, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs as synthetic shapes.  The default -- and the only one the driver's
# plain `python bench.py` measures -- is configs[1] (collab).  The others are selectable for profiling.
CONFIGS = {
    'collab': dict(n=235868, e_und=1179052, h=2, batch=65536),      # configs[1] / [2]
    'cora': dict(n=2485, e_und=3550, h=2, batch=1024),              # configs[0] shape (plumbing)
    'ppa': dict(n=576289, e_und=21231931, h=2, batch=131072),       # configs[3]
    'citation2': dict(n=2927963, e_und=30387995, h=3, batch=261424),  # configs[4]
}
N_NODES, E_UND, H, P, HLL_P, BATCH = 235868, 1179052, 2, 128, 8, 65536
ROW_BYTES = 4 * P + (1 << HLL_P)
HBM_PEAK_GBS = 8000.0
GRAPH_KIND, PL_ALPHA = 'uniform', 0.5


def synthetic_graph(seed=1):
    rng = np.random.RandomState(seed)
    if GRAPH_KIND == 'uniform':
        e = rng.randint(0, N_NODES, size=(2, E_UND)).astype(np.int64)
    else:  # power-law endpoint weights w_i ~ (i+1)^-alpha (Chung-Lu style): exercises hub rows
        w = np.arange(1, N_NODES + 1, dtype=np.float64) ** -PL_ALPHA
        cdf = np.cumsum(w / w.sum())
        e = np.stack([np.searchsorted(cdf, rng.random_sample(E_UND)), rng.randint(0, N_NODES, size=E_UND)]).astype(np.int64)
        e = np.minimum(e, N_NODES - 1)
    return np.concatenate([e, e[::-1]], axis=1)


def synthetic_links(seed):
    return np.random.RandomState(seed).randint(0, N_NODES, size=(BATCH, 2)).astype(np.int64)


class KernelTimer(object):
    """HIP-event pairs around the engine's launches, on the stream they are launched on"""

    def __init__(self, only=None):
        self.events = {}
        self.only = only  # None = time every launch; else only these span names (fewer event packets in the stream)

    def wants(self, name):
        return self.only is None or name in self.only

    def record(self, name, stream):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def span(self, name, start, end):
        self.events.setdefault(name, []).append((start, end))

    def mean_ms(self, name):
        spans = self.events.get(name, [])
        return sum(a.elapsed_time(b) for a, b in spans) / max(len(spans), 1), len(spans)


def cpu_baseline(ei, links):
    """oracle C port (OpenMP) on one full step; returns the cpu_baseline object"""
    import subgraph_sketching_amd as ssa
    from oracle import oracle
    t = ssa.hll_tables.load(HLL_P)
    prm = oracle.HllParams(t.p, t.threshold, t.raw_estimate, t.bias, alpha=t.alpha,
                           lc_table=ssa.hashing.linear_counting_table(1 << t.p).numpy())
    cores = os.cpu_count()
    oracle.lib()
    reps = 5 if N_NODES > 100000 else 50
    t_build = t_query = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        rowptr, col = oracle.csr_build(N_NODES, ei)
        n_self = int(ei.max()) + 1
        mh, hll = oracle.minhash_init(N_NODES, P), oracle.hll_init(N_NODES, HLL_P)
        tables, cards = {0: {'minhash': mh, 'hll': hll}}, np.zeros((N_NODES, H), dtype=np.float32)
        for k in range(1, H + 1):
            mh, hll, c = oracle.propagate_csr(N_NODES, rowptr, col, n_self, mh, hll, prm)
            tables[k] = {'minhash': mh, 'hll': hll}
            cards[:, k - 1] = c
        t1 = time.perf_counter()
        feats = oracle.pair_features(links, tables, cards, H, prm)
        t2 = time.perf_counter()
        t_build += t1 - t0
        t_query += t2 - t1
    return {'value': reps * BATCH / (t_build + t_query), 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} full steps (build N={N_NODES}, E_dir={2 * E_UND}, h={H} + {BATCH} pairs each); mean '
                      f'build {t_build / reps:.2f} s, query {t_query / reps:.3f} s; C/OpenMP restatement of the reference, '
                      f'not the torch/PyG code itself'}, feats


def cpu_baseline_reference_style(ei, links):
    """the reference's dataflow in stock torch CPU ops (oracle/torch_refstyle.py): materialised per-edge messages +
    scatter-amax, int64 MinHash, h^2 x 4 row gathers, argsort-based bias lookup.  One full step (all h hops + the query of
    one batch) is timed."""
    import subgraph_sketching_amd as ssa
    from oracle import oracle, torch_refstyle as tr
    t = ssa.hll_tables.load(HLL_P)
    raw, bias = torch.tensor(t.raw_estimate, dtype=torch.float), torch.tensor(t.bias, dtype=torch.float)
    threads = min(16, os.cpu_count())  # measured best on the 256-core GPU host (8: 3.5 s/hop, 16: 2.5, 32: 3.0, 128: 6.0)
    torch.set_num_threads(threads)
    mh0 = torch.from_numpy(oracle.minhash_init(N_NODES, P).astype(np.int64))
    hll0 = torch.from_numpy(oracle.hll_init(N_NODES, HLL_P).view(np.int8))
    t0 = time.perf_counter()
    tables, cards = tr.build_tables(N_NODES, torch.from_numpy(ei), H, mh0, hll0, HLL_P, t.alpha, t.threshold, raw, bias, hops_to_run=H)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    tr.pair_intersections(torch.from_numpy(links), tables, H, P, HLL_P, t.alpha, t.threshold, raw, bias)
    t_query = time.perf_counter() - t0
    return {'value': BATCH / (t_build + t_query), 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
            'sample': f'reference-style torch CPU ops, 1 full step: {H}-hop build {t_build:.2f} s + '
                      f'{BATCH}-pair query ({t_query:.3f} s); torch threads = {torch.get_num_threads()}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--config', default='collab', choices=sorted(CONFIGS), help='synthetic shape (default: BASELINE configs[1])')
    ap.add_argument('--graph', default='uniform', choices=['uniform', 'powerlaw'])
    ap.add_argument('--alpha', type=float, default=0.5, help='power-law exponent of the endpoint weights')
    ap.add_argument('--api', default='build_query', choices=['build_query', 'elph', 'buddy'],
                    help='build_query (default, the BASELINE metric): build_hash_tables + get_subgraph_features per step; '
                         'elph: the exact call sequence of ELPH.forward (models/elph.py:186-213) + one query per step; '
                         'buddy: one build amortised over --buddy-batches query batches (datasets/elph.py:200-208)')
    ap.add_argument('--buddy-batches', type=int, default=40)
    ap.add_argument('--build', default='replicated', choices=['replicated', 'sharded'],
                    help='N > 1 only. replicated (default): every rank builds the whole table; sharded: destination rows split '
                         'across ranks + in-place all-gather after every hop (pays off at ogbl-ppa / citation2 sizes)')
    ap.add_argument('--time-all-kernels', action='store_true', help='HIP-event spans around every launch (default: only the roofline kernel)')
    a = ap.parse_args()
    global N_NODES, E_UND, H, BATCH, GRAPH_KIND, PL_ALPHA
    cfg = CONFIGS[a.config]
    N_NODES, E_UND, H, BATCH, GRAPH_KIND, PL_ALPHA = cfg['n'], cfg['e_und'], cfg['h'], cfg['batch'], a.graph, a.alpha

    launched = 'RANK' in os.environ  # under torchrun (also with one rank, so the RCCL path can be smoke-tested)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if launched:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)
    assert a.gpus == world, f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {a.gpus}'

    import subgraph_sketching_amd as ssa
    from subgraph_sketching_amd import hashing
    ssa._native.lib()  # fail loudly if the HIP engine is missing
    eh = ssa.ElphHashes(Namespace(max_hash_hops=H, hll_p=HLL_P, minhash_num_perm=P, floor_sf=False, use_zero_one=True))
    eh.strict_bounds = False  # no host sync inside a step

    ei_np = synthetic_graph()
    links_np = synthetic_links(2 + rank)
    ei = torch.from_numpy(ei_np).to(dev)
    links = torch.from_numpy(links_np).to(dev)
    # the per-batch feature gather runs on RCCL's stream UNDER the next step's build (double-buffered); every gather is
    # completed inside the timed region (drain() before the closing fence)
    gathered = [torch.empty((world * BATCH, H * (H + 2)), dtype=torch.float32, device=dev) for _ in range(2)] if launched else None
    inflight = {'works': [], 'n': 0}

    def gather(f):
        if not launched:
            return
        # collectives of one process group run in issue order on RCCL's stream, so the buffer written two gathers ago is free
        # again without the compute stream ever waiting for a gather; torch keeps `f` alive (record_stream) until it was read
        inflight['works'].append((dist.all_gather_into_tensor(gathered[inflight['n'] % 2], f, async_op=True), f))
        inflight['n'] += 1
        if len(inflight['works']) > 4:  # host-side bookkeeping only: the oldest ones finished steps ago
            inflight['works'].pop(0)[0].wait()

    def drain():
        for work, _ in inflight['works']:
            work.wait()
        inflight['works'].clear()

    sharded_build = launched and world > 1 and a.build == 'sharded'

    def build_tables():
        if sharded_build:
            return ssa.dist.sharded_build_hash_tables(eh, N_NODES, ei)
        return eh.build_hash_tables(N_NODES, ei)

    # SURVEY 8(d) asks for T_build and the query rate beside the combined figure: measured on a few EXTRA steps after the timed
    # region (three event records cost a step about 2 %, so the timed steps carry none)
    phase_marks = []
    stream = torch.cuda.current_stream(dev)

    def step_build_query(mark=False):
        if mark:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
        table, cards = build_tables()
        if mark:
            e1.record(stream)
        f = eh.get_subgraph_features(links, table, cards)
        if mark:
            e2.record(stream)
            phase_marks.append((e0, e1, e2))
        gather(f)
        return f

    elph_state = {}

    def step_elph():
        """what reference models/elph.py:186-213 + runners/train.py:204 execute per training step"""
        loops = torch.arange(N_NODES, device=dev).repeat(2, 1)
        hash_edge_index = torch.cat([ei, loops], dim=1)                      # add_self_loops
        if 'mh0' not in elph_state:                                          # init once (elph.py:189-192)
            elph_state['mh0'], elph_state['hll0'] = eh.initialise_minhash(N_NODES), eh.initialise_hll(N_NODES)
        table = {0: {'minhash': elph_state['mh0'], 'hll': elph_state['hll0']}}
        cards = torch.zeros((N_NODES, H), device=dev)
        for k in range(1, H + 1):
            table[k] = {'hll': eh.hll_prop(table[k - 1]['hll'], hash_edge_index),
                        'minhash': eh.minhash_prop(table[k - 1]['minhash'], hash_edge_index)}
            cards[:, k - 1] = eh.hll_count(table[k]['hll'])
        f = eh.get_subgraph_features(links, table, cards)
        gather(f)
        return f

    def step_buddy():
        """one build, then --buddy-batches batches of B pairs; a 'step' is one batch incl. its share of the build"""
        table, cards = build_tables()
        for _ in range(a.buddy_batches):
            f = eh.get_subgraph_features(links, table, cards)
            gather(f)
        return f

    step = {'build_query': step_build_query, 'elph': step_elph, 'buddy': step_buddy}[a.api]
    pairs_per_step = BATCH * (a.buddy_batches if a.api == 'buddy' else 1)

    def fence():
        drain()
        torch.cuda.synchronize(dev)
        if launched:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        feats = step()
    timer = KernelTimer(None if a.time_all_kernels else set())
    hashing.KERNEL_TIMER = timer
    lib = ssa._native.lib()
    lib.ss_profile_enable(1 << ssa._native.PROF_MINHASH_HOP)  # HIP events around every launch of the dominant kernel (MinHash table hop), on its stream
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        feats = step()
    fence()
    elapsed = time.perf_counter() - t0
    hashing.KERNEL_TIMER = None
    from ctypes import byref, c_float, c_int32
    dom_ms, dom_n = c_float(), c_int32()
    lib.ss_profile_read(ssa._native.PROF_MINHASH_HOP, byref(dom_ms), byref(dom_n))  # the launches of the timed region only
    lib.ss_profile_enable(0)
    if a.api == 'build_query':
        for _ in range(5):
            step_build_query(mark=True)
        fence()
    if launched:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    prop_ms, prop_n = timer.mean_ms('propagate')
    prop_call_ms = prop_ms
    pair_ms, pair_n = timer.mean_ms('pair_features')
    csr_ms, _ = timer.mean_ms('csr_build')
    first_ms, _ = timer.mean_ms('first_hop')
    e_prime = 2 * E_UND + N_NODES
    prop_bytes = (e_prime + N_NODES) * ROW_BYTES + 4 * e_prime + 8 * (N_NODES + 1) + 4 * N_NODES
    roof_kernel = 'ss::propagate_kernel<128,256> (two-sketch launch)'
    if dom_n.value:  # the library launches the MinHash and HLL hops separately: the MinHash table hop is the dominant kernel
        prop_ms, prop_n = dom_ms.value, dom_n.value
        prop_bytes = (e_prime + N_NODES) * 4 * P + 4 * e_prime + 8 * (N_NODES + 1)
        roof_kernel = 'ss::propagate_kernel<128,256> (MinHash table hop: (E\'+N)*4P + 4E\' + 8(N+1) bytes)'
        if sharded_build:  # each launch covers this rank's N/G destination rows and (on the uniform graph) E'/G in-edges
            prop_bytes //= world
            roof_kernel += f' / {world} ranks (row-sharded build)'
    pair_bytes = BATCH * (2 * H * ROW_BYTES + 16 + 8 * H + 4 * H * (H + 2))
    traffic = None
    pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(pmc_path):
        with open(pmc_path) as fh:
            traffic = json.load(fh).get('propagate_kernel_hbm_bytes_per_launch')

    out = {
        'metric': 'edge-pairs/sec subgraph-feature extraction (build+query)',
        'value': world * pairs_per_step * a.steps / elapsed, 'unit': 'pairs/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'u32/u8 sketches, f32 estimator', 'data': 'synthetic',
        'config': {'workload': f'ogbl-{a.config}-like synthetic {a.graph} graph' + (' (BASELINE configs[1])' if a.config == 'collab' else '') +
                               ', BUDDY/ELPH hot path: step = build_hash_tables + get_subgraph_features, nothing cached across steps' +
                               ('' if a.api == 'build_query' else f' [api mode: {a.api}]'),
                   'num_nodes': N_NODES, 'directed_edges': 2 * E_UND, 'max_hash_hops': H, 'minhash_num_perm': P, 'hll_p': HLL_P,
                   'pairs_per_step_per_gpu': pairs_per_step, 'global_pairs_per_step': world * pairs_per_step,
                   'parallelism': f'edge-batch sharded x{world}, all_gather of features; sketch table ' +
                                  (f'built row-sharded x{world} with an in-place all_gather per hop and sketch' if sharded_build
                                   else 'replicated (every rank builds it)'),
                   'hll_tables': eh.hll_tables.provenance},
        'roofline': {'kernel': roof_kernel + ('' if H > 1 else ' (not launched at h=1)') +
                               (' [elph api mode launches it per sketch: the bytes model below does not apply]' if a.api == 'elph' else ''), 'bound': 'hbm', 'achieved': prop_bytes / (prop_ms * 1e-3) / 1e9 if prop_ms else None,
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': prop_bytes / (prop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if prop_ms else None, 'traffic': traffic,
                     'algorithmic_bytes_per_launch': prop_bytes, 'mean_launch_ms': prop_ms, 'launches_timed': prop_n}
    }
    if phase_marks:
        build_ms = sum(s0.elapsed_time(s1) for s0, s1, _ in phase_marks) / len(phase_marks)
        query_ms = sum(s1.elapsed_time(s2) for _, s1, s2 in phase_marks) / len(phase_marks)
        out['breakdown'] = {'build_ms': build_ms, 'query_ms': query_ms,
                            'build_directed_edges_per_s': H * (2 * E_UND + N_NODES) / (build_ms * 1e-3),  # h * E' / T_build
                            'query_pairs_per_s': BATCH / (query_ms * 1e-3), 'scope': 'this rank, HIP events on the launch stream'}
    if a.time_all_kernels:  # host-side HIP-event spans around every library call (perturbs the step by ~7 %)
        out['kernels'] = {'propagate_call_ms': prop_call_ms, 'first_hop_call_ms': first_ms, 'pair_features_ms': pair_ms,
                          'csr_build_ms': csr_ms, 'pair_features_algorithmic_bytes': pair_bytes,
                          'pair_features_GBps': pair_bytes / (pair_ms * 1e-3) / 1e9 if pair_ms else None,
                          'query_only_pairs_per_s': BATCH / (pair_ms * 1e-3) if pair_ms else None}
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.config in ('collab', 'cora'):
        base, ofeat = cpu_baseline(ei_np, links_np)
        out['cpu_baseline'] = base
        diff = float(np.abs(feats.cpu().numpy() - ofeat).max())
        out['cpu_baseline']['max_abs_feature_diff_vs_gpu'] = diff
        out['speedup_vs_cpu_baseline'] = out['value'] / base['value']
        if a.config == 'collab':
            out['cpu_baseline_reference_style'] = cpu_baseline_reference_style(ei_np, links_np)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if launched:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
