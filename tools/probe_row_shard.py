"""probe: per-rank compute of the row-sharded build measured on ONE GPU: this process plays rank r of G by
launching only rows [r*N/G, (r+1)*N/G) of every hop; the exchange is not measured (needs xGMI) but its size is printed.
usage: python tools/probe_row_shard.py [collab|ppa|citation2] [G]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import subgraph_sketching_amd as ssa
from argparse import Namespace
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'ppa']
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
n, h = cfg['n'], cfg['h']
ei = torch.from_numpy(bench.synthetic_graph(cfg['n'], cfg['e_und'])).to(dev)
eh = ssa.ElphHashes(Namespace(max_hash_hops=h, hll_p=8, minhash_num_perm=128, floor_sf=False, use_zero_one=True))
eh.strict_bounds = False
per = (n + G - 1) // G


class Shard(object):  # no exchange: rows of the other ranks stay uninitialised (timing only)
    def __init__(self, r):
        self.rows, self.padded_rows = (min(r * per, n), min((r + 1) * per, n)), per * G

    def gather(self, full):
        return None

    def wait(self, handle):
        pass


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


full = timed(lambda: eh.build_hash_tables(n, ei))
print(f'N={n} E_dir={2 * cfg["e_und"]} h={h}: unsharded build {full:.3f} ms', flush=True)
for r in (0, G // 2, G - 1):
    t = timed(lambda: eh._build(n, ei, Shard(r)))
    print(f'rank {r} of {G}: CSR (replicated) + own rows of {h} hops {t:.3f} ms', flush=True)
block = per * 768
print(f'exchange per hop: each rank sends its {block / 1e6:.1f} MB block to {G - 1} peers (receives {(G - 1) * block / 1e6:.1f} MB); '
      f'at 7 xGMI links x 153 GB/s that is ~{(G - 1) * block / (7 * 153e9) * 1e3:.2f} ms per hop (twice that if the figure is bidirectional), '
      f'overlapped per sketch')
