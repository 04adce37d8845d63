import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_batch
from halo2_experiments_amd.replay import _sparse_column, _rand_fr
dev = torch.device("cuda", 0)
n = 1 << 18
bases = h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR)
hd = h.register_bases(bases)
dense = [_rand_fr(n, 10 + i, dev) for i in range(2)]
sparse = [_sparse_column(n, 1100, 20 + i, dev) for i in range(2)]
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for cnt in (1, 2, 3, 5, 8, 19):
    print("dense x", cnt, round(t(lambda: best_multiexp_batch([dense[i & 1] for i in range(cnt)], hd)), 3), "ms")
for cnt in (1, 2, 5, 9, 36):
    print("sparse x", cnt, round(t(lambda: best_multiexp_batch([sparse[i & 1] for i in range(cnt)], hd)), 3), "ms")
print("single dense", round(t(lambda: h.best_multiexp(dense[0], hd)), 3), "single sparse", round(t(lambda: h.best_multiexp(sparse[0], hd)), 3))
st = h.msm_stats()
print(st)
