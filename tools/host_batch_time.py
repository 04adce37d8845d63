#!/usr/bin/env python3
"""hm_msm_batch_bn256_g1_h on host arrays: time per call at a prover size: development aid.  python tools/host_batch_time.py 11 6,10,16"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_batch
from halo2_experiments_amd.replay import _rand_fr, _sparse_column
k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
counts = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6, 10, 16]
n = 1 << k
dev = torch.device("cuda", 0)
hd = h.register_bases(h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR))
cols = {"dense": _rand_fr(n, 2, dev).cpu().numpy().view(np.uint64), "sparse": _sparse_column(n, min(1100, n // 4), 3, dev).cpu().numpy().view(np.uint64)}
for name, col in cols.items():
    for c in counts:
        best_multiexp_batch([col] * c, hd); best_multiexp_batch([col] * c, hd)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); best_multiexp_batch([col] * c, hd); ts.append(time.perf_counter() - t)
        print(f"2^{k} {name:6s} host arrays x{c:3d}: min {min(ts)*1e3:8.3f} ms  median {sorted(ts)[3]*1e3:8.3f} ms  max {max(ts)*1e3:8.3f}", flush=True)
