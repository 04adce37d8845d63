#!/usr/bin/env python3
"""The commitment phases of the k = 18 replay, timed one by one, in a chosen order: development aid.
    python tools/phase_mix.py 18 SDd   (S = 36 sparse on g_lagrange, D = 12 dense on g_lagrange, d = 7 dense on g)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import best_multiexp_batch
from halo2_experiments_amd.kzg import ParamsKZG
from halo2_experiments_amd.replay import REPLAY_S, _rand_fr, _sparse_column
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
order = sys.argv[2] if len(sys.argv) > 2 else "SDd"
n = 1 << k
dev = torch.device("cuda", 0)
params = ParamsKZG.setup(k, REPLAY_S, device=dev)
g_h, gl_h = params.g_handle, params.g_lagrange_handle
dense = [_rand_fr(n, 100 + i, dev) for i in range(2)]
sparse = [_sparse_column(n, 1100, 200 + i, dev) for i in range(2)]
kinds = {"S": ("sparse x36 gl", [sparse[i & 1] for i in range(36)], gl_h), "D": ("dense x12 gl", [dense[i & 1] for i in range(12)], gl_h),
         "d": ("dense x7 g", [dense[i & 1] for i in range(7)], g_h), "s": ("sparse x36 g", [sparse[i & 1] for i in range(36)], g_h)}
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = []
    for ch in order:
        name, cols, hd = kinds[ch]
        t = time.perf_counter()
        best_multiexp_batch(cols, hd)
        out.append((name, (time.perf_counter() - t) * 1e3))
    tot = (time.perf_counter() - t0) * 1e3
    print(f"round {rep}: total {tot:.3f} ms | " + " | ".join(f"{nm}: {ms:.3f}" for nm, ms in out), flush=True)
params.release()
