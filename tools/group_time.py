#!/usr/bin/env python3
"""Cost of a phase of `count` equal-size commitments through hm_msm_batch_bn256_g1_dev, dense and sparse columns: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.arithmetic import G1_GENERATOR
from halo2_experiments_amd.replay import _rand_fr, _sparse_column
_lib.load().hm_msm_set_phase_timing(1)
if os.environ.get("WINDOW"): _lib.check(_lib.load().hm_msm_set_window(int(os.environ["WINDOW"])))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
counts = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 20, 32, 36]
n = 1 << k
dev = torch.device("cuda", 0)
hd = h.register_bases(h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR), precompute=os.environ.get("PRECOMP") == "1")
cols = {"dense": _rand_fr(n, 2, dev), "sparse": _sparse_column(n, 1100, 3, dev)}
if os.environ.get("KIND"): cols = {os.environ["KIND"]: cols[os.environ["KIND"]]}
for name, col in cols.items():
    for c in counts:
        h.best_multiexp_batch([col] * c, hd); h.best_multiexp_batch([col] * c, hd); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); h.best_multiexp_batch([col] * c, hd); ts.append(time.perf_counter() - t)
        st = h.msm_stats()
        print(f"2^{k} {name:6s} count {c:3d}: {min(ts)*1e3:8.3f} ms = {min(ts)*1e3/c:.3f} each (median {sorted(ts)[2]*1e3:.3f}); last chain: dev {st['total_ms']:.3f} dig {st['digits_ms']:.3f} "
              f"sort {st['sort_ms']:.3f} k3 {st['accumulate_kernel_ms']:.3f} red {st['reduce_ms']:.3f}", flush=True)
