#!/usr/bin/env python3
"""MSM wall time vs size (and optionally vs window size): development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.arithmetic import G1_GENERATOR

def rand_fr(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)

_lib.load().hm_msm_set_phase_timing(1)      # per-phase events also for the five-launch plan
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12, 14, 16, 18, 20, 22, 24]
windows = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for k in sizes:
    n = 1 << k
    pre = os.environ.get("PRECOMP", "0") == "1"
    t0 = time.perf_counter()
    hd = h.register_bases(h.g1_fixed_base_mul(rand_fr(n, 11), G1_GENERATOR), precompute=pre)
    torch.cuda.synchronize(); print(f"  register(precompute={pre}) {(time.perf_counter()-t0)*1e3:.1f} ms")
    s = rand_fr(n, 12)
    for c in windows:
        _lib.check(_lib.load().hm_msm_set_window(c))
        h.best_multiexp(s, hd); torch.cuda.synchronize()
        reps = 5 if k < 22 else 3
        t = time.perf_counter()
        for _ in range(reps): h.best_multiexp(s, hd)
        dt = (time.perf_counter() - t) / reps
        st = h.msm_stats()
        print(f"2^{k} c={st['window_bits']:2d} W={st['windows']:2d}: {dt*1e3:7.3f} ms {n/dt/1e6:7.1f} Mpts/s | dig {st['digits_ms']:.3f} sort {st['sort_ms']:.3f} "
              f"acc {st['accumulate_ms']:.3f} (k3 {st['accumulate_kernel_ms']:.3f}) red {st['reduce_ms']:.3f} dev {st['total_ms']:.3f} tasks {st['tasks']}")
        sys.stdout.flush()
    _lib.load().hm_msm_set_window(0)
    h.release_bases(hd)
