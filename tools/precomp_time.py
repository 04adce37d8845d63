#!/usr/bin/env python3
"""Plain vs fixed-base (precomputed table) MSM at 2^k: registration time, phase times, same result: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR
from halo2_experiments_amd.replay import _rand_fr
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << k
dev = torch.device("cuda", 0)
bases = h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR)
s = _rand_fr(n, 2, dev)
hp = h.register_bases(bases, plain=True)
ref = h.best_multiexp(s, hp)
for _ in range(2): h.best_multiexp(s, hp)
t = time.perf_counter()
for _ in range(3): h.best_multiexp(s, hp)
dt = (time.perf_counter() - t) / 3
st = h.msm_stats()
print(f"2^{k} plain      : {dt*1e3:8.3f} ms  c={st['window_bits']} W={st['windows']} dig {st['digits_ms']:.3f} sort {st['sort_ms']:.3f} k3 {st['accumulate_kernel_ms']:.3f} red {st['reduce_ms']:.3f}")
h.release_bases(hp)
torch.cuda.synchronize()
t = time.perf_counter()
hq = h.register_bases(bases, precompute=True)
torch.cuda.synchronize()
treg = time.perf_counter() - t
got = h.best_multiexp(s, hq)
for _ in range(2): h.best_multiexp(s, hq)
t = time.perf_counter()
for _ in range(3): h.best_multiexp(s, hq)
dt = (time.perf_counter() - t) / 3
st = h.msm_stats()
print(f"2^{k} fixed-base : {dt*1e3:8.3f} ms  c={st['window_bits']} W={st['windows']} dig {st['digits_ms']:.3f} sort {st['sort_ms']:.3f} k3 {st['accumulate_kernel_ms']:.3f} red {st['reduce_ms']:.3f}  register {treg*1e3:.1f} ms  same result {bool(np.array_equal(ref, got))}")
