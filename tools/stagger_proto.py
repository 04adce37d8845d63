#!/usr/bin/env python3
"""Prototype: one 2^k MSM as two staggered half-MSMs (separate half tables, two streams), the second submitted `delay` after the
first so that its digits + sort run under the first half's accumulation.  Measures whether hiding the sort inside ONE MSM pays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_submit, best_multiexp_wait
from halo2_experiments_amd.replay import _rand_fr
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << k
dev = torch.device("cuda", 0)
bases = h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR)
s = _rand_fr(n, 2, dev)
full = h.register_bases(bases)
for _ in range(2): ref = h.best_multiexp(s, full)
ts = []
for _ in range(4):
    t = time.perf_counter(); h.best_multiexp(s, full); ts.append(time.perf_counter() - t)
print(f"one MSM: {min(ts)*1e3:.3f} ms", flush=True)
h.release_bases(full)
half = n // 2
ha, hb = h.register_bases(bases[:half].contiguous()), h.register_bases(bases[half:].contiguous())
sa, sb = s[:half].contiguous(), s[half:].contiguous()
st1, st2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for delay_us in (0, 500, 1000, 1500, 2000, 2500, 3000):
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st1): ta = best_multiexp_submit(sa, ha)
        while (time.perf_counter() - t0) * 1e6 < delay_us: pass
        with torch.cuda.stream(st2): tb = best_multiexp_submit(sb, hb)
        ra = best_multiexp_wait(ta); rb = best_multiexp_wait(tb)
        dt = time.perf_counter() - t0
        best = min(best, dt)
    print(f"two halves, second submitted {delay_us:5d} us later: {best*1e3:.3f} ms", flush=True)
