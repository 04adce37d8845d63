#!/usr/bin/env python3
"""Host time of hm_msm_submit_dev (the launches of one MSM, nothing awaited) at 2^k, phase events on and off: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_submit, best_multiexp_wait
from halo2_experiments_amd.replay import _rand_fr
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
dev = torch.device("cuda", 0)
hd = h.register_bases(h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR))
col = _rand_fr(n, 2, dev)
for timing in (1, 0):
    _lib.load().hm_msm_set_phase_timing(timing)
    for rnd in range(3):
        torch.cuda.synchronize()
        ts = []
        t0 = time.perf_counter()
        tickets = []
        for _ in range(8):
            t = time.perf_counter(); tickets.append(best_multiexp_submit(col, hd)); ts.append(time.perf_counter() - t)
        t1 = time.perf_counter()
        for t in tickets: best_multiexp_wait(t)
        t2 = time.perf_counter()
        print(f"2^{k} phase events {timing}: submit x8 {1e3*(t1-t0):.3f} ms ({' '.join(f'{1e6*x:.0f}' for x in ts)} us), waits {1e3*(t2-t1):.3f} ms", flush=True)
