#!/usr/bin/env python3
"""Where a prover-sized MSM's time goes, host included: wall per call, device time (events) and the host fold (hm_get_stats),
one call at a time and as a phase of 16 through the batch call.  Development aid."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_batch
from halo2_experiments_amd.replay import _rand_fr
dev = torch.device("cuda", 0)
lib = _lib.load()
def stats():
    st = _lib.Stats()
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    return st
for k in (9, 11, 13, 15):
    n = 1 << k
    hd = h.register_bases(h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR))
    cols = [_rand_fr(n, 10 + i, dev) for i in range(16)]
    for _ in range(3):
        h.best_multiexp(cols[0], hd); best_multiexp_batch(cols, hd)
    torch.cuda.synchronize()
    _lib.check(lib.hm_reset_stats())
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        h.best_multiexp(cols[0], hd)
    wall = (time.perf_counter() - t0) / reps * 1e3
    st = stats()
    print(f"k={k} single: wall {wall:.3f} ms, device {st.msm_device_us / st.msm_calls / 1e3:.3f} ms, host fold {st.msm_host_us / st.msm_calls / 1e3:.3f} ms, windows {h.msm_stats()['windows']} x {h.msm_stats()['window_bits']} bits", flush=True)
    _lib.check(lib.hm_reset_stats())
    t0 = time.perf_counter()
    for _ in range(reps):
        best_multiexp_batch(cols, hd)
    wall = (time.perf_counter() - t0) / reps * 1e3
    st = stats()
    print(f"k={k} phase of 16: wall {wall:.3f} ms, host fold per MSM {st.msm_host_us / st.msm_calls / 1e3:.3f} ms (x16 = {st.msm_host_us / reps / 1e3:.3f} ms of host time per phase)", flush=True)
    h.release_bases(hd)
