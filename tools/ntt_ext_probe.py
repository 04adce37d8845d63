#!/usr/bin/env python3
"""Where a slow hm_ntt_bn256_fr call spends its time: 20 host-pointer transforms of one 2^21 array (64 MiB each way) in the state a
replay leaves the process in, each with its h2d / device / d2h microseconds from hm_get_stats.  Development aid.

    python tools/ntt_ext_probe.py [replay-first: 0|1] [array: zeros|full]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import halo2_experiments_amd as h  # noqa: E402
from halo2_experiments_amd import _lib  # noqa: E402
from halo2_experiments_amd.domain import EvaluationDomain, fr_words  # noqa: E402

replay_first = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
kind = sys.argv[2] if len(sys.argv) > 2 else "zeros"
dev = torch.device("cuda", 0)
if replay_first:
    from halo2_experiments_amd.replay import run_replay
    run_replay("merkle_sum_tree_k18", device=dev, include_host_pointer_estimate=False)
dom = EvaluationDomain(7, 18)
hs = h.random_fr(dom.n, 5, dev).cpu().numpy().view(np.uint64).copy()
if kind == "zeros":
    a = np.zeros((dom.extended_len(), 4), dtype=np.uint64)
    a[:dom.n] = hs
else:
    a = h.random_fr(dom.extended_len(), 6, dev).cpu().numpy().view(np.uint64).copy()
w = fr_words(dom.extended_omega)
lib = _lib.load()


def stats():
    st = _lib.Stats()
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    return st.ntt_h2d_us, st.ntt_device_us, st.ntt_d2h_us


print(f"replay_first={replay_first} array={kind}")
for i in range(20):
    s0 = stats()
    t0 = time.perf_counter()
    h.best_fft(a, w, dom.extended_k)
    dt = (time.perf_counter() - t0) * 1e3
    s1 = stats()
    print(f"call {i:2d}: {dt:7.2f} ms   h2d {s1[0] - s0[0]:8.0f} us   device {s1[1] - s0[1]:8.0f} us   d2h {s1[2] - s0[2]:8.0f} us")
