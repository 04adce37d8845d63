#!/usr/bin/env python3
"""NTT wall time per size / batch, and the EvaluationDomain wrappers of a k = 18 proof: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS, FR_ROOT_OF_UNITY, fr_words

def rand_fr(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for k, batch in [(12, 1), (16, 1), (18, 1), (18, 8), (20, 1), (21, 1), (21, 8), (22, 1), (24, 1), (26, 1)]:
    omega = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS))
    a = rand_fr(batch << k, 5).reshape(batch, 1 << k, 4)
    dom = None
    import ctypes
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    def f():
        _lib.check(_lib.load().hm_ntt_batch_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), batch, _ptr(omega), k, None, None,
                                                          ctypes.c_void_p(_stream_ptr(a))))
    ms = timed(f)
    print(f"ntt 2^{k} x{batch}: {ms:8.4f} ms  {(batch << k) / ms / 1e6:8.1f} Gel/s-ish (M elements / ms)   per stage-element {ms * 1e6 / ((batch << k) * k):.3f} ns")
    del a

dom = EvaluationDomain(7, 18)
polys = rand_fr(8 << 18, 7).reshape(8, 1 << 18, 4)
print("lagrange_to_coeff x8 (k=18):", timed(lambda: dom.lagrange_to_coeff(polys)))
print("coeff_to_extended x8 (k=18 -> 21):", timed(lambda: dom.coeff_to_extended(polys)))
ext = dom.coeff_to_extended(polys)
print("extended_to_coeff x1:", timed(lambda: dom.extended_to_coeff(ext[0])))
print("torch.zeros(8, 2^21, 4) + slice copy:", timed(lambda: torch.zeros((8, 1 << 21, 4), dtype=torch.int64, device="cuda").__setitem__((slice(None), slice(0, 1 << 18)), polys)))
