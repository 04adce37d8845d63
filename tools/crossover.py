#!/usr/bin/env python3
"""Where does the host-pointer drop-in start to pay?  The patched best_multiexp / best_fft (rust/halo2_proofs-patch/src/mi355x.rs)
send a call to the GPU from 2^GPU_MIN_LOG_N_* elements and run the CPU body below that.  This measures both sides at 2^8 .. 2^15,
on the box that runs it: the host-pointer C-ABI forms (hm_msm_bn256_g1_jacobian with the same base array on every call, as
create_proof passes params.g_lagrange -- a digest hit; hm_ntt_bn256_fr) against oracle/cpu_ref.c (the C restatement of upstream's
bodies) on the cores the box grants.  Prints one JSON object; the crossover is the smallest size from which the GPU form is
faster at every larger size measured.

    python tools/crossover.py > gpurun_out/crossover.json          (profiles/r04_crossover.json is a copy)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import halo2_experiments_amd as h  # noqa: E402
from halo2_experiments_amd.arithmetic import G1_GENERATOR  # noqa: E402
from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY, fr_words  # noqa: E402
from oracle import cpu_ref  # noqa: E402  (the CPU side of the comparison: a measurement tool, not the product)


def rand_fr(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)


def median_ms(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def crossover(rows, gpu_key, cpu_key):
    best = None
    for r in reversed(rows):
        if r[gpu_key] < r[cpu_key]:
            best = r["log_n"]
        else:
            break
    return best


def main():
    cpu_ref.build()
    threads = cpu_ref.default_threads()
    msm_rows, ntt_rows = [], []
    for k in range(8, 16):
        n = 1 << k
        s = rand_fr(n, 100 + k).cpu().numpy().view(np.uint64).copy()
        b = h.g1_fixed_base_mul(rand_fr(n, 200 + k), G1_GENERATOR).cpu().numpy().view(np.uint64).copy()
        reps = 30 if k <= 12 else 10
        gpu = median_ms(lambda: h.best_multiexp(s, b), reps)
        cpu = median_ms(lambda: cpu_ref.best_multiexp(s, b, threads), reps)
        cpu1 = median_ms(lambda: cpu_ref.best_multiexp(s, b, 1), max(3, reps // 3))
        ok = bool(np.array_equal(h.best_multiexp(s, b)[:8], cpu_ref.g1_to_affine(cpu_ref.best_multiexp(s, b, threads))[0]))
        msm_rows.append({"log_n": k, "gpu_host_pointer_ms": gpu, "cpu_ms": cpu, "cpu_1_thread_ms": cpu1, "same_result": ok})
        a = rand_fr(n, 300 + k).cpu().numpy().view(np.uint64).copy()
        w = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS))

        def gpu_ntt():
            x = a.copy()
            h.best_fft(x, w, k)
            return x
        copy_ms = median_ms(lambda: a.copy(), reps)
        gpu = median_ms(gpu_ntt, reps) - copy_ms
        cpu = median_ms(lambda: cpu_ref.best_fft(a, w, k, threads), reps) - copy_ms          # (the wrapper copies its input too)
        cpu1 = median_ms(lambda: cpu_ref.best_fft(a, w, k, 1), reps) - copy_ms
        ok = bool(np.array_equal(gpu_ntt(), cpu_ref.best_fft(a, w, k, threads)))
        ntt_rows.append({"log_n": k, "gpu_host_pointer_ms": gpu, "cpu_ms": cpu, "cpu_1_thread_ms": cpu1, "same_result": ok})
    out = {"cpu": {"threads": threads, "cpus_visible": os.cpu_count(), "kind": "port (oracle/cpu_ref.c, C restatement of halo2_proofs v2023_02_02)"},
           "msm": msm_rows, "ntt": ntt_rows,
           "crossover_log_n": {"msm_vs_all_threads": crossover(msm_rows, "gpu_host_pointer_ms", "cpu_ms"),
                               "msm_vs_one_thread": crossover(msm_rows, "gpu_host_pointer_ms", "cpu_1_thread_ms"),
                               "ntt_vs_all_threads": crossover(ntt_rows, "gpu_host_pointer_ms", "cpu_ms"),
                               "ntt_vs_one_thread": crossover(ntt_rows, "gpu_host_pointer_ms", "cpu_1_thread_ms")},
           "note": "medians of 10-30 calls, ms; GPU = the host-pointer drop-in forms (uploads and the result copy included; MSM bases "
                   "unchanged between calls: digest hit, as in create_proof); upstream's log_n <= log_threads NTT branch and its "
                   "per-thread MSM chunks are what the CPU side runs"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
