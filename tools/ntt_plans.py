#!/usr/bin/env python3
"""A/B of NTT plans at the prover's extended-domain size: the same transforms through builds of the library that
differ in ntt.hip's plan knobs (tools/ab_ntt.sh), each in its own process (HALO2_MI355X_LIB), results compared bit for bit with the
default build's, times by HIP events.

    python tools/ntt_plans.py                       # every gpurun_ab/libhalo2_mi355x_*.so against the tree's library
    python tools/ntt_plans.py --child               # (internal) one build: prints a JSON line
    python tools/ntt_plans.py --pmc-child K BATCH   # (internal) the workload of one `rocprofv3 --pmc SQ_INSTS_VALU` pass
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("ntt", 21, 1), ("ntt", 21, 48), ("c2e", 18, 48), ("c2e", 17, 10), ("ntt", 20, 10), ("ntt", 24, 1), ("ntt", 18, 48)]


def child():
    import ctypes
    import torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS, FR_ROOT_OF_UNITY, fr_words
    lib = _lib.load()

    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    for kind, k, batch in SHAPES:
        if kind == "ntt":
            omega = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS))
            src = h.random_fr(batch << k, 11 + k, "cuda")
            a = src.clone()
            f = lambda: _lib.check(lib.hm_ntt_batch_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), batch, _ptr(omega), k, None, None,
                                                                 ctypes.c_void_p(_stream_ptr(a))))
            f(); torch.cuda.synchronize()
            digest = hashlib.sha256(a.cpu().numpy().tobytes()).hexdigest()[:16]
            ms = timed(f)
            elements = batch << k
        else:
            dom = EvaluationDomain(7, k)
            polys = h.random_fr(batch << k, 13 + k, "cuda").reshape(batch, 1 << k, 4)
            ext = dom.coeff_to_extended(polys, internal=True)
            torch.cuda.synchronize()
            digest = hashlib.sha256(ext.cpu().numpy().tobytes()).hexdigest()[:16]
            del ext
            ms = timed(lambda: dom.coeff_to_extended(polys, internal=True))
            elements = batch << dom.extended_k
        out[f"{kind}_{k}_x{batch}"] = {"ms": ms, "ns_per_element": ms * 1e6 / elements, "sha": digest}
        torch.cuda.empty_cache()
    print(json.dumps(out))


def pmc_child(k, batch):
    import ctypes
    import torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY, fr_words
    lib = _lib.load()
    omega = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS))
    a = h.random_fr(batch << k, 3, "cuda")
    for _ in range(2):
        _lib.check(lib.hm_ntt_batch_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), batch, _ptr(omega), k, None, None, ctypes.c_void_p(_stream_ptr(a))))
    torch.cuda.synchronize()


def main():
    if "--child" in sys.argv:
        return child()
    if "--pmc-child" in sys.argv:
        i = sys.argv.index("--pmc-child")
        return pmc_child(int(sys.argv[i + 1]), int(sys.argv[i + 2]))
    builds = [("default", None)] + [(os.path.basename(p)[len("libhalo2_mi355x_"):-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_ab", "libhalo2_mi355x_*.so")))]
    results = {}
    for tag, path in builds:
        env = dict(os.environ)
        if path:
            env["HALO2_MI355X_LIB"] = path
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"{tag}: FAILED rc={r.returncode} {r.stderr[-400:]}")
            continue
        results[tag] = json.loads(line[-1])
    base = results.get("default", {})
    print(f"{'shape':16s}" + "".join(f"{t:>22s}" for t in results))
    for shape in base:
        row = f"{shape:16s}"
        for t, res in results.items():
            same = "" if res[shape]["sha"] == base[shape]["sha"] else " !=RESULT"
            row += f"{res[shape]['ms']:12.4f} ms {res[shape]['ms'] / base[shape]['ms']:5.3f}x{same}"
        print(row)
    print(json.dumps(results))


if __name__ == "__main__":
    main()
