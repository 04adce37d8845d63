#!/usr/bin/env python3
"""The host-pointer (drop-in) entry points with their copies through the library's pinned staging lanes (csrc/xfer.hip, HALO2_MI355X_HOST_COPIES=lanes) and
through the runtime's pageable path (HALO2_MI355X_HOST_COPIES=direct), and with the default policy `auto` (lanes for every range the caller has not registered with
hm_host_register), each mode in its own process: median AND worst of 12 calls per shape -- the worst is the point: on some boxes the runtime's path pays ~9 us per
4 KiB page for every array it has not pinned before (what a prover hands it on every call).  Every "fresh" shape allocates a new output / input array per call, like
the Rust glue; the "registered" rows reuse ONE array the caller has registered (the direct path by the caller's choice).

    python tools/host_copies.py            # both modes, table
    python tools/host_copies.py --child    # (internal) one mode, JSON line
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 12


def child():
    import ctypes
    import numpy as np
    import torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_batch
    from halo2_experiments_amd.domain import EvaluationDomain, fr_words
    from halo2_experiments_amd.replay import run_replay
    dev = torch.device("cuda", 0)
    run_replay("poseidon_k11", device=dev)                      # the state a prover process is in: streams, staging, tables
    out = {}

    def rec(name, fn, reps=REPS):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        out[name] = {"median_ms": ts[len(ts) // 2], "max_ms": ts[-1], "min_ms": ts[0]}

    for k in (17, 18):
        dom = EvaluationDomain(7, k)
        hs = h.random_fr(dom.n, 5, dev).cpu().numpy().view(np.uint64).copy()
        keep = dom.coeff_to_extended(hs)
        rec(f"coeff_to_extended k={k}, fresh output", lambda: dom.coeff_to_extended(hs))
        rec(f"coeff_to_extended k={k}, touched output", lambda: dom.coeff_to_extended(hs, out=keep))
        rec(f"extended_to_coeff k={k}, fresh copy of the input", lambda: dom.extended_to_coeff(keep.copy()))
        w = fr_words(dom.extended_omega)
        rec(f"best_fft 2^{dom.extended_k}, zero-padded fresh array", lambda: h.best_fft(np.concatenate([hs, np.zeros((dom.extended_len() - dom.n, 4), dtype=np.uint64)]), w, dom.extended_k))
        buf = keep.copy()
        rec(f"best_fft 2^{dom.extended_k}, touched array", lambda: h.best_fft(buf, w, dom.extended_k))
    # a long-lived array the caller registers: direct DMA under the default policy
    dom = EvaluationDomain(7, 18)
    reg = np.zeros((dom.extended_len(), 4), dtype=np.uint64)
    reg[:dom.n] = h.random_fr(dom.n, 6, dev).cpu().numpy().view(np.uint64)
    lib = _lib.load()
    if lib.hm_host_register(reg.ctypes.data_as(ctypes.c_void_p), reg.nbytes) == 0:
        rec("best_fft 2^21, an array registered with hm_host_register", lambda: h.best_fft(reg, fr_words(dom.extended_omega), dom.extended_k))
        _lib.check(lib.hm_host_unregister(reg.ctypes.data_as(ctypes.c_void_p)))
    n = 1 << 18
    bases = h.g1_fixed_base_mul(h.random_fr(n, 7, dev), G1_GENERATOR)
    handle = h.register_bases(bases)
    cols = [h.random_fr(n, 100 + i, dev).cpu().numpy().view(np.uint64).copy() for i in range(19)]
    rec("best_multiexp 2^18 scalars from a host array", lambda: h.best_multiexp(cols[0], handle))
    rec("best_multiexp 2^18, fresh copy of the scalars", lambda: h.best_multiexp(cols[1].copy(), handle))
    rec("a phase of 19 dense commitments from host arrays", lambda: best_multiexp_batch(cols, handle), reps=6)
    rec("... from fresh copies", lambda: best_multiexp_batch([c.copy() for c in cols], handle), reps=6)
    st = _lib.Stats()
    _lib.check(_lib.load().hm_get_stats(ctypes.byref(st)))
    out["_copies"] = {"median_ms": float(st.host_copies_direct), "max_ms": float(st.host_copies_staged), "min_ms": 0.0}     # copies handed to hipMemcpy / through the lanes
    print(json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    res = {}
    modes = [("pinned lanes", {"HALO2_MI355X_HOST_COPIES": "lanes"}), ("runtime pageable path", {"HALO2_MI355X_HOST_COPIES": "direct"}), ("auto (default)", {})]
    for tag, env in modes:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            print(tag, "FAILED", r.stderr[-500:])
            continue
        res[tag] = json.loads(line[-1])
    tags = list(res)
    print(f"{'shape (12 calls; median / worst ms)':58s}" + "".join(f"{t:>30s}" for t in tags))
    for shape in res[tags[0]]:
        print(f"{shape:58s}" + "".join(f"{res[t][shape]['median_ms']:18.2f} /{res[t][shape]['max_ms']:9.2f}" for t in tags))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
