import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.arithmetic import G1_GENERATOR
from halo2_experiments_amd.replay import _rand_fr
dev = torch.device("cuda", 0)
lib = _lib.load()
for k in (16, 17, 18, 19, 20):
    n = 1 << k
    bases = h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR)
    for thr in (17, 0):
        lib.hm_set_fixed_base_threshold(thr)
        hd = h.register_bases(bases)
        s = _rand_fr(n, 2, dev)
        h.best_multiexp(s, hd); torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter(); h.best_multiexp(s, hd); ts.append(time.perf_counter() - t0)
        lib.hm_msm_set_phase_timing(1)
        h.best_multiexp(s, hd)
        st = h.msm_stats()
        lib.hm_msm_set_phase_timing(0)
        print(k, "table" if thr else "plain", round(np.median(ts) * 1e3, 3), {a: round(b, 3) if isinstance(b, float) else b for a, b in st.items()}, h.bases_info(hd)["table_windows"], flush=True)
        h.release_bases(hd)
lib.hm_set_fixed_base_threshold(17)
