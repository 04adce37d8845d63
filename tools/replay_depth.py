#!/usr/bin/env python3
"""create_proof trace replay vs the number of MSMs kept in flight: development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2_experiments_amd.replay import run_replay
for shape in sys.argv[1].split(",") if len(sys.argv) > 1 else ["merkle_sum_tree_k18"]:
    for d in [int(x) for x in os.environ.get("DEPTHS", "1,2,3,4,6,8").split(",")]:
        r = run_replay(shape, include_host_pointer_estimate=False, in_flight=d)
        t = r["device_resident_s"]
        print(f"{shape} in_flight={d}: msm {t['msm']*1e3:7.2f} ms  ntt {t['ntt']*1e3:7.2f} ms  total {t['total']*1e3:7.2f} ms", flush=True)
