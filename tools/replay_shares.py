import json, sys, os
sys.path.insert(0, os.getcwd())
import torch
from halo2_experiments_amd.replay import run_replay
dev = torch.device("cuda", 0)
for name in ("merkle_sum_tree_k18", "merkle_v3_k17"):
    for w in (2, 4, 8):
        for r in (0, w - 1):
            rep = run_replay(name, device=dev, include_host_pointer_estimate=False, share_of=(r, w))
            print(name, (r, w), {k: round(v * 1e3, 2) for k, v in rep["device_resident_s"].items()}, rep["verified"]["commitments_checked"], flush=True)
