#!/usr/bin/env python3
"""create_proof trace replays (bench.py's side measurement) on their own: development aid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2_experiments_amd.replay import run_replay
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["poseidon_k11", "merkle_v3_k17", "merkle_sum_tree_k18"]
for name in names:
    r = run_replay(name, device=torch.device("cuda", 0), include_host_pointer_estimate=False)
    print(name, json.dumps(r["device_resident_s"]), flush=True)
