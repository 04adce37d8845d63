#!/usr/bin/env python3
"""create_proof trace replays (bench.py's side measurement) on their own: development aid.
    python tools/replay_quick.py [names] [cosets | allcosets]   -- the extended-domain steps one coset at a time: the j - 1 cosets that
                                                                 determine the quotient, or all E"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from halo2_experiments_amd.replay import run_replay
args = [a for a in sys.argv[1:] if a not in ("cosets", "allcosets")]
names = args[0].split(",") if args else ["poseidon_k11", "merkle_v3_k17", "merkle_sum_tree_k18"]
for name in names:
    r = run_replay(name, device=torch.device("cuda", 0), include_host_pointer_estimate=False, by_cosets=True if ("cosets" in sys.argv or "allcosets" in sys.argv) else None,
                   min_cosets="allcosets" not in sys.argv)
    print(name, json.dumps(r["device_resident_s"]), flush=True)
