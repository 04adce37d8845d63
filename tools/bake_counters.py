#!/usr/bin/env python3
"""Bake the per-launch counter figures bench.py quotes (SQ_INSTS_VALU of the two hot kernels, PMC HBM traffic) into one
JSON file TOGETHER WITH a hash of the kernel sources they were measured on, so that the bench line can say "stale": true
when a kernel has changed since.

    python tools/bake_counters.py --sq profiles/r03_sq_counters.txt --traffic profiles/r03_pmc_traffic.json \
        --out profiles/r03_baked_counters.json

--sq: output of tools/pmc_sq.sh (one "kernel {counter: value}" line per kernel); --traffic: output of tools/pmc_traffic.sh.
The hash covers the files whose text decides the instruction stream of the kernel (field layer, curve layer, chain /
butterfly code), not the host-side planning code around it."""
import argparse
import ast
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-experiments_amd", "csrc")
SOURCES = {
    "k3": ["ff29.h", "g1.h", "msm_dev.h", "bn256_constants.inc"],
    "ntt": ["ff29.h", "ntt.hip", "bn256_constants.inc"],
}


def sources_sha256(kind: str) -> str:
    h = hashlib.sha256()
    for name in SOURCES[kind]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sq", required=True)
    ap.add_argument("--traffic", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--k3-pairs", type=int, default=15 * (1 << 24), help="(point, bucket) pairs of the profiled K3 launch (the bench line's `pairs`)")
    ap.add_argument("--ntt-elements", type=int, default=1 << 24)
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    sq = {}
    for line in open(args.sq):
        if line.startswith("hm::") and "{" in line:
            name, d = line.split(" {", 1)
            sq[name] = ast.literal_eval("{" + d.strip())
    traffic = json.load(open(args.traffic))
    ntt_name = next((k for k in sq if k.startswith("hm::ntt_pass_kernel<11")), None)        # "<11>" in round 2, "<11, false>" since
    ntt_traffic = next((v for k, v in traffic.items() if k.startswith("hm::ntt_pass_kernel<11")), None)
    out = {
        "note": args.note,
        "k3": {"kernel": "hm::msm_accumulate_kernel", "sq_insts_valu_per_launch": sq["hm::msm_accumulate_kernel"]["SQ_INSTS_VALU"],
               "pairs_per_launch": args.k3_pairs, "grbm_gui_active": sq["hm::msm_accumulate_kernel"].get("GRBM_GUI_ACTIVE"),
               "from": os.path.relpath(args.sq, ROOT), "sources": SOURCES["k3"], "sources_sha256": sources_sha256("k3")},
        "ntt": {"kernel": ntt_name, "sq_insts_valu_per_launch": sq[ntt_name]["SQ_INSTS_VALU"],
                "elements_per_launch": args.ntt_elements, "from": os.path.relpath(args.sq, ROOT), "sources": SOURCES["ntt"],
                "sources_sha256": sources_sha256("ntt")},
        "traffic": {"from": os.path.relpath(args.traffic, ROOT),
                    "hm::msm_accumulate_kernel": traffic.get("hm::msm_accumulate_kernel"),
                    "hm::ntt_pass_kernel<11>": ntt_traffic},
    }
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
