#!/usr/bin/env python3
"""Turn one tools/profile_round.sh session into the round's committed evidence:

    python tools/bake_counters.py --tag r04_a [--note "..."]

reads gpurun_out/evidence/<tag>/ (written by `bash tools/profile_round.sh <tag>` on the GPU box, merged back by gpurun),
CHECKS it (tools/evidence.py:consistency_problems -- the K3 average of the kernel trace within 5 % of the
roofline.kernel_ms of the line printed under the profiler and below ms_per_step, the trace names the table-build kernel
iff the line says fixed-base table, the sources hashed on the box are the sources of this tree) and only then copies

    kernel_stats.csv -> profiles/<tag>_bench_kernel_stats.csv      bench_under_rocprof.json -> profiles/<tag>_bench_under_rocprof.json
    bench.json       -> profiles/<tag>_bench.json                  pmc_traffic.json         -> profiles/<tag>_pmc_traffic.json
    sq_counters.txt  -> profiles/<tag>_sq_counters.txt

and writes profiles/<tag>_baked_counters.json: the per-launch counter figures bench.py quotes (SQ_INSTS_VALU of the two
hot kernels, PMC HBM traffic), each with a SHA-256 of the kernel sources it was measured on, so that the bench line can say
"stale": true when a kernel has changed since.  bench.py reads the baked file with the greatest tag.  A session that fails
a check is refused: nothing is copied."""
import argparse
import ast
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import evidence  # noqa: E402

ROOT = evidence.ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True, help="rNN_x: names both gpurun_out/evidence/<tag>/ and the profiles/<tag>_* files")
    ap.add_argument("--note", default="")
    ap.add_argument("--ntt-elements", type=int, default=1 << 24)
    args = ap.parse_args()
    e = os.path.join(ROOT, "gpurun_out", "evidence", args.tag)
    need = ["kernel_stats.csv", "bench_under_rocprof.json", "pmc_traffic.json", "sq_counters.txt", "manifest.json"]
    missing = [n for n in need if not os.path.exists(os.path.join(e, n))]
    if missing:
        raise SystemExit(f"bake_counters: {e} lacks {missing}")
    problems = evidence.consistency_problems(os.path.join(e, "kernel_stats.csv"), os.path.join(e, "bench_under_rocprof.json"),
                                             os.path.join(e, "bench.json"))
    man = json.load(open(os.path.join(e, "manifest.json")))
    now = evidence.manifest()
    for kind in ("k3", "ntt"):
        if man[f"{kind}_sources_sha256"] != now[f"{kind}_sources_sha256"]:
            problems.append(f"the {kind} kernel sources of this tree differ from the ones the session measured")
    if problems:
        for p in problems:
            print("REFUSED:", p)
        raise SystemExit(1)
    sq = {}
    for line in open(os.path.join(e, "sq_counters.txt")):
        if line.startswith("hm::") and "{" in line:
            name, d = line.split(" {", 1)
            sq[name] = ast.literal_eval("{" + d.strip())
    traffic = json.load(open(os.path.join(e, "pmc_traffic.json")))
    under = json.load(open(os.path.join(e, "bench_under_rocprof.json")))
    under_full = under                      # round 5 on: the stdout line is compact, the full record sits beside it
    if os.path.exists(os.path.join(e, "bench_under_rocprof_extras.json")):
        under_full = json.load(open(os.path.join(e, "bench_under_rocprof_extras.json")))
    ntt_name = next(k for k in sq if k.startswith("hm::ntt_pass_kernel<11"))
    ntt_traffic = next(v for k, v in traffic.items() if k.startswith("hm::ntt_pass_kernel<11"))
    prof = os.path.join(ROOT, "profiles")
    copies = {"kernel_stats.csv": f"{args.tag}_bench_kernel_stats.csv", "bench_under_rocprof.json": f"{args.tag}_bench_under_rocprof.json",
              "bench.json": f"{args.tag}_bench.json", "pmc_traffic.json": f"{args.tag}_pmc_traffic.json",
              "sq_counters.txt": f"{args.tag}_sq_counters.txt", "bench_extras.json": f"{args.tag}_bench_extras.json"}
    for src, dst in copies.items():
        if os.path.exists(os.path.join(e, src)):
            shutil.copyfile(os.path.join(e, src), os.path.join(prof, dst))
    rows = evidence.kernel_rows(os.path.join(e, "kernel_stats.csv"))
    out = {
        "note": args.note, "tag": args.tag,
        "kernel_stats": {"file": f"profiles/{copies['kernel_stats.csv']}", "sha256": evidence.sha256_file(os.path.join(e, "kernel_stats.csv")),
                         "k3_calls": rows[evidence.K3][0], "k3_average_ms": rows[evidence.K3][1] / 1e6,
                         "kernel_ms_of_the_line_under_the_profiler": under["roofline"]["kernel_ms"]},
        "k3": {"kernel": evidence.K3, "sq_insts_valu_per_launch": sq[evidence.K3]["SQ_INSTS_VALU"],
               "pairs_per_launch": under_full["msm_phase_ms"]["pairs"], "grbm_gui_active": sq[evidence.K3].get("GRBM_GUI_ACTIVE"),
               "from": f"profiles/{copies['sq_counters.txt']}", "sources": evidence.SOURCES["k3"], "sources_sha256": now["k3_sources_sha256"]},
        "ntt": {"kernel": ntt_name, "sq_insts_valu_per_launch": sq[ntt_name]["SQ_INSTS_VALU"], "elements_per_launch": args.ntt_elements,
                "from": f"profiles/{copies['sq_counters.txt']}", "sources": evidence.SOURCES["ntt"], "sources_sha256": now["ntt_sources_sha256"]},
        "traffic": {"from": f"profiles/{copies['pmc_traffic.json']}", evidence.K3: traffic.get(evidence.K3), "hm::ntt_pass_kernel<11>": ntt_traffic},
    }
    with open(os.path.join(prof, f"{args.tag}_baked_counters.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("checked and copied:", ", ".join(sorted(copies.values())), f"+ {args.tag}_baked_counters.json")


if __name__ == "__main__":
    main()
