#!/usr/bin/env python3
"""Helpers of the profiling evidence pipeline (tools/profile_round.sh -> tools/bake_counters.py -> profiles/ ->
tests/test_profiles.py).  Round 3 committed a round-1 trace as its kernel summary because a copy step took "a"
*kernel_stats.csv out of a directory that had accumulated twenty of them; here every picker insists on EXACTLY ONE
match in a directory the session created, and the consistency rules live in one function that the bake tool and the
CPU test both call.

    evidence.py manifest OUT.json            sha256 of the sources the measured numbers depend on
    evidence.py pick DIR PATTERN DEST        copy the one file under DIR matching PATTERN (fails on 0 or > 1)
    evidence.py sq DIR                       per-kernel averages of the SQ counter pass
    evidence.py show EVIDENCE_DIR            the figures a reader wants to see after a session
    evidence.py check TAG                    the consistency rules on profiles/TAG_*
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-experiments_amd", "csrc")
# files whose text decides the instruction stream of a hot kernel (not the host-side planning around it)
SOURCES = {
    "k3": ["ff29.h", "g1.h", "msm_dev.h", "bn256_constants.inc"],
    "ntt": ["ff29.h", "ntt.hip", "bn256_constants.inc"],
}
K3 = "hm::msm_accumulate_kernel"
PRECOMP = "hm::msm_precompute_chain_kernel"
TOLERANCE = 0.05          # K3 average under the profiler against the HIP-event figure of the line printed under it


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def sources_sha256(kind):
    h = hashlib.sha256()
    for name in SOURCES[kind]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def manifest():
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")))
    files += [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "include", "halo2_mi355x.h")]
    return {"files": {os.path.relpath(f, ROOT): sha256_file(f) for f in files},
            "k3_sources_sha256": sources_sha256("k3"), "ntt_sources_sha256": sources_sha256("ntt")}


def pick_one(directory, pattern):
    found = glob.glob(os.path.join(directory, "**", pattern), recursive=True)
    if len(found) != 1:
        raise SystemExit(f"evidence: expected exactly one {pattern} under {directory}, found {len(found)}: {found[:5]}")
    return found[0]


def kernel_rows(stats_csv):
    """{short kernel name: (calls, average ns)} of a rocprofv3 kernel_stats.csv."""
    out = {}
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            name = r["Name"].split("(")[0].replace("void ", "")
            out[name] = (int(r["Calls"]), float(r["AverageNs"]))
    return out


def sq_summary(directory):
    from collections import defaultdict
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if n == K3 or n.startswith("hm::ntt_pass_kernel<11"):
                    a = acc[n][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
    if not acc:
        raise SystemExit(f"evidence: no counter rows for the hot kernels under {directory}")
    return {n: {k: round(v[0] / v[1]) for k, v in d.items()} for n, d in acc.items()}


def consistency_problems(stats_csv, under_rocprof_json, bench_json=None):
    """The rules of VERDICT r3 item 1.  Returns a list of strings; empty = consistent."""
    problems = []
    rows = kernel_rows(stats_csv)
    if K3 not in rows:
        return [f"{stats_csv}: no {K3} row"]
    k3_ms = rows[K3][1] / 1e6
    with open(under_rocprof_json) as f:
        under = json.load(f)
    lines = [("bench_under_rocprof", under)]
    if bench_json and os.path.exists(bench_json):
        with open(bench_json) as f:
            lines.append(("bench", json.load(f)))
    ev = under["roofline"]["kernel_ms"]
    if abs(k3_ms - ev) > TOLERANCE * ev:
        problems.append(f"K3 average {k3_ms:.3f} ms in the trace differs from roofline.kernel_ms {ev:.3f} of the line printed under it by more than 5 %")
    for name, line in lines:
        if k3_ms > line["ms_per_step"]:
            problems.append(f"K3 average {k3_ms:.3f} ms exceeds {name}.ms_per_step {line['ms_per_step']:.3f}")
        if line["roofline"]["kernel_ms"] > line["ms_per_step"]:
            problems.append(f"{name}: roofline.kernel_ms exceeds ms_per_step")
        if not line.get("known_answer_ok"):
            problems.append(f"{name}: known_answer_ok is not true")
    table = under["config"]["base_set"].startswith("fixed-base")
    if table != (PRECOMP in rows):
        problems.append(f"config.base_set says {'table' if table else 'plain'} but the trace {'names' if PRECOMP in rows else 'lacks'} {PRECOMP}")
    steps = under["steps"] + under["warmup"]
    if rows[K3][0] < steps:
        problems.append(f"the trace holds {rows[K3][0]} K3 launches, fewer than the {steps} steps of the command")
    return problems


def check_tag(tag):
    p = os.path.join(ROOT, "profiles")
    return consistency_problems(os.path.join(p, f"{tag}_bench_kernel_stats.csv"), os.path.join(p, f"{tag}_bench_under_rocprof.json"),
                                os.path.join(p, f"{tag}_bench.json"))


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "manifest":
        with open(sys.argv[2], "w") as f:
            json.dump(manifest(), f, indent=1)
    elif cmd == "pick":
        shutil.copyfile(pick_one(sys.argv[2], sys.argv[3]), sys.argv[4])
    elif cmd == "sq":
        for n, d in sq_summary(sys.argv[2]).items():
            print(n, d)
    elif cmd == "show":
        e = sys.argv[2]
        rows = kernel_rows(os.path.join(e, "kernel_stats.csv"))
        for n, (calls, avg) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:24]:
            if n.startswith("hm::"):
                print(f"{n:52s} calls {calls:4d} avg {avg / 1e6:9.3f} ms")
        for name in ("bench.json", "bench_under_rocprof.json"):
            path = os.path.join(e, name)
            if os.path.exists(path):
                d = json.load(open(path))
                print(name, "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), "value", f"{d['value']:.4g}")
        for p in consistency_problems(os.path.join(e, "kernel_stats.csv"), os.path.join(e, "bench_under_rocprof.json"), os.path.join(e, "bench.json")):
            print("INCONSISTENT:", p)
    elif cmd == "check":
        probs = check_tag(sys.argv[2])
        for p in probs:
            print("INCONSISTENT:", p)
        sys.exit(1 if probs else 0)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
