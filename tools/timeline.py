#!/usr/bin/env python3
"""Concurrency of the small-MSM kernels inside a rocprofv3 --kernel-trace run: per kernel name the count and mean
duration, and for the whole trace window the union of busy time against the sum of durations.  Development aid."""
import csv, glob, sys, collections
root = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "hm::msm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if tail: rows = rows[-tail:]
by = collections.defaultdict(list)
for r in rows: by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in by.items(): print(f"{k:48s} n={len(v):5d} mean {sum(v)/len(v)/1e3:9.1f} us  max {max(v)/1e3:9.1f} us")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e in iv)
span = iv[-1][1] - iv[0][0]
print(f"kernels {len(iv)}  span {span/1e6:.3f} ms  busy union {busy/1e6:.3f} ms  sum of durations {tot/1e6:.3f} ms  mean concurrency {tot/busy:.2f}  idle {100*(1-busy/span):.1f} %")
