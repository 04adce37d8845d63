#!/usr/bin/env python3
"""Per-kernel listing (start offset, duration, queue) of the last `tail` hm::msm_ kernels of a rocprofv3 --kernel-trace run,
and the time during which an msm_accumulate_kernel was running: development aid.
    python tools/gantt.py gpurun_out/phase_trace 200"""
import csv, glob, sys
root = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 200
f = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "hm::msm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-tail:]
t0 = int(rows[0]["Start_Timestamp"])
queues = {}
for r in rows:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("hm::msm_", "").replace("_kernel", "")[:40]
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  q{q:<2d} {' ' * (2 * q)}{name}  grid {r.get('Grid_Size', '?')}")
acc = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in rows if "msm_accumulate_kernel" in r["Kernel_Name"])
busy, ce = 0, None
for s, e in acc:
    if ce is None or s > ce: busy += e - s; ce = e
    elif e > ce: busy += e - ce; ce = e
print(f"accumulate kernels {len(acc)}: union {busy/1e3:.1f} us of span {(int(rows[-1]['End_Timestamp']) - t0)/1e3:.1f} us")
