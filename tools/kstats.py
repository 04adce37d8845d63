#!/usr/bin/env python3
"""Print per-kernel averages of a rocprofv3 kernel_stats.csv (newest under the given directory)."""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20000
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("hm::") and float(r["AverageNs"]) > thr:
        print("%-52s calls %4s avg %9.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e6))
