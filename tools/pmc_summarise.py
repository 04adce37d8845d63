#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE per launch for our kernels from rocprofv3 --pmc CSV output.

Units and gfx950 corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read, so the read side is
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Other access shapes are uncalibrated --
the numbers below are therefore an estimate for the gather-heavy MSM kernel."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, ctr, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != ctr:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if not name.startswith("hm::"):
                    continue
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    for name, (tot, cnt) in acc.items():
        out.setdefault(name, {})[ctr + "_KiB_per_launch"] = tot / cnt
        out[name]["launches"] = cnt
for name, d in out.items():
    f, w = d.get("FETCH_SIZE_KiB_per_launch"), d.get("WRITE_SIZE_KiB_per_launch")
    if f is not None and w is not None:
        d["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
print(json.dumps(out, indent=1, sort_keys=True))
