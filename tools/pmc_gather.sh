#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for 64-byte gathers (tools/ubench/gather64.hip): known bytes vs counters.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/pmc_gather"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/gather64 "$R/tools/ubench/gather64.hip" || exit 1
cd /tmp
for mode in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$O/m${mode}_$ctr" -- /tmp/gather64 25 26 $mode > "$O/m${mode}_$ctr.txt" 2>&1 || { tail -3 "$O/m${mode}_$ctr.txt"; exit 1; }
  done
done
cd "$R" && python3 - <<'PY'
import csv, glob, json
out = {}
for mode in (0, 1):
    d = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/pmc_gather/m{mode}_{ctr}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and "gather64_kernel" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        d[ctr + "_KiB_per_launch"] = sum(vals) / max(len(vals), 1)
    m = 1 << 26
    d["algorithmic_read_bytes"] = m * 68
    d["algorithmic_gathered_bytes"] = m * 64
    d["algorithmic_written_bytes"] = m * 4
    d["FETCH_SIZE_bytes_over_algorithmic_read"] = d["FETCH_SIZE_KiB_per_launch"] * 1024 / (m * 68)
    d["WRITE_SIZE_bytes_over_algorithmic_written"] = d["WRITE_SIZE_KiB_per_launch"] * 1024 / (m * 4)
    out["random" if mode == 0 else "sequential"] = d
    print(open(glob.glob(f"gpurun_out/pmc_gather/m{mode}_FETCH_SIZE.txt")[0]).read().strip().splitlines()[-1])
json.dump(out, open("gpurun_out/pmc_gather/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
