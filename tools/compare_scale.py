#!/usr/bin/env python3
"""Diff a driver scaling record (SCALE_rNN.json: bench.py's compact lines at N = 1, 2, 4, 8) against the predictions written down before
any multi-GPU run existed (profiles/r05_predictions.json = DESIGN.md section 6).

    python tools/compare_scale.py [SCALE_r05.json ...]      # default: every SCALE_r*.json at the repository root

The record's layout belongs to the driver; this reader takes every JSON object in it that carries `n_gpus` and `value` (a parsed
bench line) wherever it sits.  Prints one row per predicted figure and N: measured, band, verdict."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREDICTIONS = os.path.join(ROOT, "profiles", "r05_predictions.json")


def bench_lines(obj):
    """Every dict with n_gpus and value inside an arbitrary JSON tree (strings that are themselves JSON lines included)."""
    if isinstance(obj, dict):
        if "n_gpus" in obj and "value" in obj:
            yield obj
        for v in obj.values():
            yield from bench_lines(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from bench_lines(v)
    elif isinstance(obj, str) and '"n_gpus"' in obj:
        for line in obj.splitlines():
            line = line.strip()
            if line.startswith("{") and line.endswith("}"):
                try:
                    yield from bench_lines(json.loads(line))
                except ValueError:
                    pass


def compare(lines, predictions):
    """-> rows [(figure, n_gpus, measured, lo, hi, verdict)] for every predicted figure the lines carry."""
    rows, seen = [], set()
    for line in lines:
        n = str(line.get("n_gpus"))
        figures = [("value", line.get("value"), predictions["line_value"]), ("ms_per_step", line.get("ms_per_step"), predictions["line_ms_per_step"])]
        for key, spec in predictions["summary"].items():
            figures.append((key, (line.get("summary") or {}).get(key), spec))
        for name, got, spec in figures:
            band = spec["by_n_gpus"].get(n)
            if got is None or band is None or (name, n) in seen:
                continue
            seen.add((name, n))
            lo, hi = band
            verdict = "inside" if lo <= got <= hi else ("below" if got < lo else "above")
            rows.append((name, int(n), got, lo, hi, verdict))
    return sorted(rows)


def main():
    with open(PREDICTIONS) as f:
        predictions = json.load(f)
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "SCALE_r*.json")))
    for path in files:
        with open(path) as f:
            rec = json.load(f)
        if isinstance(rec, dict) and rec.get("skipped"):
            print(f"{os.path.basename(path)}: skipped ({rec.get('reason', '')[:80]})")
            continue
        rows = compare(list(bench_lines(rec)), predictions)
        print(f"{os.path.basename(path)}: {len(rows)} predicted figure(s) found")
        for name, n, got, lo, hi, verdict in rows:
            print(f"  {name:32s} N={n}  measured {got:.4g}  predicted [{lo:.4g}, {hi:.4g}]  {verdict}")


if __name__ == "__main__":
    main()
