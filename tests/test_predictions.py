"""DESIGN.md section 6's multi-GPU predictions as data (profiles/r05_predictions.json) and the tool that diffs a driver scaling record
against them (tools/compare_scale.py) -- VERDICT r4 item 5.  No multi-GPU run exists yet: the comparison is exercised on a synthetic
record shaped like bench.py's compact lines, and on whatever SCALE_r*.json the repository holds (skipped records so far)."""
import glob
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import compare_scale  # noqa: E402


def _predictions():
    with open(compare_scale.PREDICTIONS) as f:
        return json.load(f)


def test_predictions_are_well_formed_and_name_keys_the_line_carries():
    import bench
    p = _predictions()
    specs = [p["line_value"], p["line_ms_per_step"]] + list(p["summary"].values()) + list(p["extras"].values())
    for spec in specs:
        assert spec["by_n_gpus"] and all(n in ("1", "2", "4", "8") and lo < hi for n, (lo, hi) in spec["by_n_gpus"].items())
    # every predicted summary key is one compact_line() can emit
    with open(os.path.join(ROOT, "profiles", "r04_k_bench.json")) as f:
        full = json.load(f)
    full["one_process"] = {"msm_split": {"ms_per_msm": 1.0}, "create_proof_replay": {"device_resident_s": {"total": 0.01}}}
    emitted = set(bench.compact_line(full, None)["summary"])
    assert set(p["summary"]) <= emitted, set(p["summary"]) - emitted
    # weak scaling bands are consistent with the efficiencies written next to them
    v = p["line_value"]["by_n_gpus"]
    assert p["efficiency_expected"]["weak_value_n8"][0] <= v["8"][0] / (8 * v["1"][1]) * 1.08


def test_the_comparison_reads_lines_wherever_the_driver_puts_them():
    p = _predictions()
    line = lambda n, value, ms, **summary: {"metric": "BN256 G1 MSM throughput", "n_gpus": n, "value": value, "ms_per_step": ms, "summary": summary}
    record = {"runs": [{"n": 1, "parsed": line(1, 9.7e8, 17.3, k18_replay_ms=34.0, msm_2_26_global_points_per_s=1.03e9)},
                       {"n": 8, "stdout_tail": "noise\n" + json.dumps(line(8, 5.0e9, 26.0, k18_replay_ms=11.0, one_process_msm_2_26_ms=16.0)) + "\n"}],
              "efficiency": {"8": 0.64}}
    rows = compare_scale.compare(list(compare_scale.bench_lines(record)), p)
    got = {(name, n): verdict for name, n, _, _, _, verdict in rows}
    assert got[("value", 1)] == "inside" and got[("ms_per_step", 1)] == "inside" and got[("k18_replay_ms", 1)] == "inside"
    assert got[("value", 8)] == "below" and got[("ms_per_step", 8)] == "above"          # a real shortfall is reported, not hidden
    assert got[("k18_replay_ms", 8)] == "inside" and got[("one_process_msm_2_26_ms", 8)] == "inside"
    assert ("one_process_msm_2_26_ms", 1) not in got                                     # no prediction, no row


def test_committed_scale_records_are_diffed():
    """Every SCALE_r*.json at the root goes through the comparison; figures outside their band are WARNINGS (information for the next
    round), never failures."""
    p = _predictions()
    for path in sorted(glob.glob(os.path.join(ROOT, "SCALE_r*.json"))):
        with open(path) as f:
            rec = json.load(f)
        if isinstance(rec, dict) and rec.get("skipped"):
            continue
        for name, n, got, lo, hi, verdict in compare_scale.compare(list(compare_scale.bench_lines(rec)), p):
            if verdict != "inside":
                warnings.warn(f"{os.path.basename(path)}: {name} at N={n} measured {got:.4g}, predicted [{lo:.4g}, {hi:.4g}] ({verdict})")
