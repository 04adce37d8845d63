"""The committed rocprofv3 evidence must describe the build it is filed under (the round-3 kernel summary
was a round-1 trace, picked by `glob(...)[0]` out of a directory that had accumulated twenty).  CPU-only: reads
profiles/ and applies the rules tools/bake_counters.py enforces before it copies anything there."""
import glob
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import evidence  # noqa: E402

PROFILES = os.path.join(ROOT, "profiles")


def _tags():
    """rNN_x tags that have a kernel summary, from round 3 on (earlier rounds predate the rules)."""
    tags = []
    for f in sorted(glob.glob(os.path.join(PROFILES, "r*_bench_kernel_stats.csv"))):
        m = re.match(r"(r(\d\d)_\w)_bench_kernel_stats\.csv$", os.path.basename(f))
        if m and int(m.group(2)) >= 3:
            tags.append(m.group(1))
    return tags


def test_there_is_evidence_to_check():
    assert _tags(), "profiles/ holds no rNN_x_bench_kernel_stats.csv from round 3 on"


@pytest.mark.parametrize("tag", _tags())
def test_kernel_summary_matches_the_bench_lines_filed_with_it(tag):
    """K3's average in the trace <= ms_per_step, within 5 % of roofline.kernel_ms of the line printed under the profiler,
    and the trace names msm_precompute_chain_kernel iff config.base_set says fixed-base table."""
    assert evidence.check_tag(tag) == []


def test_newest_round_is_what_bench_py_quotes():
    """bench.py takes its baked counter figures from the greatest tag; that file must point at the kernel summary
    committed beside it, by content hash (round 4 on)."""
    sys.path.insert(0, ROOT)
    baked = sorted(glob.glob(os.path.join(PROFILES, "r[0-9][0-9]*_baked_counters.json")))
    assert baked
    with open(baked[-1]) as f:
        d = json.load(f)
    if "kernel_stats" not in d:          # round 3's file predates the field
        pytest.skip("newest baked counter file predates the kernel_stats field")
    path = os.path.join(ROOT, d["kernel_stats"]["file"])
    assert evidence.sha256_file(path) == d["kernel_stats"]["sha256"]
    rows = evidence.kernel_rows(path)
    assert abs(rows[evidence.K3][1] / 1e6 - d["kernel_stats"]["k3_average_ms"]) < 1e-6
    assert evidence.check_tag(d["tag"]) == []


def test_a_stale_trace_is_refused(tmp_path):
    """The round-1 trace that round 3 committed by mistake fails the rules against the round-3 lines."""
    stale = os.path.join(PROFILES, "r01_c_bench_kernel_stats.csv")
    probs = evidence.consistency_problems(stale, os.path.join(PROFILES, "r03_b_bench_under_rocprof.json"),
                                          os.path.join(PROFILES, "r03_b_bench.json"))
    assert probs and any("exceeds" in p or "differs" in p for p in probs)


def test_pickers_insist_on_exactly_one_match(tmp_path):
    (tmp_path / "a").mkdir()
    (tmp_path / "a" / "1_kernel_stats.csv").write_text("x")
    assert evidence.pick_one(str(tmp_path), "*kernel_stats.csv").endswith("1_kernel_stats.csv")
    (tmp_path / "a" / "2_kernel_stats.csv").write_text("y")
    with pytest.raises(SystemExit):
        evidence.pick_one(str(tmp_path), "*kernel_stats.csv")


def test_live_pmc_never_raises_and_falls_back(monkeypatch):
    """bench.py measures the PMC counters in child processes under rocprofv3; a box without the profiler, or a failing
    pass, must come back as {"error": ...} (the line then quotes the committed file), never as an exception."""
    import importlib
    import shutil
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert "error" in bench.live_pmc(24, 24, True)
    monkeypatch.setattr(shutil, "which", lambda name: "/usr/bin/false")

    class Failed:
        returncode, stderr, stdout = 1, "boom", ""
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Failed())
    out = bench.live_pmc(24, 24, True)
    assert "error" in out and "rc 1" in out["error"]

    def raising(*a, **k):
        raise subprocess.TimeoutExpired(cmd="rocprofv3", timeout=1)
    monkeypatch.setattr(subprocess, "run", raising)
    assert "TimeoutExpired" in bench.live_pmc(24, 24, True)["error"]
    assert bench.newest_baked_counters_file().startswith("profiles/r")
