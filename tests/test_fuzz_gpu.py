"""tools/fuzz.py -- the differential fuzzer against the oracles (single MSMs on both layouts, whole phases through the batch call,
batched scans / divisions, coset decompositions, lookup permutations) -- a short run of every mode under pytest: a seed a round,
two cases a mode; `python tools/fuzz.py <mode> SEED CASES` runs it for as long as one likes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["single", "batch", "scans", "cosets", "lookup"])
def test_a_short_fuzz_run(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz.py"), mode, "505", "2"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "done, mismatches: 0" in r.stdout, r.stdout[-1500:]
