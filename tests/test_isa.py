"""What the compiler made of the kernels, read from the BUILT objects (tools/isa_check.py; CPU only, a few seconds):

* no production kernel reads its kernarg segment with a VECTOR load.  A dynamically indexed byte table inside a by-value argument
  compiles to `global_load_ubyte v, v, s[0:1] offset:N`; a stand-alone kernel of that shape faulted once on this stack and the cause
  is not established (profiles/r05_kernarg_isa.txt: the load is legal ISA, s[0:1] is intact, the offset clean), so the by-value tables
  the kernels do take (NttCosetTables, PoCols / PoPoints, MsmGroupScalars, LkPtrs) are held to scalar loads here -- ADVICE r4;
* the detector itself is checked on the ubench that has the pattern (compiled, never launched);
* the north star's "no MFMA", and no scratch in the hot kernels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check  # noqa: E402


@pytest.fixture(scope="module")
def rows():
    from halo2_experiments_amd import _lib
    _lib.load()                                                 # builds csrc/ on first use
    missing = [o for o in isa_check.OBJECTS if not os.path.exists(os.path.join(isa_check.CSRC, o))]
    if missing:
        _lib.build()
    out = {}
    for obj in isa_check.OBJECTS:
        for name, r in isa_check.summary(obj).items():
            out[(obj, name)] = r
    return out


def test_every_device_translation_unit_was_read(rows):
    objs = {o for o, _ in rows}
    assert objs == {"ntt.o", "poly.o", "polyops.o", "lookup.o", "graph.o", "msm.o", "msm_small.o"}       # capi / multi are host only
    assert len(rows) >= 90 and all(r["kernarg_pair"] for r in rows.values())


def test_no_kernel_reads_its_kernarg_segment_with_vector_loads(rows):
    bad = {name: r["kernarg_vector_accesses"] for (_, name), r in rows.items() if r["kernarg_vector_accesses"]}
    assert bad == {}
    by_value_tables = [n for _, n in rows if any(t in n for t in ("NttCosetTables", "PoCols", "PoPoints", "MsmGroupScalars", "LkPtrs",
                                                                  "SmallGroupScalars", "LcArgs"))]
    assert len(by_value_tables) >= 10                           # the kernels the rule is about are among those read


def test_the_detector_sees_the_pattern_in_the_ubench(tmp_path):
    """tools/ubench/kernarg_byval.hip, device code only, compiled here and NOT launched anywhere: the variant with the byte table has
    the vector load through the kernarg pointer, the variant without has none."""
    asm = tmp_path / "kb.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", str(asm),
                    os.path.join(ROOT, "tools", "ubench", "kernarg_byval.hip")], check=True, capture_output=True)
    kernels, cur = {}, None
    for line in asm.read_text().splitlines():
        if line.startswith("_Z12byval_kernel") and line.rstrip().endswith(":") or (line.startswith("_Z12byval_kernel") and ":" in line):
            cur = kernels.setdefault(line.split(":")[0], [])
        elif line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None and line.startswith("\t") and not line.startswith("\t."):
            cur.append(line.split(";")[0].strip())
    with_bytes = next(v for k, v in kernels.items() if "ILb1E" in k)
    without = next(v for k, v in kernels.items() if "ILb0E" in k)
    pair, hits = isa_check.kernarg_vector_accesses(with_bytes)
    assert pair == "s[0:1]" and len(hits) == 1 and hits[0][1].startswith("global_load_ubyte") and "s[0:1]" in hits[0][1]
    assert isa_check.kernarg_vector_accesses(without) == ("s[0:1]", [])


def test_no_mfma_and_no_scratch_in_the_hot_kernels(rows):
    assert sum(r["mfma"] for r in rows.values()) == 0           # north star: 256-bit modular integer work, not a dense contraction
    hot = ("msm_accumulate_kernel", "ntt_pass_kernel", "graph_evaluate_kernel", "msm_part1_scatter_v2", "msm_part2_scatter_v2", "msm_digits_kernel")
    seen = set()
    for (_, name), r in rows.items():
        for hk in hot:
            if hk in name:
                seen.add(hk)
                assert r["scratch"] == 0, name
    assert seen == set(hot)
    spilling = sorted({n.split("EE")[0][:60] for (_, n), r in rows.items() if r["scratch"]})
    assert all(any(k in n for k in ("g1_fixed_table_kernel", "fr_batch_invert_kernel", "msm_s_reduce1_kernel")) for n in spilling), spilling
