#!/usr/bin/env python3
"""Read the gfx950 ISA of the BUILT objects (csrc/*.o -> .hip_fatbin -> clang-offload-bundler -> llvm-objdump) and report, per kernel:

  * vector memory instructions addressed THROUGH THE KERNARG POINTER (`global_load_ubyte v, v, s[0:1] offset:N`): what a dynamically
    indexed BYTE table inside a by-value kernel argument compiles to (tools/ubench/kernarg_byval.hip).  A stand-alone kernel of that
    shape faulted once on this stack (profiles/r04_kernarg_byval.txt); the load is legal ISA and the cause is NOT established
    (profiles/r05_kernarg_isa.txt), so the production kernels simply must not contain the pattern: their by-value tables
    (NttCosetTables, PoCols / PoPoints, MsmGroupScalars, LkPtrs ...) are 32- / 64-bit entries indexed by wave-uniform values and
    must keep compiling to scalar loads.  tests/test_isa.py runs this on every kernel of the library (CPU only).
  * MFMA instructions (the north star: none -- 256-bit modular integer work), scratch use, VGPR counts.

    python tools/isa_check.py                  # summary of every object
    python tools/isa_check.py ntt.o --dump     # the offending instructions with context
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-experiments_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
OBJECTS = ["capi.o", "multi.o", "xfer.o", "ntt.o", "poly.o", "polyops.o", "lookup.o", "graph.o", "msm.o", "msm_small.o"]

_VMEM = re.compile(r"^\s*(global_|flat_|buffer_|scratch_)(load|store|atomic)\w*\s+(.*)$")
_SLOAD = re.compile(r"^\s*s_load_dword(?:x\d+)?\s+\S+,\s*(s\[\d+:\d+\])")
_SDEST = re.compile(r"^\s*s_\w+\s+(s\d+|s\[\d+:\d+\])")


def device_disassembly(obj_path: str) -> str:
    """The gfx950 code object inside a host object built by hipcc, disassembled."""
    with tempfile.TemporaryDirectory(prefix="hm_isa_") as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj_path, os.path.join(tmp, "copy.o")],
                           capture_output=True, text=True)
        if r.returncode != 0:
            if "not found" in r.stderr:                     # a host-only translation unit (capi.hip, multi.hip, xfer.hip): no device code
                return ""
            raise RuntimeError(r.stderr)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--input={fat}", f"--targets={TARGET}", "--unbundle", f"--output={co}"],
                       check=True, capture_output=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def kernels(disassembly: str):
    """-> {mangled name: [instruction text, ...]}"""
    out, cur = {}, None
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
        elif cur is not None and line.strip():
            cur.append(re.sub(r"\s*//.*$", "", line).strip())
    return out


def _regs(operand: str):
    m = re.match(r"s\[(\d+):(\d+)\]$", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", operand)
    return {int(m.group(1))} if m else set()


def kernarg_vector_accesses(insts):
    """Vector memory instructions that use the kernarg pointer pair as their scalar base.  The pair = the base of the kernel's first
    s_load (every kernel here loads its arguments first); tracked until a scalar instruction overwrites either register (a linear
    scan: control flow is ignored, which can only end the tracking early at a join -- none of the kernels reuses the pair)."""
    pair = None
    hits = []
    for i, ins in enumerate(insts):
        if pair is None:
            m = _SLOAD.match(ins)
            if m:
                pair = m.group(1)
            continue
        m = _VMEM.match(ins)
        if m and re.search(r"(^|[\s,])" + re.escape(pair) + r"($|[\s,])", m.group(3)):
            hits.append((i, ins))
            continue
        m = _SDEST.match(ins)
        if m and _regs(m.group(1)) & _regs(pair) and not _SLOAD.match(ins):
            break
        if m and _SLOAD.match(ins) and _regs(m.group(1)) & _regs(pair):
            break
    return pair, hits


def summary(obj: str):
    dis = device_disassembly(os.path.join(CSRC, obj))
    rows = {}
    for name, insts in kernels(dis).items():
        if not insts or name.endswith(".kd"):
            continue
        pair, hits = kernarg_vector_accesses(insts)
        rows[name] = {"instructions": len(insts), "kernarg_pair": pair, "kernarg_vector_accesses": hits,
                      "mfma": sum(1 for x in insts if x.startswith("v_mfma") or x.startswith("v_smfma")),
                      "scratch": sum(1 for x in insts if x.startswith("scratch_"))}
    return rows


def main():
    objs = [a for a in sys.argv[1:] if not a.startswith("--")] or OBJECTS
    dump = "--dump" in sys.argv
    bad = 0
    for obj in objs:
        for name, r in summary(obj).items():
            flag = "  <-- vector access through the kernarg pointer" if r["kernarg_vector_accesses"] else ""
            print(f"{obj:14s} {r['instructions']:7d} inst  mfma {r['mfma']}  scratch {r['scratch']:4d}  kernarg {r['kernarg_pair']}  {name[:90]}{flag}")
            if r["kernarg_vector_accesses"]:
                bad += 1
                if dump:
                    for i, ins in r["kernarg_vector_accesses"]:
                        print("      ", i, ins)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
