#!/usr/bin/env python3
"""Where a slow hm_ntt_bn256_fr call spends its time: 20 host-pointer transforms of one 2^21 array (64 MiB each way) in the state a
replay leaves the process in, each with its h2d / device / d2h microseconds from hm_get_stats.  Development aid.

    python tools/ntt_ext_probe.py [replay-first: 0|1] [array: zeros|full]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import halo2_experiments_amd as h  # noqa: E402
from halo2_experiments_amd import _lib  # noqa: E402
from halo2_experiments_amd.domain import EvaluationDomain, fr_words  # noqa: E402

replay_first = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
kind = sys.argv[2] if len(sys.argv) > 2 else "zeros"
dev = torch.device("cuda", 0)
if replay_first:
    from halo2_experiments_amd.replay import run_replay
    run_replay("merkle_sum_tree_k18", device=dev, include_host_pointer_estimate=False)
dom = EvaluationDomain(7, 18)
hs = h.random_fr(dom.n, 5, dev).cpu().numpy().view(np.uint64).copy()
if kind == "zeros":
    a = np.zeros((dom.extended_len(), 4), dtype=np.uint64)
    a[:dom.n] = hs
elif kind in ("thp", "nothp"):            # an anonymous mapping with / without transparent huge pages
    import mmap
    nbytes = dom.extended_len() * 32
    mm = mmap.mmap(-1, nbytes + (2 << 20))
    mm.madvise(mmap.MADV_HUGEPAGE if kind == "thp" else mmap.MADV_NOHUGEPAGE)
    base = np.frombuffer(mm, dtype=np.uint8)
    off = (-base.ctypes.data) % (2 << 20)
    a = base[off:off + nbytes].view(np.uint64).reshape(-1, 4)
    a[:] = h.random_fr(dom.extended_len(), 6, dev).cpu().numpy().view(np.uint64)
else:
    a = h.random_fr(dom.extended_len(), 6, dev).cpu().numpy().view(np.uint64).copy()
w = fr_words(dom.extended_omega)
lib = _lib.load()


def stats():
    st = _lib.Stats()
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    return st.ntt_h2d_us, st.ntt_device_us, st.ntt_d2h_us


def numa_report(arr):
    """Where this process runs and where its pages are: the allowed CPUs, the nodes' CPU lists, the GPU's node, the node(s) of `arr`."""
    import glob
    out = {}
    try:
        out["cpus_allowed"] = [l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith("Cpus_allowed_list")][0]
        out["mems_allowed"] = [l.split(":")[1].strip() for l in open("/proc/self/status") if l.startswith("Mems_allowed_list")][0]
        out["running_on_cpu"] = ctypes.CDLL(None).sched_getcpu()
        out["nodes"] = {os.path.basename(os.path.dirname(f)): open(f).read().strip() for f in sorted(glob.glob("/sys/devices/system/node/node*/cpulist"))}
        out["gpu_numa_node"] = {f.split("/")[4]: open(f).read().strip() for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node"))}
        addr = arr.ctypes.data
        for line in open("/proc/self/numa_maps"):
            lo = int(line.split()[0], 16)
            if lo <= addr < lo + arr.nbytes + (1 << 21) and ("anon=" in line) and abs(lo - addr) < (1 << 22):
                out["array_pages"] = " ".join(t for t in line.split() if t.startswith(("N", "anon", "kernelpagesize")))
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    return out


if len(sys.argv) > 3:                       # free that many GiB of device memory just before the calls (does the driver's clearing of freed VRAM share the copy engines?)
    gib = int(sys.argv[3])
    h.best_fft(a, w, dom.extended_k)        # lanes, tables, staging exist
    x = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    x.fill_(1)
    torch.cuda.synchronize()
    del x
    t_free = time.perf_counter()
    torch.cuda.empty_cache()
    print(f"freed {gib} GiB in {(time.perf_counter() - t_free) * 1e3:.1f} ms (hipFree)")
print(f"replay_first={replay_first} array={kind}")
print("numa:", numa_report(a))
try:
    print("thp:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
    addr = a.ctypes.data
    cur, huge = None, None
    for line in open("/proc/self/smaps"):
        if "-" in line.split()[0] and len(line.split()[0].split("-")) == 2 and not line.startswith(("Size", "Rss")):
            try:
                lo, hi = (int(x, 16) for x in line.split()[0].split("-"))
                cur = (lo, hi)
            except ValueError:
                pass
        elif line.startswith("AnonHugePages") and cur and cur[0] <= addr < cur[1]:
            huge = (cur[1] - cur[0], line.split()[1])
    print("array mapping bytes / AnonHugePages kB:", huge)
except Exception as e:  # noqa: BLE001
    print("thp: ?", e)
t_begin = time.perf_counter()
for i in range(20 if len(sys.argv) <= 3 else 60):
    s0 = stats()
    t0 = time.perf_counter()
    h.best_fft(a, w, dom.extended_k)
    dt = (time.perf_counter() - t0) * 1e3
    s1 = stats()
    print(f"call {i:2d} at {(time.perf_counter() - t_begin) * 1e3:7.1f} ms: {dt:7.2f} ms   h2d {s1[0] - s0[0]:8.0f} us   device {s1[1] - s0[1]:8.0f} us   d2h {s1[2] - s0[2]:8.0f} us")
