// xfer.hip -- the host <-> device copies of the host-pointer entry points (hm_msm_bn256_g1*, hm_ntt_bn256_fr, hm_coeff_to_extended /
// hm_extended_to_coeff, the batch call's scalar uploads, hm_copy_to_device / hm_copy_to_host).
//
// Two paths.  DIRECT: hipMemcpy on the caller's pointers.  LANES: the library's own pinned staging, four lanes (up to 8 by
// HALO2_MI355X_XFER_LANES), each a host thread with its own stream and two 2 MiB pinned slots moving one contiguous share of the
// transfer (memcpy into a slot / DMA from the other, and the reverse): 64 MiB in 1.4 ms either way (46 GB/s) against 1.2-1.3 ms direct.
//
// FINDING (rounds 5-6; tools/ubench/hostcopy_probe.hip, profiles/r06_hostcopy_probe.txt, profiles/r05_host_copies.txt).  For pageable
// memory of these sizes hipMemcpy PINS the caller's pages on the fly -- a userptr registration with the kernel driver, cached by the
// runtime per (address, size) -- and then DMAs at 50-55 GB/s.  What that costs is a property of the BOX, not of the copy:
//   * on most boxes of the pool pinning costs 0.1-0.2 us per 4 KiB page (hipHostRegister of 64 MiB: 2.2 ms; first hipMemcpy from a fresh
//     64 MiB mapping 1.64 ms against 1.33 ms for the second; a re-mapped range at the same address pays it again on every OTHER cycle:
//     2.5 / 1.4 / 2.5 / 1.4 ms -- the runtime's cached pin is found stale and rebuilt);
//   * on some boxes it costs ~9 us per page (2.2-2.4 ms per MiB: the 8 MiB upload of a k = 18 column 17.7 ms instead of 0.16, a 4 MiB one
//     9.9 ms: the same per-page figure at every size, round 5's "stalls" of 7-27 ms) -- and a prover's arrays are NEW on every call
//     (a fresh Vec per polynomial, unmapped afterwards: glibc serves anything above 32 MiB straight from mmap), so there EVERY copy is a
//     first copy.  That, not a transient, is what round 5's timing detector kept tripping on; whether a session saw it depended on the
//     box it landed on (3 of ~20 sessions in round 5, the first session of round 6).
// The lanes never hand the caller's memory to the driver: nothing is pinned, nothing the caller maps or unmaps can invalidate anything
// the GPU queues depend on, and the cost is the same on every box: +0.2 ms per 64 MiB against the direct path on a fast-pinning box.
//
// POLICY -- a rule on the RANGE, never on the clock (HALO2_MI355X_HOST_COPIES = auto | lanes | direct, hm_set_host_copies; default auto):
//   auto    a host range goes DIRECT only if the caller has declared it long-lived with hm_host_register (pinned once, by the caller's
//           choice: an SRS kept in memory, a reused staging buffer) -- the copy is then a plain DMA from registered memory; every other
//           range of 256 KiB or more goes through the LANES; below 256 KiB straight to hipMemcpy (the runtime stages those itself, no pin);
//   lanes   registered ranges too;
//   direct  everything to hipMemcpy (what round 5's default did until its second "stall"): for boxes known to pin fast.
// hm_get_stats counts the copies each way.  xfer_prefault first-touches a destination the process has never written (the fresh Vec
// of a result) from helper threads while the transform is still running -- only the direct path needs it (a direct copy into untouched
// pages takes its faults one by one, 5 ms per 64 MiB; the lanes' eight threads take them in parallel while they copy out).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "hm_internal.h"

namespace hm {

constexpr size_t kXferSlot = (size_t)2 << 20;
constexpr size_t kXferDirectBelow = (size_t)256 << 10;
constexpr int kXferDefaultLanes = 4;

enum { XFER_AUTO = 0, XFER_LANES = 1, XFER_DIRECT = 2 };
static std::atomic<int>& xfer_policy_cell() {
  static std::atomic<int> v{[] {
    const char* e = std::getenv("HALO2_MI355X_HOST_COPIES");
    if (!e) return (int)XFER_AUTO;
    const std::string s(e);
    return s == "lanes" ? (int)XFER_LANES : s == "direct" ? (int)XFER_DIRECT : (int)XFER_AUTO;
  }()};
  return v;
}
static int xfer_policy() { return xfer_policy_cell().load(std::memory_order_relaxed); }
int xfer_set_policy(int mode) {
  if (mode < XFER_AUTO || mode > XFER_DIRECT) return -1;
  xfer_policy_cell().store(mode, std::memory_order_relaxed);
  return 0;
}

static double xfer_now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- ranges the caller has registered (hm_host_register): process-wide, a handful of entries --------------------------------------------
struct HostRange {
  const char* lo;
  size_t bytes;
};
static std::mutex& host_ranges_mu() {
  static std::mutex m;
  return m;
}
static std::vector<HostRange>& host_ranges() {
  static std::vector<HostRange> v;
  return v;
}
static bool host_range_registered(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(host_ranges_mu());
  const char* q = (const char*)p;
  for (const HostRange& r : host_ranges())
    if (q >= r.lo && bytes <= r.bytes && (size_t)(q - r.lo) <= r.bytes - bytes) return true;
  return false;
}
int xfer_host_register(const void* p, size_t bytes) {
  if (!p || bytes == 0) return hm_fail(HM_ERR_BAD_ARG, "hm_host_register: null or empty range");
  {
    std::lock_guard<std::mutex> lk(host_ranges_mu());
    const char* q = (const char*)p;
    for (const HostRange& r : host_ranges())
      if (q < r.lo + r.bytes && r.lo < q + bytes) return hm_fail(HM_ERR_BAD_ARG, "hm_host_register: the range overlaps a registered one");
  }
  const hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return hm_fail(HM_ERR_HIP, std::string("hm_host_register: ") + hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lk(host_ranges_mu());
  host_ranges().push_back(HostRange{(const char*)p, bytes});
  return HM_OK;
}
int xfer_host_unregister(const void* p) {
  {
    std::lock_guard<std::mutex> lk(host_ranges_mu());
    auto& v = host_ranges();
    size_t i = 0;
    while (i < v.size() && v[i].lo != (const char*)p) ++i;
    if (i == v.size()) return hm_fail(HM_ERR_BAD_ARG, "hm_host_unregister: not the start of a registered range");
    v.erase(v.begin() + (long)i);
  }
  const hipError_t e = hipHostUnregister(const_cast<void*>(p));
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return hm_fail(HM_ERR_HIP, std::string("hm_host_unregister: ") + hipGetErrorString(e));
  }
  return HM_OK;
}
size_t xfer_host_ranges() {
  std::lock_guard<std::mutex> lk(host_ranges_mu());
  return host_ranges().size();
}

// First-touch the pages of a host range from several threads WITHOUT changing its contents (every page's first byte is read and
// written back).  Threads that cannot be started leave their share to the copy.  Nothing here can fail a call.
void xfer_prefault(void* p, size_t bytes) {
  constexpr size_t kPage = 4096;
  if (bytes < ((size_t)4 << 20)) return;
  volatile unsigned char* base = (volatile unsigned char*)p;
  auto touch = [base, bytes](size_t lo, size_t hi) {
    for (size_t off = lo; off < hi && off < bytes; off += kPage) {
      const unsigned char v = base[off];
      base[off] = v;
    }
  };
  const unsigned nt = HM_XFER_LANES;
  const size_t share = ((bytes / nt) / kPage + 1) * kPage;
  JoinOnExit pool;
  for (unsigned t = 1; t < nt; ++t) {
    const size_t lo = (size_t)t * share, hi = lo + share;
    if (lo >= bytes) break;
    if (!spawn_or_false(pool, "prefault", [touch, lo, hi] { touch(lo, hi); })) break;
  }
  touch(0, share);
}

void xfer_release(DeviceCtx& ctx) {       // ctx.mu held (hm_shutdown)
  std::lock_guard<std::mutex> lk(ctx.xfer.mu);
  for (auto& l : ctx.xfer.lanes) {
    for (int s = 0; s < 2; ++s) {
      if (l.ev[s]) (void)hipEventDestroy(l.ev[s]);
      l.pin[s] = nullptr;
      l.ev[s] = nullptr;
    }
    if (l.stream) (void)hipStreamDestroy(l.stream);
    l.stream = nullptr;
  }
  if (ctx.xfer.block) (void)hipHostFree(ctx.xfer.block);
  if (ctx.xfer.d_warm) (void)hipFree(ctx.xfer.d_warm);
  ctx.xfer.block = nullptr;
  ctx.xfer.d_warm = nullptr;
  ctx.xfer.ready = 0;
}

// lanes [0, want) exist afterwards, or fewer when pinned memory / streams cannot be had (at least one, else 0).
// New lanes are WARMED before they are handed out: one slot-sized device-to-host copy on every lane at once, twice.  Without it the
// first device-to-host copy of every lane beyond the second blocked its issuing hipMemcpyAsync for ~7 ms -- found in round 6 as 8.3 ms
// instead of 1.8 for the download of a 64 MiB array in six of a process's first seven calls with eight lanes, and mistaken for a
// copy "stall" by round 5's timing detector (tools/ntt_ext_probe.py, profiles/r06_lane_warmup.txt: the time is inside the issue
// call, not in the DMA nor in the host memcpy; the count is lanes - 2 whatever the order).  A warm-up of the streams one by one does
// not remove it; all lanes copying at once does, for four lanes.
static int xfer_prepare(HostXfer& x, int want) {
  if (x.ready >= want) return want;
  if (!x.block && hipHostMalloc(&x.block, (size_t)HM_XFER_LANES * 2 * kXferSlot, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    x.block = nullptr;
    return x.ready;
  }
  if (!x.d_warm && hipMalloc(&x.d_warm, kXferSlot) != hipSuccess) {
    (void)hipGetLastError();
    x.d_warm = nullptr;                   // lanes work without the warm-up
  }
  bool made = false;
  for (int i = x.ready; i < want; ++i) {
    XferLane& l = x.lanes[i];
    bool ok = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess;
    for (int s = 0; s < 2 && ok; ++s) {
      l.pin[s] = (char*)x.block + ((size_t)i * 2 + s) * kXferSlot;
      ok = hipEventCreateWithFlags(&l.ev[s], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      for (int s = 0; s < 2; ++s) {
        if (l.ev[s]) (void)hipEventDestroy(l.ev[s]);
        l.pin[s] = nullptr;
        l.ev[s] = nullptr;
      }
      if (l.stream) (void)hipStreamDestroy(l.stream);
      l.stream = nullptr;
      break;
    }
    x.ready = i + 1;
    made = true;
  }
  if (made && x.d_warm) {
    // all lanes at once, both directions, twice: whatever the runtime sets up the first time that many streams copy concurrently
    // is set up here (errors are the first real transfer's to report)
    for (int round = 0; round < 2; ++round) {
      for (int i = 0; i < x.ready; ++i) {
        XferLane& l = x.lanes[i];
        (void)hipMemcpyAsync(l.pin[0], x.d_warm, kXferSlot, hipMemcpyDeviceToHost, l.stream);
        (void)hipEventRecord(l.ev[0], l.stream);
      }
      for (int i = 0; i < x.ready; ++i) (void)hipEventSynchronize(x.lanes[i].ev[0]);
    }
    (void)hipGetLastError();
  }
  return x.ready < want ? x.ready : want;
}

// Chunks of one lane's share: uniform, one slot each.  Measured negative (round 6, profiles/r06_host_copies.txt): chunks ramping
// 256 KiB -> 2 MiB -> 256 KiB (to shorten the pipeline's fill and drain) made every shape slower -- a phase of 19 commitments 13.3 ms against
// 10.9, coeff_to_extended into a fresh array 6.0 against 5.1 -- as round 5's 512 KiB ring had: the fixed cost per DMA (~15 us, serialised
// on the copy engines over all lanes) outweighs the shorter fill.  32 DMAs of 2 MiB per 64 MiB is where the two costs balance.  The same
// for SMALL shares (a quarter of the share per chunk, so that a lane moving 2 MiB of an 8 MiB column overlaps its memcpy with its DMA):
// a phase of 19 commitments 12.7 ms against 11.25, coeff_to_extended into a fresh array 6.6 against 5.4.  One slot per chunk, always.

// One contiguous stretch of a transfer: `len` bytes between device and host addresses.  A lane gets a short list of them (one for a
// plain copy; several when a call moves many arrays at once: hm_copy_many_*), and walks it in chunks of at most one slot.
struct XferPiece {
  char* dev;
  char* host;
  size_t len;
};
struct PieceCursor {
  const XferPiece* p;
  size_t n, i, off;
  bool next(char** dev, char** host, size_t* len) {
    while (i < n && off >= p[i].len) {
      ++i;
      off = 0;
    }
    if (i >= n) return false;
    const size_t rem = p[i].len - off;
    *len = rem < kXferSlot ? rem : kXferSlot;
    *dev = p[i].dev + off;
    *host = p[i].host + off;
    off += *len;
    return true;
  }
};

static hipError_t lane_h2d(XferLane& l, int device, const XferPiece* pieces, size_t npieces) {
  hipError_t e = hipSetDevice(device);
  bool used[2] = {false, false};
  int slot = 0;
  l.t_wait_us = l.t_copy_us = l.t_issue_us = 0;
  PieceCursor cur{pieces, npieces, 0, 0};
  char *dv = nullptr, *hs = nullptr;
  size_t len = 0;
  while (e == hipSuccess && cur.next(&dv, &hs, &len)) {
    const double t0 = xfer_now_us();
    if (used[slot]) e = hipEventSynchronize(l.ev[slot]);          // the DMA that last read this slot
    if (e != hipSuccess) break;
    const double t1 = xfer_now_us();
    std::memcpy(l.pin[slot], hs, len);
    l.t_wait_us += t1 - t0;
    l.t_copy_us += xfer_now_us() - t1;
    e = hipMemcpyAsync(dv, l.pin[slot], len, hipMemcpyHostToDevice, l.stream);
    if (e == hipSuccess) e = hipEventRecord(l.ev[slot], l.stream);
    used[slot] = true;
    slot ^= 1;
  }
  const double t2 = xfer_now_us();
  const hipError_t s = hipStreamSynchronize(l.stream);
  l.t_wait_us += xfer_now_us() - t2;
  return e != hipSuccess ? e : s;
}

static hipError_t lane_d2h(XferLane& l, int device, const XferPiece* pieces, size_t npieces) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return e;
  PieceCursor cur{pieces, npieces, 0, 0};
  char* dst[2] = {nullptr, nullptr};
  size_t len[2] = {0, 0};
  bool pending[2] = {false, false};
  auto issue = [&](int slot) {                                    // false: nothing left to issue
    char* dv = nullptr;
    const double ti = xfer_now_us();
    if (!cur.next(&dv, &dst[slot], &len[slot])) return false;
    e = hipMemcpyAsync(l.pin[slot], dv, len[slot], hipMemcpyDeviceToHost, l.stream);
    if (e == hipSuccess) e = hipEventRecord(l.ev[slot], l.stream);
    l.t_issue_us += xfer_now_us() - ti;
    return true;
  };
  int slot = 0;
  l.t_wait_us = l.t_copy_us = l.t_issue_us = 0;
  pending[0] = issue(0);
  while (pending[slot] && e == hipSuccess) {
    pending[slot ^ 1] = issue(slot ^ 1);                          // the next DMA runs while this slot is copied out
    if (e != hipSuccess) break;
    const double t0 = xfer_now_us();
    e = hipEventSynchronize(l.ev[slot]);
    if (e != hipSuccess) break;
    const double t1 = xfer_now_us();
    std::memcpy(dst[slot], l.pin[slot], len[slot]);
    l.t_wait_us += t1 - t0;
    l.t_copy_us += xfer_now_us() - t1;
    pending[slot] = false;
    slot ^= 1;
  }
  if (e != hipSuccess) (void)hipStreamSynchronize(l.stream);     // nothing of ours stays in flight behind an error
  return e;
}

// which way a copy goes is decided by the RANGE and the policy alone (see the head of this file)
static bool xfer_goes_direct(const void* host, size_t bytes) {
  if (bytes < kXferDirectBelow) return true;
  const int policy = xfer_policy();
  if (policy == XFER_DIRECT) return true;
  if (policy == XFER_LANES) return false;
  return host_range_registered(host, bytes);
}

// The transfer of `segs` (one array, or the many of hm_copy_many_*) as ONE job of the lanes: the concatenation of the segments is cut
// into page-aligned shares, a lane each, so that forty 8 MiB columns move like one 320 MiB array (one spawn, no per-array fill and
// drain) instead of forty small transfers.  All segments direct by the range rule: plain hipMemcpy each.
static int xfer_run_many(DeviceCtx& ctx, bool up, const std::vector<XferPiece>& segs, const char* who) {
  size_t bytes = 0;
  bool all_direct = true;
  for (const XferPiece& sg : segs) {
    bytes += sg.len;
    if (sg.len && !xfer_goes_direct(sg.host, sg.len)) all_direct = false;
  }
  if (bytes == 0) return HM_OK;
  static const bool trace = std::getenv("HALO2_MI355X_XFER_TRACE") != nullptr;
  auto direct = [&]() {
    for (const XferPiece& sg : segs) {
      if (!sg.len) continue;
      const hipError_t e = up ? hipMemcpy(sg.dev, sg.host, sg.len, hipMemcpyHostToDevice) : hipMemcpy(sg.host, sg.dev, sg.len, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    }
    return (int)HM_OK;
  };
  if (all_direct) {
    const int rc = direct();
    if (rc != HM_OK) return rc;
    ctx.xfer.direct.fetch_add(1, std::memory_order_relaxed);
    if (trace) std::fprintf(stderr, "[halo2_mi355x] copy: %s %zu bytes in %zu array(s) direct (%s)\n", up ? "H2D" : "D2H", bytes, segs.size(), who);
    return HM_OK;
  }
  const double t_start = xfer_now_us();
  ctx.xfer.staged.fetch_add(1, std::memory_order_relaxed);
  if (trace) std::fprintf(stderr, "[halo2_mi355x] copy: %s %zu bytes in %zu array(s) through the lanes (%s)\n", up ? "H2D" : "D2H", bytes, segs.size(), who);
  std::lock_guard<std::mutex> lk(ctx.xfer.mu);                    // the lanes' slots belong to one transfer at a time
  int want = (int)((bytes + ((size_t)1 << 20) - 1) >> 20);        // a lane per MiB ...
  // ... FOUR at most by default (HALO2_MI355X_XFER_LANES = 1 .. 8 for experiments).  Round 5 ran eight; measured in round 6 on one 64 MiB
  // round trip (profiles/r06_lane_warmup.txt): 1 lane 2.2 + 3.8 ms up + down, 2 lanes 1.38 + 2.0, 4 lanes 1.38 + 1.38, 8 lanes 1.45 + 1.78
  // -- four already fill the link, and every lane beyond the second cost its first device-to-host copy a ~7 ms stall inside
  // hipMemcpyAsync (six of a process's first seven 64 MiB downloads took 8.3 ms instead of 1.8 with eight lanes; with four, after the
  // concurrent warm-up of xfer_prepare, none after the first call).
  if (want > HM_XFER_LANES) want = HM_XFER_LANES;
  static const int lanes_cap = [] { const char* v = std::getenv("HALO2_MI355X_XFER_LANES"); return v && *v ? std::atoi(v) : kXferDefaultLanes; }();
  if (lanes_cap >= 1 && want > lanes_cap) want = lanes_cap;
  const int lanes = xfer_prepare(ctx.xfer, want);
  if (lanes == 0) return direct();                                // no pinned memory to be had: the runtime's path still works
  const size_t share = (((bytes + lanes - 1) / lanes) + 4095) & ~(size_t)4095;     // page-aligned shares of the concatenation
  std::vector<XferPiece> work[HM_XFER_LANES];
  {
    size_t at = 0;                                                // position in the concatenation
    for (const XferPiece& sg : segs) {
      size_t done = 0;
      while (done < sg.len) {
        const size_t lane = (at / share) < (size_t)lanes ? at / share : (size_t)lanes - 1;
        const size_t room = (lane + 1) * share - at, take = sg.len - done < room ? sg.len - done : room;
        work[lane].push_back(XferPiece{sg.dev + done, sg.host + done, take});
        done += take;
        at += take;
      }
    }
  }
  hipError_t errs[HM_XFER_LANES];
  for (auto& e : errs) e = hipSuccess;
  const int device = ctx.device;
  auto part = [&](int i) {
    ctx.xfer.lanes[i].t_start_us = xfer_now_us();
    if (work[i].empty()) return;
    errs[i] = up ? lane_h2d(ctx.xfer.lanes[i], device, work[i].data(), work[i].size())
                 : lane_d2h(ctx.xfer.lanes[i], device, work[i].data(), work[i].size());
  };
  {
    JoinOnExit pool;
    int started = 1;                                              // lane 0 is the calling thread's
    for (; started < lanes; ++started)
      if (!spawn_or_false(pool, "xfer_spawn", [&part, started] { part(started); })) break;
    part(0);
    for (int i = started; i < lanes; ++i) part(i);                // lanes no thread could be had for
  }
  for (int i = 0; i < lanes; ++i)
    if (errs[i] != hipSuccess) return hm_fail(HM_ERR_HIP, std::string(who) + ": " + hipGetErrorString(errs[i]));
  if (trace) {                                                    // the slowest lane's split: waiting for its DMAs / copying on the host
    double wmax = 0, cmax = 0, imax = 0, smax = 0;
    for (int i = 0; i < lanes; ++i) {
      if (ctx.xfer.lanes[i].t_wait_us > wmax) wmax = ctx.xfer.lanes[i].t_wait_us;
      if (ctx.xfer.lanes[i].t_copy_us > cmax) cmax = ctx.xfer.lanes[i].t_copy_us;
      if (ctx.xfer.lanes[i].t_issue_us > imax) imax = ctx.xfer.lanes[i].t_issue_us;
      if (ctx.xfer.lanes[i].t_start_us - t_start > smax) smax = ctx.xfer.lanes[i].t_start_us - t_start;
    }
    std::fprintf(stderr, "[halo2_mi355x]   %d lanes, %.2f ms in all: latest lane start +%.2f ms, longest DMA wait %.2f ms, issue %.2f ms, host memcpy %.2f ms\n",
                 lanes, (xfer_now_us() - t_start) / 1e3, smax / 1e3, wmax / 1e3, imax / 1e3, cmax / 1e3);
  }
  return HM_OK;
}

static int xfer_run(DeviceCtx& ctx, bool up, void* dev, void* host, size_t bytes, const char* who) {
  if (bytes == 0) return HM_OK;
  const std::vector<XferPiece> one{XferPiece{(char*)dev, (char*)host, bytes}};
  return xfer_run_many(ctx, up, one, who);
}

int xfer_many(DeviceCtx& ctx, bool up, void* const* dev, void* const* host, const size_t* bytes, size_t count, const char* who) {
  std::vector<XferPiece> segs;
  segs.reserve(count);
  for (size_t i = 0; i < count; ++i) segs.push_back(XferPiece{(char*)dev[i], (char*)host[i], bytes[i]});
  return xfer_run_many(ctx, up, segs, who);
}

// 1: a copy of `bytes` from / to `host` would go through the lanes now; 0: straight to hipMemcpy
int xfer_mode(const void* host, size_t bytes) { return xfer_goes_direct(host, bytes) ? 0 : 1; }

int xfer_h2d(DeviceCtx& ctx, void* d_dst, const void* src, size_t bytes, const char* who) {
  return xfer_run(ctx, true, d_dst, const_cast<void*>(src), bytes, who);
}
int xfer_d2h(DeviceCtx& ctx, void* dst, const void* d_src, size_t bytes, const char* who) {
  return xfer_run(ctx, false, const_cast<void*>(d_src), dst, bytes, who);
}

}  // namespace hm
