// xfer.hip -- the host <-> device copies of the host-pointer entry points (hm_msm_bn256_g1*, hm_ntt_bn256_fr, hm_coeff_to_extended /
// hm_extended_to_coeff, the batch call's scalar uploads).
//
// Two paths.  DIRECT: hipMemcpy on the caller's pointers.  For pageable memory of this size the runtime pins the user's pages on the fly
// (a userptr registration with the kernel driver): the fastest path while it works -- 52 GB/s up, 49 GB/s down on this host, 1.63 ms
// for the 8 + 64 MiB of a k = 18 coeff_to_extended -- but it ties the library's latency to what the CALLER's allocator does around the
// calls: a prover maps and unmaps a 64 MiB Vec around every coeff_to_extended, and on some boxes of the pool (3 sessions of ~20 in
// round 5, never reproduced on demand) the copy that followed stalled for 7-27 ms: an 8 MiB upload measured at 22 ms instead of 0.16,
// the same 27 ms per call at k = 17 and k = 18, mmap itself slow (0.9 ms) in those sessions.  LANES: the library's own pinned staging,
// up to 8 lanes, each a host thread with its own stream and two 2 MiB pinned slots moving one contiguous share of the transfer (memcpy
// into a slot / DMA from the other, and the reverse).  Nothing of the caller's memory is ever registered with the driver, so nothing
// the caller maps or unmaps can invalidate what the GPU queues depend on -- at 0.7 ms more per 72 MiB on a healthy box (2.38 against
// 1.63 ms; profiles/r05_host_copies.txt).
//
// Policy (HALO2_MI355X_HOST_COPIES = auto | lanes | direct, default auto): direct until a copy has TWICE taken longer than
// 2 ms + bytes / 4 GB/s -- several times the healthy worst case, first-touch page faults included --, then lanes for the rest of the
// process; hm_get_stats reports the stalls seen and the mode.  Copies below 256 KiB always go straight to hipMemcpy (the runtime
// stages those itself).  xfer_prefault first-touches a destination the process has never written (the fresh Vec of a result) from
// helper threads while the transform is still running: a direct copy into untouched pages took its faults one by one (5 ms per 64 MiB).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "hm_internal.h"

namespace hm {

constexpr size_t kXferSlot = (size_t)2 << 20;
constexpr size_t kXferDirectBelow = (size_t)256 << 10;

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
constexpr int kXferStallsBeforeLanes = 2;

static double xfer_now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
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
      if (l.pin[s]) (void)hipHostFree(l.pin[s]);
      if (l.ev[s]) (void)hipEventDestroy(l.ev[s]);
      l.pin[s] = nullptr;
      l.ev[s] = nullptr;
    }
    if (l.stream) (void)hipStreamDestroy(l.stream);
    l.stream = nullptr;
  }
  ctx.xfer.ready = 0;
}

// lanes [0, want) exist afterwards, or fewer when pinned memory / streams cannot be had (at least one, else 0)
static int xfer_prepare(HostXfer& x, int want) {
  for (int i = x.ready; i < want; ++i) {
    XferLane& l = x.lanes[i];
    bool ok = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess;
    for (int s = 0; s < 2 && ok; ++s)
      ok = hipHostMalloc(&l.pin[s], kXferSlot, hipHostMallocDefault) == hipSuccess &&
           hipEventCreateWithFlags(&l.ev[s], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      for (int s = 0; s < 2; ++s) {
        if (l.pin[s]) (void)hipHostFree(l.pin[s]);
        if (l.ev[s]) (void)hipEventDestroy(l.ev[s]);
        l.pin[s] = nullptr;
        l.ev[s] = nullptr;
      }
      if (l.stream) (void)hipStreamDestroy(l.stream);
      l.stream = nullptr;
      break;
    }
    x.ready = i + 1;
  }
  return x.ready < want ? x.ready : want;
}

static hipError_t lane_h2d(XferLane& l, int device, char* d_dst, const char* src, size_t bytes) {
  hipError_t e = hipSetDevice(device);
  bool used[2] = {false, false};
  int slot = 0;
  for (size_t off = 0; off < bytes && e == hipSuccess; off += kXferSlot, slot ^= 1) {
    const size_t len = bytes - off < kXferSlot ? bytes - off : kXferSlot;
    if (used[slot]) e = hipEventSynchronize(l.ev[slot]);          // the DMA that last read this slot
    if (e != hipSuccess) break;
    std::memcpy(l.pin[slot], src + off, len);
    e = hipMemcpyAsync(d_dst + off, l.pin[slot], len, hipMemcpyHostToDevice, l.stream);
    if (e == hipSuccess) e = hipEventRecord(l.ev[slot], l.stream);
    used[slot] = true;
  }
  const hipError_t s = hipStreamSynchronize(l.stream);
  return e != hipSuccess ? e : s;
}

static hipError_t lane_d2h(XferLane& l, int device, char* dst, const char* d_src, size_t bytes) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess || bytes == 0) return e;
  auto len_at = [&](size_t off) { return bytes - off < kXferSlot ? bytes - off : kXferSlot; };
  auto issue = [&](int slot, size_t off) {
    hipError_t r = hipMemcpyAsync(l.pin[slot], d_src + off, len_at(off), hipMemcpyDeviceToHost, l.stream);
    if (r == hipSuccess) r = hipEventRecord(l.ev[slot], l.stream);
    return r;
  };
  int slot = 0;
  e = issue(0, 0);
  for (size_t off = 0; off < bytes && e == hipSuccess; off += kXferSlot, slot ^= 1) {
    const size_t next = off + kXferSlot;
    if (next < bytes) e = issue(slot ^ 1, next);                  // the next DMA runs while this slot is copied out
    const hipError_t w = hipEventSynchronize(l.ev[slot]);
    if (e == hipSuccess) e = w;
    if (e != hipSuccess) break;
    std::memcpy(dst + off, l.pin[slot], len_at(off));
  }
  if (e != hipSuccess) (void)hipStreamSynchronize(l.stream);     // nothing of ours stays in flight behind an error
  return e;
}

static int xfer_run(DeviceCtx& ctx, bool up, void* dev, void* host, size_t bytes, const char* who) {
  if (bytes == 0) return HM_OK;
  const int policy = xfer_policy();
  const bool lanes_now = policy == XFER_LANES || (policy == XFER_AUTO && ctx.xfer.stalls.load(std::memory_order_relaxed) >= kXferStallsBeforeLanes);
  if (bytes < kXferDirectBelow || !lanes_now) {
    const double t0 = xfer_now_us();
    const hipError_t e = up ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    const double dt = xfer_now_us() - t0;
    if (bytes >= kXferDirectBelow && dt > 2000.0 + (double)bytes / 4000.0) {      // 4 GB/s = 4 000 bytes per microsecond
      ctx.xfer.stalls.fetch_add(1, std::memory_order_relaxed);
      static const bool trace = std::getenv("HALO2_MI355X_XFER_TRACE") != nullptr;
      if (trace) std::fprintf(stderr, "[halo2_mi355x] stalled copy: %s %zu bytes in %.2f ms (%s)\n", up ? "H2D" : "D2H", bytes, dt / 1e3, who);
    }
    return HM_OK;
  }
  std::lock_guard<std::mutex> lk(ctx.xfer.mu);                    // the lanes' slots belong to one transfer at a time
  int want = (int)((bytes + ((size_t)1 << 20) - 1) >> 20);        // a lane per MiB, eight at most
  if (want > HM_XFER_LANES) want = HM_XFER_LANES;
  const int lanes = xfer_prepare(ctx.xfer, want);
  if (lanes == 0) {                                               // no pinned memory to be had: the runtime's path still works
    const hipError_t e = up ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
    return HM_OK;
  }
  const size_t share = (((bytes + lanes - 1) / lanes) + 4095) & ~(size_t)4095;     // page-aligned shares
  hipError_t errs[HM_XFER_LANES];
  for (auto& e : errs) e = hipSuccess;
  const int device = ctx.device;
  auto part = [&](int i) {
    const size_t lo = (size_t)i * share;
    if (lo >= bytes) return;
    const size_t len = bytes - lo < share ? bytes - lo : share;
    errs[i] = up ? lane_h2d(ctx.xfer.lanes[i], device, (char*)dev + lo, (const char*)host + lo, len)
                 : lane_d2h(ctx.xfer.lanes[i], device, (char*)host + lo, (const char*)dev + lo, len);
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
  return HM_OK;
}

int xfer_mode(DeviceCtx& ctx) {
  const int policy = xfer_policy();
  return policy == XFER_LANES || (policy == XFER_AUTO && ctx.xfer.stalls.load(std::memory_order_relaxed) >= kXferStallsBeforeLanes) ? 1 : 0;
}

int xfer_h2d(DeviceCtx& ctx, void* d_dst, const void* src, size_t bytes, const char* who) {
  return xfer_run(ctx, true, d_dst, const_cast<void*>(src), bytes, who);
}
int xfer_d2h(DeviceCtx& ctx, void* dst, const void* d_src, size_t bytes, const char* who) {
  return xfer_run(ctx, false, const_cast<void*>(d_src), dst, bytes, who);
}

}  // namespace hm
