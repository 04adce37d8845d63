// capi.hip -- the extern "C" boundary declared in include/halo2_mi355x.h.
// Host pointers in, host results out; device staging, base-set caching and locking live here.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <atomic>
#include <chrono>
#include <string>
#include <map>
#include <random>
#include <stdexcept>
#include <thread>
#include <vector>

#include "hm_internal.h"
#include "host_fr.h"

namespace hm {

// The commitments of a prover phase run eight at a time on eight streams (hm_msm_batch_bn256_g1_dev, hm_msm_submit_dev);
// the HIP runtime maps streams onto 4 hardware queues unless told otherwise, which caps the kernels actually in flight
// (measured at k = 18: a sparse column costs 0.215 ms with 4 queues, 0.136 ms with 16; a dense one 0.654 / 0.578 ms).
// The variable is read when the runtime initialises, so it is set -- without overriding the caller's own choice -- when
// this library is loaded; a process whose runtime is already up (e.g. torch touched the GPU first) keeps what it has.
namespace {
struct HwQueueDefault {
  HwQueueDefault() { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }
} g_hw_queue_default;
}  // namespace

static thread_local std::string g_last_error;
static std::mutex g_ctx_mu;
static std::map<int, std::unique_ptr<DeviceCtx>> g_ctx;
static std::vector<int> g_msm_devices;             // hm_set_msm_devices; empty = the calling thread's device
static std::atomic<int> g_host_base_cache{1};      // hm_set_host_base_cache
static std::atomic<uint32_t> g_fixed_base_from_log{[] {      // hm_set_fixed_base_threshold
  const char* v = std::getenv("HALO2_MI355X_FIXED_BASE_FROM_LOG");
  return (uint32_t)(v && *v ? std::atoi(v) : 17);
}()};

int hm_fail(int code, const std::string& what) {
  g_last_error = what;
  return code;
}
const std::string& hm_last_error_string() { return g_last_error; }
std::vector<int> msm_device_list() {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  return g_msm_devices;
}

// The handler of HM_API_CATCH: must not throw itself (the message is built inside its own try; when even that
// fails -- no memory for a short string -- the code alone goes back and the message stays empty).
int hm_guard_fail(const char* entry, const char* what) noexcept {
  try {
    g_last_error.assign(entry);
    g_last_error.append(": internal error caught at the C boundary: ");
    g_last_error.append(what ? what : "unknown exception");
  } catch (...) {
    g_last_error.clear();
  }
  return HM_ERR_INTERNAL;
}

#ifdef HM_FAULT_INJECTION
std::mutex g_fault_mu;
std::string g_fault_name;
long g_fault_after = -1;        // < 0: disarmed
void hm_fault_point(const char* point) {
  std::lock_guard<std::mutex> lk(g_fault_mu);
  if (g_fault_after < 0 || g_fault_name != point) return;
  if (g_fault_after-- == 0) throw std::runtime_error(std::string("injected fault at ") + point);
}
#endif

void msm_set_window_override(int c);  // msm.hip
void msm_set_phase_timing(int mode);   // msm.hip

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// device_ms: the hipEvent span of the launch chain that carried this MSM -- passed ONCE per chain (a grouped chain
// carries up to eight MSMs: the others pass 0).  Chains in flight overlap, so msm_device_us is a sum of spans, not a
// wall time: hm_stats documents it as such.
static void count_msm(DeviceCtx& ctx, size_t n, double device_ms) {
  uint32_t lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  ctx.calls.msm_calls += 1;
  ctx.calls.msm_points += n;
  ctx.calls.msm_by_log[lg & 31] += 1;
  ctx.calls.msm_device_us += device_ms * 1e3;
}

// Released buffers are parked for the next registration of that size instead of hipFree'd (which waits for the whole
// device): at most four sets and at most kParkedBytesMax in total -- a caller that registers a plain set per call (the
// tensor form of best_multiexp) recycles its 64 B/point buffer for ever, while a released fixed-base table (W copies:
// 12 GiB at 2^24 points) goes back to the allocator instead of sitting in the park.
constexpr size_t kParkedBytesMax = (size_t)2 << 30;
static size_t parked_bytes(const DeviceCtx& ctx) {
  size_t t = 0;
  for (const auto& f : ctx.free_bases) t += f.xy_bytes + f.inf_bytes;
  return t;
}
static void free_bases_entry(DeviceCtx& ctx, BasesEntry& b) {
  // no kernel reads these buffers any more (synchronous calls have returned, tickets were awaited)
  if (ctx.free_bases.size() < 4 && parked_bytes(ctx) + b.xy_bytes + b.inf_bytes <= kParkedBytesMax) {
    ctx.free_bases.push_back(FreeBases{b.d_xy, b.d_inf, b.xy_bytes, b.inf_bytes});
  } else {
    if (b.d_xy) (void)hipFree(b.d_xy);
    if (b.d_inf) (void)hipFree(b.d_inf);
  }
  b.d_xy = nullptr;
  b.d_inf = nullptr;
}
// hipFree every parked buffer (an allocation failed: the memory may be sitting here); true when there was anything
static bool drop_parked_bases(DeviceCtx& ctx) {
  const bool any = !ctx.free_bases.empty();
  for (auto& f : ctx.free_bases) {
    if (f.d_xy) (void)hipFree(f.d_xy);
    if (f.d_inf) (void)hipFree(f.d_inf);
  }
  ctx.free_bases.clear();
  return any;
}

DeviceCtx* ctx_for_current_device() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    hm_fail(HM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    return nullptr;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    hm_fail(HM_ERR_NO_DEVICE, "hipGetDevice failed");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  auto it = g_ctx.find(dev);
  if (it == g_ctx.end()) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      hm_fail(HM_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
      return nullptr;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
      hm_fail(HM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
      return nullptr;
    }
    auto c = std::make_unique<DeviceCtx>();
    c->device = dev;
    it = g_ctx.emplace(dev, std::move(c)).first;
  }
  return it->second.get();
}

static BasesEntry* find_bases(DeviceCtx& ctx, uint64_t handle) {
  for (auto& b : ctx.bases)
    if (b.handle == handle) return &b;
  return nullptr;
}

// Which copy of the points a registration stores: DEFAULT = the table from hm_set_fixed_base_threshold's size on (falls back to
// the plain layout when W copies do not fit), TABLE = the caller asked for it (no fallback), PLAIN = one copy, never a table
// (transient sets: the table build costs ten MSMs).
enum class BaseLayout { DEFAULT, TABLE, PLAIN };
static std::atomic<uint64_t> g_default_table_dropped{0};   // DEFAULT registrations that fell back to the plain layout (hm_get_bases_info)

static int register_from_device(DeviceCtx& ctx, const uint32_t* d_ext, size_t n, hipStream_t stream, uint64_t* out_handle,
                                BaseLayout layout = BaseLayout::DEFAULT) {
  BasesEntry e;
  e.n = n;
  bool precomp = layout == BaseLayout::TABLE, by_default = false;
  if (layout == BaseLayout::DEFAULT) {   // the fixed-base table by default from the size where it pays (hm_set_fixed_base_threshold)
    const uint32_t from = g_fixed_base_from_log.load(std::memory_order_relaxed);
    if (from != 0 && from < 40 && n >= ((size_t)1 << from)) precomp = by_default = true;
  }
  if (precomp && n >= 256) {   // tiny sets gain nothing from a shared bucket set
    e.pc_c = msm_precomp_window(n);
    e.pc_W = (255 + e.pc_c - 1) / e.pc_c;
    if ((uint64_t)n * e.pc_W >= (1ull << 31)) { e.pc_c = 0; e.pc_W = 0; }
  }
  const size_t copies = e.pc_c ? e.pc_W : 1;
  const size_t xy_bytes = n ? n * 64 * copies : 64, inf_bytes = n ? n : 1;
  // buffers of a released set of the same size are reused (a caller that registers per call -- the
  // tensor form of best_multiexp -- then never reaches hipMalloc / hipFree and their device-wide waits)
  for (size_t i = 0; i < ctx.free_bases.size(); ++i) {
    if (ctx.free_bases[i].xy_bytes == xy_bytes && ctx.free_bases[i].inf_bytes == inf_bytes) {
      e.d_xy = ctx.free_bases[i].d_xy;
      e.d_inf = ctx.free_bases[i].d_inf;
      ctx.free_bases.erase(ctx.free_bases.begin() + i);
      break;
    }
  }
  if (!e.d_xy) {
    hipError_t err = hipMalloc((void**)&e.d_xy, xy_bytes);
    if (err != hipSuccess && drop_parked_bases(ctx)) {     // the memory may be parked: give it back and try once more
      (void)hipGetLastError();
      err = hipMalloc((void**)&e.d_xy, xy_bytes);
    }
    if (err != hipSuccess) {
      (void)hipGetLastError();
      e.d_xy = nullptr;
      // no room for W copies: a caller of the plain entry point asked for a base set, not for the table
      if (by_default && e.pc_c) {
        g_default_table_dropped.fetch_add(1, std::memory_order_relaxed);
        return register_from_device(ctx, d_ext, n, stream, out_handle, BaseLayout::PLAIN);
      }
      return hm_fail(HM_ERR_HIP, "register bases: allocation failed");
    }
    if (hipMalloc((void**)&e.d_inf, inf_bytes) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(e.d_xy);
      return hm_fail(HM_ERR_HIP, "register bases: allocation failed");
    }
  }
  e.xy_bytes = xy_bytes;
  e.inf_bytes = inf_bytes;
  int rc = msm_convert_bases(d_ext, e.d_xy, e.d_inf, n, stream);
  if (rc == HM_OK && e.pc_c) rc = msm_precompute(e.d_xy, e.d_inf, n, e.pc_c, e.pc_W, stream);
  if (rc != HM_OK) {
    (void)hipStreamSynchronize(stream);
    (void)hipFree(e.d_xy);
    (void)hipFree(e.d_inf);
    return rc;
  }
  e.handle = ctx.next_handle++;
  ctx.bases.push_back(e);
  *out_handle = e.handle;
  return HM_OK;
}

static int jac_to_affine_out(const uint64_t jac[12], int is_id, uint64_t out_xy[8], int* out_is_identity) {
  if (is_id) {
    std::memset(out_xy, 0, 64);
  } else {
    std::memcpy(out_xy, jac, 64);  // msm_run returns (x, y, 1): already affine
  }
  if (out_is_identity) *out_is_identity = is_id;
  return HM_OK;
}

// one device: upload the scalars, run against the registered set, Jacobian (x, y, 1) / zeros out
int msm_h_local(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n, uint64_t jac[12], int* is_id) {
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  BasesEntry* b = find_bases(*ctx, handle);
  if (!b) return hm_fail(HM_ERR_NOT_FOUND, "hm_msm_bn256_g1_h: unknown base handle");
  if (offset > b->n || n > b->n - offset) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1_h: offset + n exceeds the base set");
  void* d_s = ctx->io.ensure(n ? n * 32 : 32);
  if (!d_s) return hm_fail(HM_ERR_HIP, "hm_msm_bn256_g1_h: staging allocation failed");
  const double t0 = now_us();
  {
    const int rc = xfer_h2d(*ctx, d_s, scalars, n * 32, "hm_msm_bn256_g1_h: scalar upload");
    if (rc != HM_OK) return rc;
  }
  ctx->calls.msm_h2d_us += now_us() - t0;
  ctx->calls.h2d_bytes += n * 32;
  const uint32_t pc = (offset == 0 && n == b->n) ? b->pc_c : 0u;
  int rc = msm_run(*ctx, (const uint32_t*)d_s, b->d_xy + offset * 16, b->d_inf + offset, n, pc, jac, is_id, nullptr);
  if (rc != HM_OK) return rc;
  count_msm(*ctx, n, ctx->last_msm.t_total_ms);
  return HM_OK;
}

}  // namespace hm

using namespace hm;

extern "C" {

int hm_device_count(void) {      // no allocation, nothing that throws: "never fails" stays literal
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

int hm_set_device(int device) try {
  int count = hm_device_count();
  if (count <= 0) return hm_fail(HM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= count) return hm_fail(HM_ERR_BAD_ARG, "hm_set_device: device index out of range");
  HM_HIP_CHECK(hipSetDevice(device));
  return HM_OK;
} HM_API_CATCH("hm_set_device")

const char* hm_last_error(void) { return g_last_error.c_str(); }
const char* hm_version(void) { return "halo2_mi355x 0.3 (gfx950; ff29 field layer)"; }

int hm_shutdown(void) try {
  int dev = 0;
  if (hm_device_count() <= 0 || hipGetDevice(&dev) != hipSuccess) return HM_OK;
  multi_release_touching(dev);     // multi-device sets with a part here go first (their parts elsewhere are released too)
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  auto it = g_ctx.find(dev);
  if (it == g_ctx.end()) return HM_OK;
  DeviceCtx& c = *it->second;
  std::lock_guard<std::mutex> lk2(c.mu);
  (void)hipDeviceSynchronize();
  for (auto& t : c.ntt_tables) ntt_tables_release(*t);
  c.ntt_tables.clear();
  c.ntt_table_bytes = 0;
  xfer_release(c);
  coset_tables_release(c);
  for (auto* list : {&c.bases, &c.zombie_bases}) {
    for (auto& b : *list) {
      if (b.d_xy) (void)hipFree(b.d_xy);
      if (b.d_inf) (void)hipFree(b.d_inf);
    }
    list->clear();
  }
  for (auto& f : c.free_bases) {
    if (f.d_xy) (void)hipFree(f.d_xy);
    if (f.d_inf) (void)hipFree(f.d_inf);
  }
  c.free_bases.clear();
  for (auto& g : c.graphs) graph_release(*g);
  c.graphs.clear();
  if (c.batch_streams_ready) {
    for (auto& st : c.batch_streams) (void)hipStreamDestroy(st);
    (void)hipEventDestroy(c.batch_event);
    c.batch_streams_ready = false;
  }
  for (auto& b : c.batch_io) b.release();
  {
    std::lock_guard<std::mutex> lk3(c.live_mu);
    c.live_io.release();
  }
  c.io.release(); c.io_bases.release(); c.conv_bases.release(); c.conv_inf.release();
  c.cached_host_n = 0;
  c.cached_xy = nullptr;
  for (auto& a : c.aux) {
    a.scratch.release();
    a.table.release();
    a.args.release();
    a.work.release();
    if (a.done) (void)hipEventDestroy(a.done);
    a = AuxSlot{};
  }
  for (auto& sl : c.msm_slots) {
    sl.ws.release();
    sl.live_ptr = nullptr;
    sl.busy = false;
    if (sl.h_land) { (void)hipHostFree(sl.h_land); sl.h_land = nullptr; }
    if (sl.ev_ready) { for (auto& e : sl.ev) (void)hipEventDestroy(e); sl.ev_ready = false; }
  }
  return HM_OK;
} HM_API_CATCH("hm_shutdown")

int hm_set_fixed_base_threshold(uint32_t log2_n) try {
  if (log2_n != 0 && (log2_n < 8 || log2_n > 31)) return hm_fail(HM_ERR_BAD_ARG, "hm_set_fixed_base_threshold: 0 or a size in [2^8, 2^31]");
  g_fixed_base_from_log.store(log2_n, std::memory_order_relaxed);
  return HM_OK;
} HM_API_CATCH("hm_set_fixed_base_threshold")

int hm_set_host_base_cache(int enable) try {
  g_host_base_cache.store(enable != 0, std::memory_order_relaxed);
  return HM_OK;
} HM_API_CATCH("hm_set_host_base_cache")

int hm_device_malloc(size_t bytes, void** d_out) try {
  if (!d_out) return hm_fail(HM_ERR_BAD_ARG, "hm_device_malloc: null output");
  *d_out = nullptr;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  if (bytes == 0) return HM_OK;
  void* p = nullptr;
  hipError_t e = hipSuccess;
#ifdef HM_FAULT_INJECTION      // test build: an armed "device_malloc_oom" makes the FIRST attempt fail like an exhausted device
  try {
    hm_fault_point("device_malloc_oom");
    e = hipMalloc(&p, bytes);
  } catch (const std::exception&) {
    e = hipErrorOutOfMemory;
  }
#else
  e = hipMalloc(&p, bytes);
#endif
  if (e != hipSuccess) {                                  // the library's own caches give back what they can: parked base sets first,
    (void)hipGetLastError();                              // then every cached twiddle / coset table (up to 3 GiB, rebuilt on demand)
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (drop_parked_bases(*ctx)) e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (ntt_caches_give_back(*ctx) != 0) e = hipMalloc(&p, bytes);
    }
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return hm_fail(HM_ERR_HIP, std::string("hm_device_malloc: ") + hipGetErrorString(e));
  }
  *d_out = p;
  return HM_OK;
} HM_API_CATCH("hm_device_malloc")

int hm_device_free(void* d_ptr) try {
  if (!d_ptr) return HM_OK;
  if (!ctx_for_current_device()) return HM_ERR_NO_DEVICE;
  HM_HIP_CHECK(hipFree(d_ptr));
  return HM_OK;
} HM_API_CATCH("hm_device_free")

int hm_copy_to_device(void* d_dst, const void* src, size_t bytes) try {
  if (bytes && (!d_dst || !src)) return hm_fail(HM_ERR_BAD_ARG, "hm_copy_to_device: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  // the same ordering whichever way the bytes travel (the lanes' streams are non-blocking; hipMemcpy alone would wait for the
  // default stream and every blocking stream): behind everything queued on the default stream / the blocking streams
  HM_HIP_CHECK(hipStreamSynchronize(nullptr));
  const int rc = xfer_h2d(*ctx, d_dst, src, bytes, "hm_copy_to_device");
  if (rc == HM_OK) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->calls.h2d_bytes += bytes;
  }
  return rc;
} HM_API_CATCH("hm_copy_to_device")

int hm_copy_to_host(void* dst, const void* d_src, size_t bytes) try {
  if (bytes && (!dst || !d_src)) return hm_fail(HM_ERR_BAD_ARG, "hm_copy_to_host: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  HM_HIP_CHECK(hipStreamSynchronize(nullptr));              // (see hm_copy_to_device)
  if (xfer_d2h(*ctx, dst, d_src, bytes, "hm_copy_to_host") != HM_OK)
    return hm_fail(HM_ERR_PARTIAL_OUTPUT, "hm_copy_to_host: the destination is partly written: " + hm_last_error_string());
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->calls.d2h_bytes += bytes;
  return HM_OK;
} HM_API_CATCH("hm_copy_to_host")

static int copy_many(const char* who, bool up, void* const* dev, void* const* host, const size_t* bytes, size_t count) {
  if (count && (!dev || !host || !bytes)) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null argument");
  size_t total = 0;
  for (size_t i = 0; i < count; ++i) {
    if (bytes[i] && (!dev[i] || !host[i])) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null array");
    total += bytes[i];
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  if (total == 0) return HM_OK;
  HM_HIP_CHECK(hipStreamSynchronize(nullptr));              // the same ordering as hm_copy_to_device / _to_host
  const int rc = xfer_many(*ctx, up, dev, host, bytes, count, who);
  if (rc != HM_OK) return up ? rc : hm_fail(HM_ERR_PARTIAL_OUTPUT, std::string(who) + ": the destinations are partly written: " + hm_last_error_string());
  std::lock_guard<std::mutex> lk(ctx->mu);
  (up ? ctx->calls.h2d_bytes : ctx->calls.d2h_bytes) += total;
  return HM_OK;
}

int hm_copy_many_to_device(void* const* d_dsts, const void* const* srcs, const size_t* bytes, size_t count) try {
  return copy_many("hm_copy_many_to_device", true, d_dsts, const_cast<void* const*>(srcs), bytes, count);
} HM_API_CATCH("hm_copy_many_to_device")

int hm_copy_many_to_host(void* const* dsts, const void* const* d_srcs, const size_t* bytes, size_t count) try {
  return copy_many("hm_copy_many_to_host", false, const_cast<void* const*>(d_srcs), dsts, bytes, count);
} HM_API_CATCH("hm_copy_many_to_host")

int hm_host_register(const void* p, size_t bytes) try {
  if (!ctx_for_current_device()) return HM_ERR_NO_DEVICE;
  return xfer_host_register(p, bytes);
} HM_API_CATCH("hm_host_register")

int hm_host_unregister(const void* p) try {
  if (!ctx_for_current_device()) return HM_ERR_NO_DEVICE;
  return xfer_host_unregister(p);
} HM_API_CATCH("hm_host_unregister")

int hm_device_synchronize(void) try {
  if (!ctx_for_current_device()) return HM_ERR_NO_DEVICE;
  HM_HIP_CHECK(hipDeviceSynchronize());
  return HM_OK;
} HM_API_CATCH("hm_device_synchronize")

int hm_set_host_copies(int mode) try {
  if (xfer_set_policy(mode) != 0) return hm_fail(HM_ERR_BAD_ARG, "hm_set_host_copies: mode must be 0 (auto), 1 (lanes) or 2 (direct)");
  return HM_OK;
} HM_API_CATCH("hm_set_host_copies")

int hm_msm_set_window(int c) try {
  if (c != 0 && (c < 2 || c > 22)) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_set_window: c must be 0 or in [2, 22]");
  msm_set_window_override(c);
  return HM_OK;
} HM_API_CATCH("hm_msm_set_window")

int hm_msm_set_phase_timing(int mode) try {
  msm_set_phase_timing(mode);
  return HM_OK;
} HM_API_CATCH("hm_msm_set_phase_timing")

// ---- MSM -------------------------------------------------------------------------------------

// the six registration entry points: {host array, device array} x {default, table, plain layout}
static int register_entry(const char* who, const uint64_t* bases_host, const void* d_bases, size_t n, void* stream, BaseLayout layout,
                          uint64_t* out_handle) {
  if (!out_handle || (n && !bases_host && !d_bases)) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null argument");
  {
    const std::vector<int> devs = multi_worker_flag() ? std::vector<int>() : msm_device_list();
    if (devs.size() >= 2) return multi_register(bases_host, d_bases, n, stream, (int)layout, devs, out_handle);
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const uint32_t* d_ext = (const uint32_t*)d_bases;
  hipStream_t st = (hipStream_t)stream;
  if (bases_host) {
    void* stage = ctx->io_bases.ensure(n ? n * 64 : 64);
    if (!stage) return hm_fail(HM_ERR_HIP, std::string(who) + ": staging allocation failed");
    {
      const int rc = xfer_h2d(*ctx, stage, bases_host, n * 64, who);
      if (rc != HM_OK) return rc;
    }
    d_ext = (const uint32_t*)stage;
    st = nullptr;
  }
  const int rc = register_from_device(*ctx, d_ext, n, st, out_handle, layout);
  if (rc != HM_OK) return rc;
  HM_HIP_CHECK(hipStreamSynchronize(st));
  return HM_OK;
}

int hm_register_bases(const uint64_t* bases, size_t n, uint64_t* out_handle) try {
  return register_entry("hm_register_bases", bases, nullptr, n, nullptr, BaseLayout::DEFAULT, out_handle);
} HM_API_CATCH("hm_register_bases")

int hm_register_bases_dev(const void* d_bases, size_t n, void* stream, uint64_t* out_handle) try {
  return register_entry("hm_register_bases_dev", nullptr, d_bases, n, stream, BaseLayout::DEFAULT, out_handle);
} HM_API_CATCH("hm_register_bases_dev")

int hm_register_bases_precomp(const uint64_t* bases, size_t n, uint64_t* out_handle) try {
  return register_entry("hm_register_bases_precomp", bases, nullptr, n, nullptr, BaseLayout::TABLE, out_handle);
} HM_API_CATCH("hm_register_bases_precomp")

int hm_register_bases_precomp_dev(const void* d_bases, size_t n, void* stream, uint64_t* out_handle) try {
  return register_entry("hm_register_bases_precomp_dev", nullptr, d_bases, n, stream, BaseLayout::TABLE, out_handle);
} HM_API_CATCH("hm_register_bases_precomp_dev")

int hm_register_bases_plain(const uint64_t* bases, size_t n, uint64_t* out_handle) try {
  return register_entry("hm_register_bases_plain", bases, nullptr, n, nullptr, BaseLayout::PLAIN, out_handle);
} HM_API_CATCH("hm_register_bases_plain")

int hm_register_bases_plain_dev(const void* d_bases, size_t n, void* stream, uint64_t* out_handle) try {
  return register_entry("hm_register_bases_plain_dev", nullptr, d_bases, n, stream, BaseLayout::PLAIN, out_handle);
} HM_API_CATCH("hm_register_bases_plain_dev")

int hm_get_bases_info(uint64_t handle, hm_bases_info* out) try {
  if (!out) return hm_fail(HM_ERR_BAD_ARG, "hm_get_bases_info: null output");
  std::memset(out, 0, sizeof *out);
  out->default_tables_dropped = g_default_table_dropped.load(std::memory_order_relaxed);
  if (is_multi_handle(handle)) return multi_bases_info(handle, out);
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const BasesEntry* b = find_bases(*ctx, handle);
  if (!b) return hm_fail(HM_ERR_NOT_FOUND, "hm_get_bases_info: unknown handle");
  out->n = b->n;
  out->table_windows = b->pc_W;
  out->table_window_bits = b->pc_c;
  out->device_bytes = b->xy_bytes + b->inf_bytes;
  out->devices = 1;
  out->parked_bytes = parked_bytes(*ctx);
  return HM_OK;
} HM_API_CATCH("hm_get_bases_info")

int hm_release_bases(uint64_t handle) try {
  if (is_multi_handle(handle)) return multi_release(handle);
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (size_t i = 0; i < ctx->bases.size(); ++i) {
    if (ctx->bases[i].handle == handle) {
      // Every synchronous user has returned by now; only an un-awaited ticket can still read the set.
      // Then the buffers outlive the handle until that ticket's hm_msm_wait -- never a device-wide wait.
      bool in_flight = false;
      for (int k = 1; k < HM_MSM_SLOTS; ++k)
        if (ctx->msm_slots[k].busy && ctx->msm_slots[k].bases_handle == handle) in_flight = true;
      if (in_flight) ctx->zombie_bases.push_back(ctx->bases[i]);
      else free_bases_entry(*ctx, ctx->bases[i]);
      ctx->bases.erase(ctx->bases.begin() + i);
      return HM_OK;
    }
  }
  return hm_fail(HM_ERR_NOT_FOUND, "hm_release_bases: unknown handle");
} HM_API_CATCH("hm_release_bases")

int hm_msm_bn256_g1_dev(uint64_t handle, size_t offset, const void* d_scalars, size_t n, void* stream, uint64_t out_xyz[12]) try {
  if (!out_xyz || (n && !d_scalars)) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1_dev: null argument");
  if (is_multi_handle(handle)) {
    int id = 0;
    return multi_msm(handle, offset, d_scalars, false, n, stream, out_xyz, &id);
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  BasesEntry* b = find_bases(*ctx, handle);
  if (!b) return hm_fail(HM_ERR_NOT_FOUND, "hm_msm_bn256_g1_dev: unknown base handle");
  if (offset > b->n || n > b->n - offset) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1_dev: offset + n exceeds the base set");
  int is_id = 0;
  const uint32_t pc = (offset == 0 && n == b->n) ? b->pc_c : 0u;   // the table only fits whole-set calls
  const int rc = msm_run(*ctx, (const uint32_t*)d_scalars, b->d_xy + offset * 16, b->d_inf + offset, n, pc, out_xyz, &is_id,
                         (hipStream_t)stream);
  if (rc == HM_OK) count_msm(*ctx, n, ctx->last_msm.t_total_ms);
  return rc;
} HM_API_CATCH("hm_msm_bn256_g1_dev")

// One ticket = one launch chain = `group` MSMs over the same base range (group > 1 only where the five-launch plan applies).
// use_table = false: run on the plain copy of the points even when the set carries a fixed-base table (the five-launch
// plan with its block compaction is what a SPARSE column of a prover-sized phase wants)
static int submit_chain(DeviceCtx* ctx, uint64_t handle, size_t offset, const void* const* d_scalars_list, uint32_t group, size_t n,
                        void* stream, uint64_t* out_ticket, const char* who, bool* all_busy = nullptr, bool use_table = true,
                        size_t live_rows = 0) {
  if (all_busy) *all_busy = false;
  std::lock_guard<std::mutex> lk(ctx->mu);
  BasesEntry* b = find_bases(*ctx, handle);
  if (!b) return hm_fail(HM_ERR_NOT_FOUND, std::string(who) + ": unknown base handle");
  if (offset > b->n || n > b->n - offset) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": offset + n exceeds the base set");
  int slot = -1;
  for (int i = 1; i < HM_MSM_SLOTS; ++i)        // slot 0 stays free for the synchronous calls
    if (!ctx->msm_slots[i].busy) { slot = i; break; }
  if (slot < 0) {
    if (all_busy) *all_busy = true;             // the batch call retries: another thread's tickets hold the slots
    return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": every slot is in flight; hm_msm_wait one first");
  }
  const uint32_t pc = (use_table && offset == 0 && n == b->n) ? b->pc_c : 0u;
  int rc;
  if (group == 1 && live_rows == 0) {
    ctx->msm_slots[slot].group = 1;
    rc = msm_enqueue(*ctx, slot, (const uint32_t*)d_scalars_list[0], b->d_xy + offset * 16, b->d_inf + offset, n, pc, (hipStream_t)stream);
  } else if (use_table && live_rows == 0) {     // dense columns of a phase on the table: one chain of the general pipeline for all of them
    if (pc == 0 || group > msm_table_group_max(n, pc)) return hm_fail(HM_ERR_INTERNAL, std::string(who) + ": a dense group needs the set's table");
    rc = msm_enqueue_table_group(*ctx, slot, reinterpret_cast<const uint32_t* const*>(d_scalars_list), group, b->d_xy, b->d_inf, n, pc,
                                 (hipStream_t)stream);
  } else {                                      // the five-launch plan, sized for the rows known to survive (a lone sparse column too)
    rc = msm_enqueue_group(*ctx, slot, reinterpret_cast<const uint32_t* const*>(d_scalars_list), group, b->d_xy + offset * 16,
                           b->d_inf + offset, n, (hipStream_t)stream, live_rows);
  }
  if (rc != HM_OK) return rc;
  ctx->msm_slots[slot].busy = true;
  ctx->msm_slots[slot].bases_handle = handle;
  ctx->msm_slots[slot].ticket = ctx->next_ticket++;
  *out_ticket = ctx->msm_slots[slot].ticket;
  return HM_OK;
}

int hm_msm_submit_dev(uint64_t handle, size_t offset, const void* d_scalars, size_t n, void* stream, uint64_t* out_ticket) try {
  if (!out_ticket || (n && !d_scalars)) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_submit_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  if (is_multi_handle(handle)) {      // tickets belong to one device: a replicated set is used through its copy here
    const int rc = multi_local_part(handle, ctx->device, &handle);
    if (rc != HM_OK) return rc;
  }
  return submit_chain(ctx, handle, offset, &d_scalars, 1, n, stream, out_ticket, "hm_msm_submit_dev");
} HM_API_CATCH("hm_msm_submit_dev")

static int wait_chain(DeviceCtx* ctx, uint64_t ticket, uint64_t* out_xyz, uint32_t capacity);
// a ticket whose wait failed half-way (an exception): let its chain drain and give the slot back, whatever state it is in
static void abandon_ticket(DeviceCtx* ctx, uint64_t ticket) noexcept {
  try {
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 1; i < HM_MSM_SLOTS; ++i) {
      MsmSlot& sl = ctx->msm_slots[i];
      if (!sl.busy || sl.ticket != ticket) continue;
      if (sl.n != 0 && sl.ev_ready) (void)hipEventSynchronize(sl.ev[4]);
      sl.awaiting = false;
      sl.busy = false;
      sl.live_ptr = nullptr;
    }
  } catch (...) {
  }
}

// The commitments of one prover phase in one call: `count` scalar arrays against the same base range, kept eight in
// flight on the library's own streams (created on first use), results in call order.  What a caller of
// hm_msm_submit_dev / hm_msm_wait would write by hand.
static int msm_batch_impl(DeviceCtx* ctx, uint64_t handle, size_t offset, const void* const* d_scalars, bool from_host, size_t n,
                          size_t count, void* stream, uint64_t* out_xyz) {
  constexpr int kLanes = HM_MSM_SLOTS - 1;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->batch_streams_ready) {
      for (int i = 0; i < kLanes; ++i) HM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->batch_streams[i], hipStreamNonBlocking));
      HM_HIP_CHECK(hipEventCreateWithFlags(&ctx->batch_event, hipEventDisableTiming));
      ctx->batch_streams_ready = true;
    }
    // the scalar arrays are produced on the caller's stream: every lane starts behind it
    HM_HIP_CHECK(hipEventRecord(ctx->batch_event, (hipStream_t)stream));
    for (int i = 0; i < kLanes; ++i) HM_HIP_CHECK(hipStreamWaitEvent(ctx->batch_streams[i], ctx->batch_event, 0));
  }
  // Submitting an MSM (launches, event records) and finishing one (the wait, the host fold over its window sums) each
  // cost tens of microseconds of host time, and a prover-sized MSM alone is bound by launch gaps and chain depth, not by
  // the GPU.  So (1) where the five-launch plan applies the commitments go through it in GROUPS -- one launch chain
  // carries up to HM_MSM_GROUP of them -- and (2) the calling thread only submits chains while a second thread of this
  // call awaits the tickets in order and folds.
  for (size_t i = 0; i < count; ++i)
    if (!d_scalars[i] && n) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_batch_bn256_g1_dev: null scalar array");
  // The chain plan: `order` lists the columns in submission order, chain ch carries order[first[ch] .. first[ch + 1]).
  std::vector<uint32_t> order(count), first;
  std::vector<uint8_t> chain_plain;                // per chain: 1 = run on the plain copy of the points (no table)
  std::vector<uint32_t> chain_live;                // per chain: rows known to survive the compaction at most (0 = unknown)
  for (size_t i = 0; i < count; ++i) order[i] = (uint32_t)i;
  {
    bool small_plan = false;
    const uint8_t* d_inf = nullptr;
    uint32_t dense_group = 1;                       // dense columns one chain of the general pipeline may carry (a table set, whole-set MSMs)
    {
      std::lock_guard<std::mutex> lk(ctx->mu);
      BasesEntry* b = find_bases(*ctx, handle);
      if (!b) return hm_fail(HM_ERR_NOT_FOUND, "hm_msm_batch_bn256_g1_dev: unknown base handle");
      if (offset > b->n || n > b->n - offset) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_batch_bn256_g1_dev: offset + n exceeds the base set");
      // the five-launch plan applies to the plain copy of the points, which a table set holds too (its first n entries):
      // sparse columns go there, dense ones take the table's shared bucket set through the general pipeline
      small_plan = msm_group_applies(n, 0);
      d_inf = b->d_inf + offset;
      if (offset == 0 && n == b->n && b->pc_c) dense_group = msm_table_group_max(n, b->pc_c);
    }
    static const size_t group_max_n = [] { const char* v = std::getenv("HALO2_MI355X_GROUP_MAX_LOG"); return (size_t)1 << (v && *v ? std::atoi(v) : 16); }();
    static const bool group_sparse = [] { const char* v = std::getenv("HALO2_MI355X_GROUP_SPARSE"); return !(v && *v == '0'); }();
    if (small_plan && n <= group_max_n) {
      // a commitment alone is bound by launch gaps here: consecutive groups, enough chains to keep several in flight,
      // none longer than a group: ceil(count / chains)
      // (and as many chains as waiter threads when there are commitments for them: the host fold of a chain's results is
      // serial per chain)
      const size_t chains_min = (count + HM_MSM_GROUP - 1) / HM_MSM_GROUP;
#ifndef HM_BATCH_SMALL_CHAINS
#define HM_BATCH_SMALL_CHAINS 4
#endif
      const size_t want = count / 2 < (size_t)HM_BATCH_SMALL_CHAINS ? count / 2 : (size_t)HM_BATCH_SMALL_CHAINS;
      const size_t chains = chains_min > want ? chains_min : want;
      size_t per_chain = (count + chains - 1) / (chains ? chains : 1);
      if (per_chain < 1) per_chain = 1;
      for (size_t f = 0; f < count; f += per_chain) {
        first.push_back((uint32_t)f);
        chain_plain.push_back(1);
      }
    } else if (small_plan && group_sparse && count >= 2) {
      // 2^17 .. 2^18: what a column costs depends on how many of its 256-row blocks SURVIVE the digits kernel's
      // compaction (zero scalars and identity bases contribute nothing), not on n.  A dense column is throughput-bound
      // and keeps a chain of its own (eight separate chains in flight interleave their phases better than one chain of
      // eight: measured 0.57 against 0.79 ms per commitment at 2^18); a SPARSE one -- an advice column with ~1 100 used
      // rows of 2^18 -- is pure chain latency (0.098 ms each one chain at a time, 36 of them 3.5 ms), so up to eight of
      // them share one launch chain.  The classification only decides the grouping, never the result.
      std::vector<uint32_t> live(count, 0);
      const uint32_t total_blocks = (uint32_t)((n + 255) / 256);
      int crc = from_host ? HM_OK : msm_count_live_blocks(*ctx, d_scalars, count, d_inf, n, (hipStream_t)stream, live.data());
      if (crc != HM_OK) return crc;
      if (from_host) {
        // host arrays: a sample decides (reading every word on the host would cost more than the upload): every 64th
        // block, never the first two or the last one (used rows lead, blinding rows trail)
        for (size_t i = 0; i < count; ++i) {
          const uint64_t* s = (const uint64_t*)d_scalars[i];
          uint32_t hits = 0, seen = 0;
          for (uint32_t blk = 2; blk + 1 < total_blocks; blk += 64) {
            const size_t lo = (size_t)blk * 256, hi = lo + 256 < n ? lo + 256 : n;
            uint64_t any = 0;
            for (size_t w = lo * 4; w < hi * 4; ++w) any |= s[w];
            hits += any != 0;
            ++seen;
          }
          live[i] = seen == 0 || hits != 0 ? total_blocks : 0;
        }
      }
      // Dense columns on a table set: up to `dense_group` of them share one chain of the general pipeline (every element a
      // bucket set of the same launches): the sort and the two-launch reduction then run at the chip's throughput instead
      // of as 26 small launches per commitment, and K3 is one launch over all of them.  How many per chain: as many chains
      // as keep three in flight, none longer than the plan allows.
      uint32_t n_dense = 0;
      for (size_t i = 0; i < count; ++i) n_dense += (uint64_t)live[i] * 16 > total_blocks ? 1u : 0u;
      uint32_t dense_per_chain = 1;
      if (dense_group > 1 && n_dense > 1) {
        const uint32_t chains = std::max<uint32_t>((n_dense + dense_group - 1) / dense_group, std::min<uint32_t>(3u, n_dense / 2));
        dense_per_chain = (n_dense + chains - 1) / chains;
      }
      std::vector<uint32_t> pending, plan, dense_pending;
      auto flush_dense = [&]() {
        if (dense_pending.empty()) return;
        first.push_back((uint32_t)plan.size());
        chain_plain.push_back(0);
        chain_live.resize(first.size(), 0);
        plan.insert(plan.end(), dense_pending.begin(), dense_pending.end());
        dense_pending.clear();
      };
      auto flush = [&]() {
        if (pending.empty()) return;
        first.push_back((uint32_t)plan.size());
        chain_plain.push_back(1);
        uint64_t rows = 0;                                       // device columns: counted blocks; host columns: a sample said "sparse" only
        if (!from_host)
          for (uint32_t i : pending) rows = std::max<uint64_t>(rows, (uint64_t)live[i] * 256);
        chain_live.resize(first.size() - 1, 0);
        chain_live.push_back((uint32_t)std::min<uint64_t>(rows ? rows : 0, n));
        plan.insert(plan.end(), pending.begin(), pending.end());
        pending.clear();
      };
      for (size_t i = 0; i < count; ++i) {
        if ((uint64_t)live[i] * 16 <= total_blocks) {           // sparse: joins the pending group
          pending.push_back((uint32_t)i);
          if (pending.size() == (size_t)HM_MSM_GROUP) flush();
        } else {                                                 // dense: on the table when the set has one, several per chain
          dense_pending.push_back((uint32_t)i);
          if (dense_pending.size() >= dense_per_chain) flush_dense();
        }
      }
      flush_dense();
      flush();
      order = plan;
    } else {
      for (size_t f = 0; f < count; ++f) first.push_back((uint32_t)f);
    }
    first.push_back((uint32_t)count);
  }
  const size_t n_chains = first.size() - 1;
  chain_plain.resize(n_chains, 0);
  chain_live.resize(n_chains, 0);

  uint64_t tickets[kLanes];
  // chain ch uses lane ch % kLanes; the lane is free again once chain ch - kLanes has been awaited (finished[] is set)
  std::unique_ptr<std::atomic<uint8_t>[]> finished(new std::atomic<uint8_t>[n_chains + 1]);
  for (size_t i = 0; i <= n_chains; ++i) finished[i].store(0, std::memory_order_relaxed);
  std::atomic<size_t> issued{0};
  std::atomic<int> submit_rc{HM_OK}, wait_rc{HM_OK};
  std::atomic<bool> no_more{false};
  int device = 0;
  HM_HIP_CHECK(hipGetDevice(&device));
  auto await_chain = [&](size_t d) {           // chain d carries the MSMs order[first[d] .. first[d + 1])
    int wrc;
    try {
      hm_fault_point("batch_await");
      uint64_t res[12 * HM_MSM_GROUP];
      const uint32_t members = first[d + 1] - first[d];
      wrc = wait_chain(ctx, tickets[d % kLanes], res, members);
      if (wrc == HM_OK)
        for (uint32_t e = 0; e < members; ++e) std::memcpy(out_xyz + 12 * (size_t)order[first[d] + e], res + 12 * e, 96);
    } catch (...) {                            // nothing may escape a waiter thread, and the chain must still count as awaited
      wrc = HM_ERR_INTERNAL;
      abandon_ticket(ctx, tickets[d % kLanes]);
    }
    if (wrc != HM_OK) {
      int expect = HM_OK;
      (void)wait_rc.compare_exchange_strong(expect, wrc);
    }
    finished[d].store(1, std::memory_order_release);
  };
  // Finishing a chain is host work too: the event wait, then a 255-doubling fold per commitment (~50 us each: 2 ms for
  // the 36 advice columns of a k = 18 proof on one thread -- measured as 40 % GPU idle time in that phase).  So up to
  // kWaiters threads of this call await the chains, waiter t taking chains t, t + T, t + 2T, ...; the calling thread
  // only submits.  They are started after the first lanes are filled, so that their creation overlaps the GPU's work.
#ifndef HM_BATCH_WAITERS
#define HM_BATCH_WAITERS 8      // A/B knob (tools/ab_build.sh): 8 against 4 -- 36 sparse commitments at k = 18 1.60 -> 1.41 ms, dense phases unchanged
#endif
  constexpr size_t kWaiters = HM_BATCH_WAITERS;
  size_t n_waiters = 0;                         // set before any waiter starts
  auto waiter = [&](size_t t) {
    (void)hipSetDevice(device);
    for (size_t d = t; d < n_chains; d += n_waiters) {
      while (issued.load(std::memory_order_acquire) <= d) {
        if (no_more.load(std::memory_order_acquire) && issued.load(std::memory_order_acquire) <= d) return;
        std::this_thread::yield();
      }
      await_chain(d);                          // even after an error: no ticket is left behind
    }
  };
  // host arrays are staged per lane in buffers of the device context: whole _h batch calls of different threads take turns.
  // Declared BEFORE the waiter guard, so that on every way out -- an exception unwinding out of the submit loop included --
  // the lock is released only AFTER the waiters have been joined and no chain of this call reads batch_io[] any more.
  std::unique_lock<std::mutex> host_turn;
  if (from_host) host_turn = std::unique_lock<std::mutex>(ctx->batch_h_mu);
  // the waiters are told to finish and are joined on EVERY way out of this function (an exception in the submit loop included)
  struct WaiterGuard {
    std::atomic<bool>& no_more;
    JoinOnExit pool;
    ~WaiterGuard() { no_more.store(true, std::memory_order_release); }     // members are destroyed after this body: then the join
  } guard{no_more, {}};
  bool threaded = false, waiters_started = false;
  size_t next_unthreaded = 0;                   // without waiters: the next chain the calling thread has to await itself
  auto start_waiters = [&]() {
    waiters_started = true;
    if (n_chains <= 1) return;                  // a lone chain: the calling thread awaits it
    n_waiters = n_chains < kWaiters ? n_chains : kWaiters;
    size_t started = 0;
    for (size_t t = 0; t < n_waiters; ++t)
      if (spawn_or_false(guard.pool, "batch_waiter_spawn", [&waiter, t] { waiter(t); })) ++started;
      else break;
    if (started == n_waiters) {
      threaded = true;
    } else {                                    // not every waiter could be had: the ones that started take what they take,
      no_more.store(true, std::memory_order_release);   // ... are drained, and this thread awaits everything still open
      for (auto& th : guard.pool.th) th.join();
      guard.pool.th.clear();
      no_more.store(false, std::memory_order_release);
    }
  };
  auto await_unthreaded_upto = [&](size_t limit) {      // the calling thread awaits every issued, unfinished chain below `limit`
    for (; next_unthreaded < limit && next_unthreaded < issued.load(); ++next_unthreaded)
      if (!finished[next_unthreaded].load(std::memory_order_acquire)) await_chain(next_unthreaded);
  };
  std::string submit_error;
  try {
  for (size_t ch = 0; ch < n_chains; ++ch) {
    hm_fault_point("batch_submit");
    if (ch >= (size_t)kLanes) {                 // every lane holds a ticket: chain ch - kLanes has to be awaited first
      if (!waiters_started) start_waiters();
      while (!finished[ch - kLanes].load(std::memory_order_acquire)) {
        if (threaded) std::this_thread::yield();
        else await_unthreaded_upto(ch - kLanes + 1);
      }
    }
    const uint32_t group = first[ch + 1] - first[ch];
    int rc;
    // host arrays: this chain's scalars cross PCIe on its own lane's stream (the lane's staging buffer is free again:
    // the chain that used it eight chains ago has been awaited), while the other lanes' chains compute
    const void* staged[HM_MSM_GROUP];
    for (uint32_t e = 0; e < group; ++e) staged[e] = d_scalars[order[first[ch] + e]];
    const void* const* chain_scalars = staged;
    if (from_host && n) {
      const int lane = (int)(ch % kLanes);
      const double t_h2d0 = now_us();
      uint8_t* buf;
      {
        std::lock_guard<std::mutex> lk(ctx->mu);
        buf = (uint8_t*)ctx->batch_io[lane].ensure((size_t)group * n * 32);
      }
      if (!buf) {
        submit_rc.store(hm_fail(HM_ERR_HIP, "hm_msm_batch_bn256_g1_h: staging allocation failed"));
        submit_error = hm_last_error();
        break;
      }
      // through the library's pinned lanes (xfer.hip), synchronous for this thread: the other lanes' chains compute meanwhile, and
      // this lane's chain is submitted behind it (a pageable hipMemcpyAsync blocked the submitting thread just the same)
      // the chain's arrays as ONE job of the lanes (an 8 MiB copy alone spends half its time starting threads and filling its pipeline)
      void* up_dev[HM_MSM_GROUP];
      void* up_host[HM_MSM_GROUP];
      size_t up_bytes[HM_MSM_GROUP];
      for (uint32_t e = 0; e < group; ++e) {
        up_dev[e] = buf + (size_t)e * n * 32;
        up_host[e] = const_cast<void*>(staged[e]);
        up_bytes[e] = n * 32;
      }
      const int xrc = xfer_many(*ctx, true, up_dev, up_host, up_bytes, group, "hm_msm_batch_bn256_g1_h: scalar upload");
      for (uint32_t e = 0; e < group; ++e) staged[e] = up_dev[e];
      if (xrc != HM_OK) {
        submit_rc.store(xrc);
        submit_error = hm_last_error();
        break;
      }
      std::lock_guard<std::mutex> lk(ctx->mu);
      ctx->calls.msm_h2d_us += now_us() - t_h2d0;
      ctx->calls.h2d_bytes += (uint64_t)group * n * 32;
    }
    const double t_wait0 = now_us();
    for (;;) {                                  // slots held by other callers' tickets (another thread's batch): wait for one
      bool all_busy = false;
      rc = submit_chain(ctx, handle, offset, chain_scalars, group, n, ctx->batch_streams[ch % kLanes], &tickets[ch % kLanes],
                        "hm_msm_batch_bn256_g1_dev", &all_busy, chain_plain[ch] == 0, chain_live[ch]);
      if (rc == HM_OK || !all_busy) break;
      if (now_us() - t_wait0 > 60e6) break;     // nobody awaits the tickets that hold the slots: report instead of spinning
      if (!waiters_started) start_waiters();
      if (!threaded && next_unthreaded < issued.load()) await_unthreaded_upto(next_unthreaded + 1);     // free one of our own first
      else std::this_thread::yield();
    }
    if (rc != HM_OK) {
      submit_rc.store(rc);
      submit_error = hm_last_error();
      break;
    }
    issued.store(ch + 1, std::memory_order_release);
  }
  } catch (...) {
    // Something threw between two submissions (vector growth, a fault point).  Chains already issued are in flight on the
    // lanes' staging buffers: the waiters (if any) are drained, then this thread awaits -- or, failing that, abandons --
    // every ticket no waiter took, so that no slot stays busy and nothing reads batch_io[] when host_turn is released.
    no_more.store(true, std::memory_order_release);
    for (auto& t : guard.pool.th)
      if (t.joinable()) t.join();
    guard.pool.th.clear();
    const size_t upto = issued.load();
    for (size_t d = 0; d < upto; ++d)
      if (!finished[d].load(std::memory_order_acquire)) await_chain(d);      // await_chain never throws: it abandons the ticket instead
    throw;
  }
  if (!waiters_started) start_waiters();
  no_more.store(true, std::memory_order_release);
  if (threaded) {
    for (auto& t : guard.pool.th) t.join();
    guard.pool.th.clear();
  }
  await_unthreaded_upto(n_chains);              // whatever no waiter took (none when they all started)
  if (submit_rc.load() != HM_OK) return hm_fail(submit_rc.load(), submit_error);
  if (wait_rc.load() != HM_OK) return hm_fail(wait_rc.load(), "hm_msm_batch_bn256_g1_dev: a commitment of the batch failed (see the waiter's error)");
  return HM_OK;
}

int hm_msm_batch_bn256_g1_dev(uint64_t handle, size_t offset, const void* const* d_scalars, size_t n, size_t count, void* stream,
                              uint64_t* out_xyz) try {
  if ((count && (!d_scalars || !out_xyz))) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_batch_bn256_g1_dev: null argument");
  if (is_multi_handle(handle)) return multi_msm_batch(handle, offset, d_scalars, false, n, count, stream, out_xyz);
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return msm_batch_impl(ctx, handle, offset, d_scalars, false, n, count, stream, out_xyz);
} HM_API_CATCH("hm_msm_batch_bn256_g1_dev")

// The same for scalar arrays in HOST memory (what halo2's prover holds today): each chain's upload runs on its own
// lane's stream, so PCIe time hides behind the other chains' kernels.
int hm_msm_batch_bn256_g1_h(uint64_t handle, size_t offset, const uint64_t* const* scalars, size_t n, size_t count, uint64_t* out_xyz) try {
  if ((count && (!scalars || !out_xyz))) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_batch_bn256_g1_h: null argument");
  if (is_multi_handle(handle))
    return multi_msm_batch(handle, offset, reinterpret_cast<const void* const*>(scalars), true, n, count, nullptr, out_xyz);
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return msm_batch_impl(ctx, handle, offset, reinterpret_cast<const void* const*>(scalars), true, n, count, nullptr, out_xyz);
} HM_API_CATCH("hm_msm_batch_bn256_g1_h")

// Await one ticket: out_xyz receives 12 words per MSM of its chain (at most `capacity` of them).
static int wait_chain(DeviceCtx* ctx, uint64_t ticket, uint64_t* out_xyz, uint32_t capacity) {
  int slot = -1;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int i = 1; i < HM_MSM_SLOTS; ++i)
      if (ctx->msm_slots[i].busy && ctx->msm_slots[i].ticket == ticket) slot = i;
    if (slot < 0) return hm_fail(HM_ERR_NOT_FOUND, "hm_msm_wait: unknown ticket");
    if (ctx->msm_slots[slot].awaiting) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_wait: another thread is already waiting for this ticket");
    if (ctx->msm_slots[slot].n != 0 && ctx->msm_slots[slot].group > capacity)
      return hm_fail(HM_ERR_BAD_ARG, "hm_msm_wait: the ticket belongs to a batch call");
    ctx->msm_slots[slot].awaiting = true;
  }
  // the blocking part -- the device-side wait and the host fold -- runs WITHOUT the context lock: other threads keep
  // submitting while this one waits (the slot stays busy, so nobody else touches it)
  MsmSlot& sl = ctx->msm_slots[slot];
  int is_id[HM_MSM_GROUP] = {};
  double host_us = 0;
  const int rc = msm_finish_wait_fold(sl, out_xyz, is_id, &host_us);
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (rc != HM_OK) sl.live_ptr = nullptr;      // a chain that failed may have left its block counters anywhere
  if (rc == HM_OK) {
    msm_finish_record(*ctx, slot, host_us);
    for (uint32_t e = 0; e < (sl.n ? sl.group : 1u); ++e) count_msm(*ctx, sl.n, e == 0 ? ctx->last_msm.t_total_ms : 0.0);
  }
  sl.awaiting = false;
  sl.busy = false;
  // a base set released while this ticket was in flight: free it once no other ticket reads it
  for (size_t z = 0; z < ctx->zombie_bases.size();) {
    bool used = false;
    for (int k = 1; k < HM_MSM_SLOTS; ++k)
      if (ctx->msm_slots[k].busy && ctx->msm_slots[k].bases_handle == ctx->zombie_bases[z].handle) used = true;
    if (used) { ++z; continue; }
    free_bases_entry(*ctx, ctx->zombie_bases[z]);
    ctx->zombie_bases.erase(ctx->zombie_bases.begin() + z);
  }
  return rc;
}

int hm_msm_wait(uint64_t ticket, uint64_t out_xyz[12]) try {
  if (!out_xyz) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_wait: null output");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return wait_chain(ctx, ticket, out_xyz, 1);
} HM_API_CATCH("hm_msm_wait")

int hm_msm_bn256_g1_h(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n, uint64_t out_xy[8],
                      int* out_is_identity) try {
  if (!out_xy || (n && !scalars)) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1_h: null argument");
  uint64_t jac[12];
  int is_id = 0;
  const int rc = is_multi_handle(handle) ? multi_msm(handle, offset, scalars, true, n, nullptr, jac, &is_id)
                                         : msm_h_local(handle, offset, scalars, n, jac, &is_id);
  if (rc != HM_OK) return rc;
  return jac_to_affine_out(jac, is_id, out_xy, out_is_identity);
} HM_API_CATCH("hm_msm_bn256_g1_h")

extern "C++" {
// Keyed digest of a host array over EVERY word.  It keys the converted-base cache of the drop-in call: unlike round 1's
// 64-point probe it reads the whole array, so a buffer that was mutated at any index -- or re-allocated at the same
// address with other contents -- hashes differently.  The patched best_multiexp also serves the verifier, whose base
// array holds prover-chosen commitments, so an unkeyed mixing function would let a third party construct two arrays
// with one digest.  This one is a universal hash under a per-process random key the caller of the library never sees:
//   inner  NH (UMAC): per 512-byte block  sum_j (m[2j] + k[2j]) * (m[2j+1] + k[2j+1])  mod 2^128 -- two equal-length
//          blocks that differ collide with probability 2^-64 over the key
//   outer  the 128-bit block values as three coefficients each of two polynomials over GF(2^61 - 1), evaluated at two
//          secret points (Horner): a difference anywhere survives with probability 1 - (3 blocks / 2^61)^2
// ~1 multiplication per 16 bytes: as fast as the multiply-rotate lanes it replaces (the digest must stay cheaper than
// the upload it saves).
namespace {
constexpr uint64_t kP61 = (1ull << 61) - 1;
inline uint64_t mulmod61(uint64_t a, uint64_t b) {
  const unsigned __int128 t = (unsigned __int128)a * b;
  uint64_t r = (uint64_t)(t & kP61) + (uint64_t)(t >> 61);
  r = (r & kP61) + (r >> 61);
  return r >= kP61 ? r - kP61 : r;
}
inline uint64_t addmod61(uint64_t a, uint64_t b) {
  uint64_t r = a + b;             // both < 2^61
  return r >= kP61 ? r - kP61 : r;
}
struct DigestKey {
  uint64_t nh[64];
  uint64_t r1, r2, s1, s2;
};
const DigestKey& digest_key() {
  static const DigestKey key = [] {
    DigestKey k;
    uint64_t seed[8];
    try {
      std::random_device rd;
      for (auto& w : seed) w = ((uint64_t)rd() << 32) ^ rd();
    } catch (...) {             // no entropy source: address-space layout and the clock still differ per process
      const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
      for (int i = 0; i < 8; ++i) seed[i] = t * (2 * i + 1) ^ (uint64_t)(uintptr_t)&k ^ (0x9E3779B97F4A7C15ULL * (i + 1));
    }
    uint64_t x = seed[0] ^ seed[1] ^ seed[2] ^ seed[3] ^ seed[4] ^ seed[5] ^ seed[6] ^ seed[7], y = seed[3] * 3 + seed[5];
    auto next = [&]() {        // splitmix64 over the seeded state: expands the entropy, adds none
      x += 0x9E3779B97F4A7C15ULL + y;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      return z ^ (z >> 31);
    };
    for (auto& w : k.nh) w = next();
    k.r1 = next() % (kP61 - 2) + 1; k.r2 = next() % (kP61 - 2) + 1;
    k.s1 = next() % (kP61 - 2) + 1; k.s2 = next() % (kP61 - 2) + 1;
    return k;
  }();
  return key;
}
}  // namespace

// digest of words [0, count): out = { poly1, poly2 } (both < 2^61)
static void digest_words(const uint64_t* w, size_t count, uint64_t out[2]) {
  const DigestKey& K = digest_key();
  uint64_t a1 = 0, a2 = 0;
  auto absorb = [&](unsigned __int128 nh) {
    const uint64_t lo = (uint64_t)nh, hi = (uint64_t)(nh >> 64);
    const uint64_t c0 = lo & kP61, c1 = ((lo >> 61) | (hi << 3)) & kP61, c2 = hi >> 58;
    a1 = addmod61(mulmod61(addmod61(mulmod61(addmod61(mulmod61(a1, K.r1), c0), K.r1), c1), K.r1), c2);
    a2 = addmod61(mulmod61(addmod61(mulmod61(addmod61(mulmod61(a2, K.r2), c0), K.r2), c1), K.r2), c2);
  };
  size_t i = 0;
  for (; i + 64 <= count; i += 64) {
    unsigned __int128 nh = 0;
    for (int j = 0; j < 64; j += 2) nh += (unsigned __int128)(w[i + j] + K.nh[j]) * (w[i + j + 1] + K.nh[j + 1]);
    absorb(nh);
  }
  if (i < count) {                               // last, partial block: zero-padded (the length is part of the key)
    uint64_t pad[64] = {};
    std::memcpy(pad, w + i, (count - i) * 8);
    unsigned __int128 nh = 0;
    for (int j = 0; j < 64; j += 2) nh += (unsigned __int128)(pad[j] + K.nh[j]) * (pad[j + 1] + K.nh[j + 1]);
    absorb(nh);
  }
  out[0] = a1;
  out[1] = a2;
}

static void digest_bases(const uint64_t* bases, size_t n, uint64_t out[4]) {
  hm_fault_point("digest");
  const size_t words = n * 8;
  const unsigned parts = words >= (1u << 20) ? 4u : 1u;      // >= 8 MiB: four host threads (the digest must stay cheaper than the upload)
  uint64_t part[4][2] = {};
  auto lo_of = [&](unsigned p) { return (words * p / parts) & ~(size_t)63; };    // block-aligned cuts
  auto run = [&](unsigned p) {
    const size_t lo = lo_of(p), hi = p + 1 == parts ? words : lo_of(p + 1);
    digest_words(bases + lo, hi - lo, part[p]);
  };
  {
    JoinOnExit pool;                              // joined before `part` is read, and on every other way out
    bool done[4] = {true, false, false, false};
    for (unsigned p = 1; p < parts; ++p) done[p] = spawn_or_false(pool, "digest_spawn", [&run, p] { run(p); });
    run(0);
    for (unsigned p = 1; p < parts; ++p)
      if (!done[p]) run(p);                       // no thread to be had: this one does the part too
  }
  const DigestKey& K = digest_key();
  uint64_t h1 = 0, h2 = 0;
  for (unsigned p = 0; p < parts; ++p) {
    h1 = addmod61(mulmod61(h1, K.s1), part[p][0]);
    h2 = addmod61(mulmod61(h2, K.s2), part[p][1]);
  }
  out[0] = h1;
  out[1] = h2;
  out[2] = (uint64_t)words;
  out[3] = parts;
}

}  // extern "C++"

// The drop-in form of best_multiexp: both arrays are host memory.  The scalars cross PCIe in every call; the converted
// bases of the previous call are kept per device and reused only when the FULL-CONTENT digest and the length match (the
// pointer is not part of the key: create_proof passes the same params.g / g_lagrange prefix to every commitment, and a
// buffer reused with other contents -- the verifier's MSMs -- simply misses).  hm_set_host_base_cache(0) disables it.
static int msm_host_one(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t jac[12], int* is_id) {
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n == 0) return msm_run(*ctx, nullptr, nullptr, nullptr, 0, 0, jac, is_id, nullptr);
  uint32_t* d_xy = (uint32_t*)ctx->conv_bases.ensure(n * 64);
  uint8_t* d_inf = (uint8_t*)ctx->conv_inf.ensure(n);
  void* d_s = ctx->io.ensure(n * 32);
  if (!d_xy || !d_inf || !d_s) return hm_fail(HM_ERR_HIP, "hm_msm_bn256_g1: staging allocation failed");
  const double t0 = now_us();
  uint64_t dg[4] = {0, 0, 0, 0};
  const bool use_cache = g_host_base_cache.load(std::memory_order_relaxed) != 0;
  if (use_cache) digest_bases(bases, n, dg);
  const bool hit = use_cache && ctx->cached_host_n == n && ctx->cached_xy == d_xy &&
                   std::memcmp(dg, ctx->cached_digest, sizeof dg) == 0;
  if (!hit) {
    ctx->cached_host_n = 0;
    void* stage = ctx->io_bases.ensure(n * 64);
    if (!stage) return hm_fail(HM_ERR_HIP, "hm_msm_bn256_g1: staging allocation failed");
    int rc = xfer_h2d(*ctx, stage, bases, n * 64, "hm_msm_bn256_g1: base upload");
    if (rc != HM_OK) return rc;
    rc = msm_convert_bases((const uint32_t*)stage, d_xy, d_inf, n, nullptr);
    if (rc != HM_OK) return rc;
    ctx->calls.h2d_bytes += n * 64;
    if (use_cache) {
      ctx->cached_host_n = n;
      ctx->cached_xy = d_xy;                    // a regrown buffer is a miss
      std::memcpy(ctx->cached_digest, dg, sizeof dg);
    }
  }
  {
    const int rc = xfer_h2d(*ctx, d_s, scalars, n * 32, "hm_msm_bn256_g1: scalar upload");
    if (rc != HM_OK) return rc;
  }
  ctx->calls.msm_h2d_us += now_us() - t0;
  ctx->calls.h2d_bytes += n * 32;
  int rc = msm_run(*ctx, (const uint32_t*)d_s, d_xy, d_inf, n, 0, jac, is_id, nullptr);
  if (rc == HM_OK) count_msm(*ctx, n, ctx->last_msm.t_total_ms);
  return rc;
}

// Single-process multi-GPU form of the host-pointer call (hm_set_msm_devices): contiguous index ranges, one host
// thread per device (multi.hip: run_per_device), each running the ordinary one-device path on its slice (its own
// uploads, no shared state), partial sums folded on the host.  No inter-GPU traffic: the only thing that leaves a
// device is a 96-byte point.
static int msm_host(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t jac[12], int* is_id) {
  if (n && (!scalars || !bases)) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1: null argument");
  const std::vector<int> devs = msm_device_list();
  const size_t parts = devs.size();
  if (parts < 2) return msm_host_one(scalars, bases, n, jac, is_id);
  if (n < parts * kMinShardPoints)     // too small to be worth splitting: the first listed device takes it whole
    return run_per_device({devs[0]}, [&](size_t) { return msm_host_one(scalars, bases, n, jac, is_id); });
  std::vector<uint64_t> partial(parts * 12, 0);
  const int rc = run_per_device(devs, [&](size_t r) {
    const size_t lo = n * r / parts, hi = n * (r + 1) / parts;
    int id = 0;
    return msm_host_one(scalars + lo * 4, bases + lo * 8, hi - lo, &partial[r * 12], &id);
  });
  if (rc != HM_OK) return rc;
  host_sum_points(partial.data(), parts, jac, is_id);
  return HM_OK;
}

int hm_set_msm_devices(const int* devices, int count) try {
  if (count < 0 || count > 64 || (count && !devices)) return hm_fail(HM_ERR_BAD_ARG, "hm_set_msm_devices: bad device list");
  const int visible = hm_device_count();
  if (count && visible <= 0) return hm_fail(HM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  for (int i = 0; i < count; ++i)
    if (devices[i] < 0 || devices[i] >= visible) return hm_fail(HM_ERR_BAD_ARG, "hm_set_msm_devices: device index out of range");
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  g_msm_devices.assign(devices, devices + count);
  return HM_OK;
} HM_API_CATCH("hm_set_msm_devices")

int hm_msm_bn256_g1(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xy[8], int* out_is_identity) try {
  if (!out_xy) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1: null output");
  uint64_t jac[12];
  int is_id = 0;
  int rc = msm_host(scalars, bases, n, jac, &is_id);
  if (rc != HM_OK) return rc;
  return jac_to_affine_out(jac, is_id, out_xy, out_is_identity);
} HM_API_CATCH("hm_msm_bn256_g1")

int hm_msm_bn256_g1_jacobian(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xyz[12]) try {
  if (!out_xyz) return hm_fail(HM_ERR_BAD_ARG, "hm_msm_bn256_g1_jacobian: null output");
  int is_id = 0;
  return msm_host(scalars, bases, n, out_xyz, &is_id);
} HM_API_CATCH("hm_msm_bn256_g1_jacobian")

int hm_g1_sum(const uint64_t* points_xyz, size_t count, uint64_t out_xyz[12]) try {
  if (!out_xyz || (count && !points_xyz)) return hm_fail(HM_ERR_BAD_ARG, "hm_g1_sum: null argument");
  if (count > (1u << 20)) return hm_fail(HM_ERR_BAD_ARG, "hm_g1_sum: meant for a handful of partial results");
  hm_fault_point("g1_sum");
  int is_id = 0;
  host_sum_points(points_xyz, count, out_xyz, &is_id);
  return HM_OK;
} HM_API_CATCH("hm_g1_sum")

int hm_get_msm_stats(hm_msm_stats* out) try {
  if (!out) return hm_fail(HM_ERR_BAD_ARG, "hm_get_msm_stats: null output");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const MsmStats& s = ctx->last_msm;
  out->digits_ms = s.t_digits_ms; out->sort_ms = s.t_sort_ms; out->accumulate_ms = s.t_accum_ms;
  out->reduce_ms = s.t_reduce_ms; out->total_ms = s.t_total_ms; out->accumulate_kernel_ms = s.t_accum_kernel_ms;
  out->pairs = s.pairs; out->tasks = s.tasks; out->window_bits = s.c; out->windows = s.windows;
  return HM_OK;
} HM_API_CATCH("hm_get_msm_stats")

int hm_get_stats(hm_stats* out) try {
  if (!out) return hm_fail(HM_ERR_BAD_ARG, "hm_get_stats: null output");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const CallStats& c = ctx->calls;
  std::memset(out, 0, sizeof *out);
  out->msm_calls = c.msm_calls; out->msm_points = c.msm_points;
  out->ntt_calls = c.ntt_calls; out->ntt_elements = c.ntt_elements;
  for (int i = 0; i < 32; ++i) { out->msm_calls_by_log2[i] = c.msm_by_log[i]; out->ntt_calls_by_log2[i] = c.ntt_by_log[i]; }
  out->msm_h2d_us = c.msm_h2d_us; out->msm_device_us = c.msm_device_us; out->msm_host_us = c.msm_host_us;
  out->ntt_h2d_us = c.ntt_h2d_us; out->ntt_device_us = c.ntt_device_us; out->ntt_d2h_us = c.ntt_d2h_us;
  out->h2d_bytes = c.h2d_bytes; out->d2h_bytes = c.d2h_bytes;
  for (int i = 0; i < 8; ++i) { out->vector_calls[i] = c.vector_calls[i]; out->vector_elements[i] = c.vector_elements[i]; }
  out->coset_table_bytes = ctx->coset_table_bytes;
  out->coset_tables = ctx->coset_tables.size();
  out->ntt_table_bytes = ctx->ntt_table_bytes;
  out->ntt_tables = ctx->ntt_tables.size();
  out->host_copies_direct = ctx->xfer.direct.load(std::memory_order_relaxed);
  out->host_copies_staged = ctx->xfer.staged.load(std::memory_order_relaxed);
  out->host_ranges_registered = xfer_host_ranges();
  return HM_OK;
} HM_API_CATCH("hm_get_stats")

int hm_reset_stats(void) try {
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->calls = CallStats{};
  return HM_OK;
} HM_API_CATCH("hm_reset_stats")

// ---- NTT -------------------------------------------------------------------------------------

static void count_vector(DeviceCtx& ctx, int kind, uint64_t calls, uint64_t elements) {   // ctx.mu held
  ctx.calls.vector_calls[kind] += calls;
  ctx.calls.vector_elements[kind] += elements;
}

static void count_ntt(DeviceCtx& ctx, uint32_t log_n, size_t batch) {
  ctx.calls.ntt_calls += batch;
  ctx.calls.ntt_elements += (uint64_t)batch << log_n;
  ctx.calls.ntt_by_log[log_n & 31] += batch;
}

int hm_ntt_bn256_fr_dev(void* d_a, const uint64_t omega[4], uint32_t log_n, void* stream) try {
  if (!d_a || !omega) return hm_fail(HM_ERR_BAD_ARG, "hm_ntt_bn256_fr_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int rc = ntt_run(*ctx, (uint32_t*)d_a, omega, log_n, 1, NttFused{}, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, 1);
  return rc;
} HM_API_CATCH("hm_ntt_bn256_fr_dev")

int hm_ntt_batch_bn256_fr_dev(void* d_a, size_t batch, const uint64_t omega[4], uint32_t log_n, const uint64_t* scale,
                              const uint64_t* coset, void* stream) try {
  if ((batch && !d_a) || !omega) return hm_fail(HM_ERR_BAD_ARG, "hm_ntt_batch_bn256_fr_dev: null argument");
  if (batch > 65535) return hm_fail(HM_ERR_BAD_ARG, "hm_ntt_batch_bn256_fr_dev: batch > 65535");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  NttFused f;
  f.scale = scale;
  f.coset = coset;
  const int rc = ntt_run(*ctx, (uint32_t*)d_a, omega, log_n, (uint32_t)batch, f, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, batch);
  return rc;
} HM_API_CATCH("hm_ntt_batch_bn256_fr_dev")

// ctx->mu is held by the caller.  The zero part of the padded array is neither written nor read when the plan allows it.
static int coeff_to_extended_locked(DeviceCtx* ctx, const void* d_coeffs, void* d_ext, size_t batch, const uint64_t extended_omega[4],
                                    uint32_t log_n, uint32_t log_ext, const uint64_t* coset, void* stream) {
  NttFused f;
  f.coset = coset;
  const uint32_t log_z = log_ext - log_n;
  int passes = 0;
  const int first_digit = ntt_plan_first_digit(log_ext, &passes);
  int rc;
  if (log_z > 0 && passes >= 2 && (int)log_z <= first_digit) {   // the zero part is never written or read
    rc = ntt_run(*ctx, (uint32_t*)d_ext, extended_omega, log_ext, (uint32_t)batch, f, (hipStream_t)stream,
                 (const uint32_t*)d_coeffs, log_z);
  } else {
    // small or un-extended domains: materialise the padded arrays, then the ordinary in-place transform
    const size_t row_in = (size_t)32 << log_n, row_out = (size_t)32 << log_ext;
    if (log_z) HM_HIP_CHECK(hipMemsetAsync(d_ext, 0, row_out * batch, (hipStream_t)stream));
    HM_HIP_CHECK(hipMemcpy2DAsync(d_ext, row_out, d_coeffs, row_in, row_in, batch, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    rc = ntt_run(*ctx, (uint32_t*)d_ext, extended_omega, log_ext, (uint32_t)batch, f, (hipStream_t)stream);
  }
  if (rc == HM_OK) count_ntt(*ctx, log_ext, batch);
  return rc;
}

int hm_coeff_to_extended_bn256_fr_dev(const void* d_coeffs, void* d_ext, size_t batch, const uint64_t extended_omega[4],
                                      uint32_t log_n, uint32_t log_ext, const uint64_t* coset, void* stream) try {
  if ((batch && (!d_coeffs || !d_ext)) || !extended_omega)
    return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_extended_bn256_fr_dev: null argument");
  if (log_ext < log_n || log_ext > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_extended_bn256_fr_dev: need log_n <= log_ext <= 28");
  if (batch > 65535) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_extended_bn256_fr_dev: batch > 65535");
  if (batch == 0) return HM_OK;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return coeff_to_extended_locked(ctx, d_coeffs, d_ext, batch, extended_omega, log_n, log_ext, coset, stream);
} HM_API_CATCH("hm_coeff_to_extended_bn256_fr_dev")

// Host-pointer forms of the two EvaluationDomain steps that cross PCIe in a drop-in prover: only what upstream's arrays really
// hold travels -- the 2^log_n coefficients up (never the zero padding), the first `keep` coefficients down (never the part
// extended_to_coeff truncates).
int hm_coeff_to_extended_bn256_fr(const uint64_t* coeffs, uint64_t* ext, const uint64_t extended_omega[4], uint32_t log_n,
                                  uint32_t log_ext, const uint64_t* coset) try {
  if (!coeffs || !ext || !extended_omega) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_extended_bn256_fr: null argument");
  if (log_ext < log_n || log_ext > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_extended_bn256_fr: need log_n <= log_ext <= 28");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t bytes_in = (size_t)32 << log_n, bytes_out = (size_t)32 << log_ext;
  char* d_out = (char*)ctx->io.ensure(bytes_out + bytes_in);
  if (!d_out) return hm_fail(HM_ERR_HIP, "hm_coeff_to_extended_bn256_fr: staging allocation failed");
  char* d_in = d_out + bytes_out;
  const double t0 = now_us();
  {
    const int rc = xfer_h2d(*ctx, d_in, coeffs, bytes_in, "hm_coeff_to_extended_bn256_fr: upload");
    if (rc != HM_OK) return rc;
  }
  const double t1 = now_us();
  const int rc = coeff_to_extended_locked(ctx, d_in, d_out, 1, extended_omega, log_n, log_ext, coset, nullptr);
  if (rc != HM_OK) return rc;
  if (xfer_mode(ext, bytes_out) == 0) xfer_prefault(ext, bytes_out);   // direct copies: under the transform (`ext` is normally a fresh allocation)
  HM_HIP_CHECK(hipStreamSynchronize(nullptr));
  const double t2 = now_us();
  // `ext` is written by this copy alone (it may be the very allocation `coeffs` lives in: the input has been uploaded whole).  It is
  // normally a FRESH allocation (the new Vec of the result): its first-touch page faults are taken by xfer_prefault's threads above or
  // by the lanes' copying threads.
  if (xfer_d2h(*ctx, ext, d_out, bytes_out, "hm_coeff_to_extended_bn256_fr") != HM_OK)
    return hm_fail(HM_ERR_PARTIAL_OUTPUT, "hm_coeff_to_extended_bn256_fr: copying the result back failed, the output is partly written: " +
                                              hm_last_error_string());
  const double t3 = now_us();
  ctx->calls.ntt_h2d_us += t1 - t0;
  ctx->calls.ntt_device_us += t2 - t1;
  ctx->calls.ntt_d2h_us += t3 - t2;
  ctx->calls.h2d_bytes += bytes_in;
  ctx->calls.d2h_bytes += bytes_out;
  return HM_OK;
} HM_API_CATCH("hm_coeff_to_extended_bn256_fr")

int hm_extended_to_coeff_bn256_fr(uint64_t* a, size_t keep, const uint64_t extended_omega_inv[4], uint32_t log_ext,
                                  const uint64_t divisor[4], const uint64_t coset_inv[12]) try {
  if (!a || !extended_omega_inv || !divisor || !coset_inv) return hm_fail(HM_ERR_BAD_ARG, "hm_extended_to_coeff_bn256_fr: null argument");
  if (log_ext > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_extended_to_coeff_bn256_fr: log_ext > 28");
  if (keep > ((size_t)1 << log_ext)) return hm_fail(HM_ERR_BAD_ARG, "hm_extended_to_coeff_bn256_fr: keep exceeds 2^log_ext");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t bytes = (size_t)32 << log_ext;
  void* d_a = ctx->io.ensure(bytes);
  if (!d_a) return hm_fail(HM_ERR_HIP, "hm_extended_to_coeff_bn256_fr: staging allocation failed");
  const double t0 = now_us();
  {
    const int rc = xfer_h2d(*ctx, d_a, a, bytes, "hm_extended_to_coeff_bn256_fr: upload");
    if (rc != HM_OK) return rc;
  }
  const double t1 = now_us();
  NttFused f;
  f.scale = divisor;
  f.post3 = coset_inv;
  const int rc = ntt_run(*ctx, (uint32_t*)d_a, extended_omega_inv, log_ext, 1, f, nullptr);
  if (rc != HM_OK) return rc;
  HM_HIP_CHECK(hipStreamSynchronize(nullptr));
  const double t2 = now_us();
  if (keep && xfer_d2h(*ctx, a, d_a, keep * 32, "hm_extended_to_coeff_bn256_fr") != HM_OK)
    return hm_fail(HM_ERR_PARTIAL_OUTPUT, "hm_extended_to_coeff_bn256_fr: copying the result back failed, the array is partly overwritten: " +
                                              hm_last_error_string());
  const double t3 = now_us();
  count_ntt(*ctx, log_ext, 1);
  ctx->calls.ntt_h2d_us += t1 - t0;
  ctx->calls.ntt_device_us += t2 - t1;
  ctx->calls.ntt_d2h_us += t3 - t2;
  ctx->calls.h2d_bytes += bytes;
  ctx->calls.d2h_bytes += keep * 32;
  return HM_OK;
} HM_API_CATCH("hm_extended_to_coeff_bn256_fr")

int hm_extended_to_coeff_bn256_fr_dev(void* d_a, size_t batch, const uint64_t extended_omega_inv[4], uint32_t log_ext,
                                      const uint64_t divisor[4], const uint64_t coset_inv[12], void* stream) try {
  if ((batch && !d_a) || !extended_omega_inv || !divisor || !coset_inv)
    return hm_fail(HM_ERR_BAD_ARG, "hm_extended_to_coeff_bn256_fr_dev: null argument");
  if (batch > 65535) return hm_fail(HM_ERR_BAD_ARG, "hm_extended_to_coeff_bn256_fr_dev: batch > 65535");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  NttFused f;
  f.scale = divisor;
  f.post3 = coset_inv;
  const int rc = ntt_run(*ctx, (uint32_t*)d_a, extended_omega_inv, log_ext, (uint32_t)batch, f, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_ext, batch);
  return rc;
} HM_API_CATCH("hm_extended_to_coeff_bn256_fr_dev")

int hm_coeff_to_coset_bn256_fr_dev(const void* d_coeffs, void* d_out, size_t batch, const uint64_t omega[4], uint32_t log_n,
                                   const uint64_t shift[4], int columns_internal, void* stream) try {
  if ((batch && (!d_coeffs || !d_out)) || !omega || !shift) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_coset_bn256_fr_dev: null argument");
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_coset_bn256_fr_dev: log_n > 28");
  if (batch > 65535) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_coset_bn256_fr_dev: batch > 65535");
  if (batch == 0) return HM_OK;
  if (d_coeffs != d_out) {
    const size_t bytes = ((size_t)32 << log_n) * batch;
    const char *a = (const char*)d_coeffs, *b = (const char*)d_out;
    if (a < b + bytes && b < a + bytes)
      return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_coset_bn256_fr_dev: output partially overlaps the coefficients");
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int rc = ntt_coset_run(*ctx, (const uint32_t*)d_coeffs, (uint32_t*)d_out, (uint32_t)batch, omega, log_n, shift, columns_internal != 0,
                               (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, batch);
  return rc;
} HM_API_CATCH("hm_coeff_to_coset_bn256_fr_dev")

int hm_coeff_to_cosets_bn256_fr_dev(const void* d_coeffs, void* d_out, size_t batch, const uint64_t omega[4], uint32_t log_n,
                                    const uint64_t* shifts, size_t count, int columns_internal, void* stream) try {
  if ((batch && count && (!d_coeffs || !d_out)) || !omega || (count && !shifts))
    return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_cosets_bn256_fr_dev: null argument");
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_cosets_bn256_fr_dev: log_n > 28");
  if (count > 16) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_cosets_bn256_fr_dev: at most 16 cosets per call");
  if (batch * (count ? count : 1) > 65535) return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_cosets_bn256_fr_dev: batch * count > 65535");
  if (batch == 0 || count == 0) return HM_OK;
  {
    const size_t in_bytes = ((size_t)32 << log_n) * batch, out_bytes = in_bytes * count;
    const char *a = (const char*)d_coeffs, *b = (const char*)d_out;
    if (a < b + out_bytes && b < a + in_bytes)
      return hm_fail(HM_ERR_BAD_ARG, "hm_coeff_to_cosets_bn256_fr_dev: the output overlaps the coefficients");
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int rc = ntt_cosets_run(*ctx, (const uint32_t*)d_coeffs, (uint32_t*)d_out, (uint32_t)batch, omega, log_n, shifts, (uint32_t)count,
                                columns_internal != 0, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, batch * count);
  return rc;
} HM_API_CATCH("hm_coeff_to_cosets_bn256_fr_dev")

int hm_cosets_to_coeff_bn256_fr_dev(void* d_a, size_t count, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4],
                                    const uint64_t* shift_invs, void* stream) try {
  if ((count && (!d_a || !shift_invs)) || !omega_inv || !divisor) return hm_fail(HM_ERR_BAD_ARG, "hm_cosets_to_coeff_bn256_fr_dev: null argument");
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_cosets_to_coeff_bn256_fr_dev: log_n > 28");
  if (count > 16) return hm_fail(HM_ERR_BAD_ARG, "hm_cosets_to_coeff_bn256_fr_dev: at most 16 cosets per call");
  if (count == 0) return HM_OK;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int rc = ntt_cosets_inverse_run(*ctx, (uint32_t*)d_a, (uint32_t)count, omega_inv, log_n, divisor, shift_invs, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, count);
  return rc;
} HM_API_CATCH("hm_cosets_to_coeff_bn256_fr_dev")

int hm_coset_to_coeff_bn256_fr_dev(void* d_a, size_t batch, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4],
                                   const uint64_t shift_inv[4], void* stream) try {
  if ((batch && !d_a) || !omega_inv || !divisor || !shift_inv) return hm_fail(HM_ERR_BAD_ARG, "hm_coset_to_coeff_bn256_fr_dev: null argument");
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_coset_to_coeff_bn256_fr_dev: log_n > 28");
  if (batch > 65535) return hm_fail(HM_ERR_BAD_ARG, "hm_coset_to_coeff_bn256_fr_dev: batch > 65535");
  if (batch == 0) return HM_OK;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const int rc = ntt_coset_inverse_run(*ctx, (uint32_t*)d_a, (uint32_t)batch, omega_inv, log_n, divisor, shift_inv, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, batch);
  return rc;
} HM_API_CATCH("hm_coset_to_coeff_bn256_fr_dev")

int hm_ntt_bn256_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n) try {
  if (!a || !omega) return hm_fail(HM_ERR_BAD_ARG, "hm_ntt_bn256_fr: null argument");
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "hm_ntt_bn256_fr: log_n > 28");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const size_t bytes = ((size_t)32) << log_n;
  void* d_a = ctx->io.ensure(bytes);
  if (!d_a) return hm_fail(HM_ERR_HIP, "hm_ntt_bn256_fr: staging allocation failed");
  const double t0 = now_us();
  int rc = xfer_h2d(*ctx, d_a, a, bytes, "hm_ntt_bn256_fr: upload");
  if (rc != HM_OK) return rc;
  const double t1 = now_us();
  rc = ntt_run(*ctx, (uint32_t*)d_a, omega, log_n, 1, NttFused{}, nullptr);
  if (rc != HM_OK) return rc;
  HM_HIP_CHECK(hipStreamSynchronize(nullptr));
  const double t2 = now_us();
  // from here on `a` is being overwritten: a failure is NOT one a caller may answer by running its CPU body on `a`
  if (xfer_d2h(*ctx, a, d_a, bytes, "hm_ntt_bn256_fr") != HM_OK)
    return hm_fail(HM_ERR_PARTIAL_OUTPUT, "hm_ntt_bn256_fr: copying the result back failed, the array is partly overwritten: " +
                                              hm_last_error_string());
  const double t3 = now_us();
  count_ntt(*ctx, log_n, 1);
  ctx->calls.ntt_h2d_us += t1 - t0;
  ctx->calls.ntt_device_us += t2 - t1;
  ctx->calls.ntt_d2h_us += t3 - t2;
  ctx->calls.h2d_bytes += bytes;
  ctx->calls.d2h_bytes += bytes;
  return HM_OK;
} HM_API_CATCH("hm_ntt_bn256_fr")

int hm_ifft_bn256_fr_dev(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void* stream) try {
  if (!d_a || !omega_inv || !divisor) return hm_fail(HM_ERR_BAD_ARG, "hm_ifft_bn256_fr_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  NttFused f;
  f.scale = divisor;
  const int rc = ntt_run(*ctx, (uint32_t*)d_a, omega_inv, log_n, 1, f, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, 1);
  return rc;
} HM_API_CATCH("hm_ifft_bn256_fr_dev")

int hm_coset_ntt_bn256_fr_dev(void* d_a, const uint64_t omega[4], uint32_t log_n, const uint64_t coset[12], void* stream) try {
  if (!d_a || !omega || !coset) return hm_fail(HM_ERR_BAD_ARG, "hm_coset_ntt_bn256_fr_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  NttFused f;
  f.coset = coset;
  const int rc = ntt_run(*ctx, (uint32_t*)d_a, omega, log_n, 1, f, (hipStream_t)stream);
  if (rc == HM_OK) count_ntt(*ctx, log_n, 1);
  return rc;
} HM_API_CATCH("hm_coset_ntt_bn256_fr_dev")

int hm_eval_polynomial_bn256_fr_dev(const void* d_polys, size_t n, const uint32_t* poly_index, const uint64_t* points, size_t count,
                                    uint64_t* out, void* stream) try {
  if ((count && (!points || !out)) || (count && n && !d_polys))
    return hm_fail(HM_ERR_BAD_ARG, "hm_eval_polynomial_bn256_fr_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_EVAL_POLYNOMIAL, count, (uint64_t)count * n);
  return fr_eval_polynomial_run(*ctx, (const uint32_t*)d_polys, n, poly_index, points, count, out, (hipStream_t)stream);
} HM_API_CATCH("hm_eval_polynomial_bn256_fr_dev")

int hm_kate_division_bn256_fr_dev(const void* d_poly, size_t n, const uint64_t z[4], void* d_quotient, void* stream) try {
  if (!z || (n >= 2 && (!d_poly || !d_quotient))) return hm_fail(HM_ERR_BAD_ARG, "hm_kate_division_bn256_fr_dev: null argument");
  if (n >= 2) {
    const char *a = (const char*)d_poly, *q = (const char*)d_quotient;
    if (q < a + n * 32 && a < q + (n - 1) * 32)
      return hm_fail(HM_ERR_BAD_ARG, "hm_kate_division_bn256_fr_dev: quotient overlaps the polynomial");
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_KATE_DIVISION, 1, n);
  return fr_kate_division_run(*ctx, (const uint32_t*)d_poly, n, z, (uint32_t*)d_quotient, (hipStream_t)stream);
} HM_API_CATCH("hm_kate_division_bn256_fr_dev")

int hm_fr_grand_product_dev(const void* d_factors, size_t n, const uint64_t start[4], void* d_out, void* stream) try {
  if (!start || (n && (!d_factors || !d_out))) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_grand_product_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_GRAND_PRODUCT, 1, n);
  return fr_grand_product_run(*ctx, (const uint32_t*)d_factors, n, start, (uint32_t*)d_out, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_grand_product_dev")

static bool ranges_overlap(const void* a, size_t a_bytes, const void* b, size_t b_bytes) {
  const char *pa = (const char*)a, *pb = (const char*)b;
  return pa < pb + b_bytes && pb < pa + a_bytes;
}

int hm_kate_division_batch_bn256_fr_dev(const void* const* d_polys, size_t n, const uint64_t* z, void* const* d_quotients, size_t count,
                                        void* stream) try {
  if (count == 0) return HM_OK;
  if (!z || !d_polys || !d_quotients) return hm_fail(HM_ERR_BAD_ARG, "hm_kate_division_batch_bn256_fr_dev: null argument");
  if (n >= 2) {
    for (size_t j = 0; j < count; ++j)
      if (!d_polys[j] || !d_quotients[j]) return hm_fail(HM_ERR_BAD_ARG, "hm_kate_division_batch_bn256_fr_dev: null device pointer");
    for (size_t j = 0; j < count; ++j)
      for (size_t i = 0; i < count; ++i)
        if (ranges_overlap(d_quotients[j], (n - 1) * 32, d_polys[i], n * 32) ||
            (i != j && ranges_overlap(d_quotients[j], (n - 1) * 32, d_quotients[i], (n - 1) * 32)))
          return hm_fail(HM_ERR_BAD_ARG, "hm_kate_division_batch_bn256_fr_dev: a quotient overlaps another array of the call");
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_KATE_DIVISION, count, (uint64_t)count * n);
  return fr_kate_division_batch_run(*ctx, d_polys, n, z, d_quotients, count, (hipStream_t)stream);
} HM_API_CATCH("hm_kate_division_batch_bn256_fr_dev")

int hm_fr_grand_product_batch_dev(const void* const* d_factors, size_t n, const uint64_t start[4], size_t chain_row, void* const* d_out,
                                  size_t count, void* stream) try {
  if (count == 0) return HM_OK;
  if (!start || !d_factors || !d_out) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_grand_product_batch_dev: null argument");
  if (n) {
    for (size_t j = 0; j < count; ++j)
      if (!d_factors[j] || !d_out[j]) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_grand_product_batch_dev: null device pointer");
    for (size_t j = 0; j < count; ++j)
      for (size_t i = 0; i < count; ++i)
        if (i != j && (ranges_overlap(d_out[j], n * 32, d_factors[i], n * 32) || ranges_overlap(d_out[j], n * 32, d_out[i], n * 32)))
          return hm_fail(HM_ERR_BAD_ARG, "hm_fr_grand_product_batch_dev: an output overlaps another column of the call");
    for (size_t j = 0; j < count; ++j)
      if (d_out[j] != d_factors[j] && ranges_overlap(d_out[j], n * 32, d_factors[j], n * 32))
        return hm_fail(HM_ERR_BAD_ARG, "hm_fr_grand_product_batch_dev: an output partially overlaps its factors");
  }
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_GRAND_PRODUCT, count, (uint64_t)count * n);
  return fr_grand_product_batch_run(*ctx, d_factors, n, start, chain_row < n ? chain_row : n, d_out, count, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_grand_product_batch_dev")

int hm_fr_batch_invert_dev(void* d_values, size_t n, void* stream) try {
  if (n && !d_values) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_batch_invert_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    count_vector(*ctx, HM_STAT_BATCH_INVERT, 1, n);
  }
  return fr_batch_invert_run((uint32_t*)d_values, n, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_batch_invert_dev")

int hm_fr_linear_combination_dev(const void* const* d_polys, const uint64_t* coeffs, size_t count, size_t n, void* d_out,
                                 void* stream) try {
  if ((n && !d_out) || (count && (!d_polys || !coeffs))) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_linear_combination_dev: null argument");
  if (n)
    for (size_t j = 0; j < count; ++j)
      if (!d_polys[j]) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_linear_combination_dev: null polynomial");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    count_vector(*ctx, HM_STAT_LINEAR_COMBINATION, 1, (uint64_t)count * n);
  }
  return fr_linear_combination_run(d_polys, coeffs, count, n, (uint32_t*)d_out, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_linear_combination_dev")

int hm_lookup_permute_bn256_fr_dev(const void* d_input, const void* d_table, size_t rows, void* d_permuted_input,
                                   void* d_permuted_table, void* stream) try {
  if (rows && (!d_input || !d_table || !d_permuted_input || !d_permuted_table))
    return hm_fail(HM_ERR_BAD_ARG, "hm_lookup_permute_bn256_fr_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_LOOKUP_PERMUTE, 1, rows);
  return lookup_permute_run(*ctx, &d_input, &d_table, 1, rows, &d_permuted_input, &d_permuted_table, nullptr, (hipStream_t)stream);
} HM_API_CATCH("hm_lookup_permute_bn256_fr_dev")

int hm_lookup_permute_batch_bn256_fr_dev(const void* const* d_inputs, const void* const* d_tables, size_t count, size_t rows,
                                         void* const* d_permuted_inputs, void* const* d_permuted_tables, int* missing, void* stream) try {
  if (count && (!d_inputs || !d_tables || !d_permuted_inputs || !d_permuted_tables))
    return hm_fail(HM_ERR_BAD_ARG, "hm_lookup_permute_batch_bn256_fr_dev: null argument");
  if (rows)
    for (size_t p = 0; p < count; ++p)
      if (!d_inputs[p] || !d_tables[p] || !d_permuted_inputs[p] || !d_permuted_tables[p])
        return hm_fail(HM_ERR_BAD_ARG, "hm_lookup_permute_batch_bn256_fr_dev: null column");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  count_vector(*ctx, HM_STAT_LOOKUP_PERMUTE, count, (uint64_t)count * rows);
  return lookup_permute_run(*ctx, d_inputs, d_tables, count, rows, d_permuted_inputs, d_permuted_tables, missing, (hipStream_t)stream);
} HM_API_CATCH("hm_lookup_permute_batch_bn256_fr_dev")

int hm_fr_mul_periodic_dev(void* d_a, size_t n, const uint64_t* pattern, uint32_t period, void* stream) try {
  if ((n && !d_a) || !pattern) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_mul_periodic_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return fr_mul_periodic_run((uint32_t*)d_a, n, pattern, period, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_mul_periodic_dev")

int hm_fr_powers_dev(void* d_out, size_t n, const uint64_t x[4], void* stream) try {
  if ((n && !d_out) || !x) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_powers_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return fr_powers_run((uint32_t*)d_out, n, x, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_powers_dev")

int hm_graph_create(const uint32_t* calcs, size_t n_calc, const uint64_t* constants, size_t n_const, size_t n_dynamic,
                    const int32_t* rotations, size_t n_rot, size_t n_columns, uint32_t n_intermediates, uint64_t* out_handle) try {
  if (!out_handle || (n_calc && !calcs) || (n_const && !constants) || (n_rot && !rotations))
    return hm_fail(HM_ERR_BAD_ARG, "hm_graph_create: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return graph_create(*ctx, calcs, n_calc, constants, n_const, n_dynamic, rotations, n_rot, n_columns, n_intermediates, out_handle);
} HM_API_CATCH("hm_graph_create")

static int graph_evaluate_entry(const char* who, uint64_t handle, const void* const* d_columns, size_t n_columns,
                                const uint64_t* dynamic_constants, size_t n_dynamic, uint32_t log_size, uint32_t segments, void* d_values,
                                uint32_t flags, void* stream) {
  if (!d_values || (n_columns && !d_columns) || (n_dynamic && !dynamic_constants))
    return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto& g : ctx->graphs)
    if (g->handle == handle) {
      count_vector(*ctx, HM_STAT_GRAPH_EVALUATE, 1, log_size < 40 ? (uint64_t)segments << log_size : 0);
      return graph_evaluate(*ctx, *g, d_columns, n_columns, dynamic_constants, n_dynamic, log_size, segments, d_values, flags,
                            (hipStream_t)stream);
    }
  return hm_fail(HM_ERR_NOT_FOUND, std::string(who) + ": unknown program handle");
}

int hm_graph_evaluate_dev(uint64_t handle, const void* const* d_columns, size_t n_columns, const uint64_t* dynamic_constants,
                          size_t n_dynamic, uint32_t log_size, void* d_values, void* stream) try {
  return graph_evaluate_entry("hm_graph_evaluate_dev", handle, d_columns, n_columns, dynamic_constants, n_dynamic, log_size, 1, d_values, 0, stream);
} HM_API_CATCH("hm_graph_evaluate_dev")

int hm_graph_evaluate_flags_dev(uint64_t handle, const void* const* d_columns, size_t n_columns, const uint64_t* dynamic_constants,
                                size_t n_dynamic, uint32_t log_size, void* d_values, uint32_t flags, void* stream) try {
  return graph_evaluate_entry("hm_graph_evaluate_flags_dev", handle, d_columns, n_columns, dynamic_constants, n_dynamic, log_size, 1, d_values,
                              flags, stream);
} HM_API_CATCH("hm_graph_evaluate_flags_dev")

int hm_graph_evaluate_segments_dev(uint64_t handle, const void* const* d_columns, size_t n_columns, const uint64_t* dynamic_constants,
                                   size_t n_dynamic, uint32_t log_segment, uint32_t segments, void* d_values, uint32_t flags, void* stream) try {
  return graph_evaluate_entry("hm_graph_evaluate_segments_dev", handle, d_columns, n_columns, dynamic_constants, n_dynamic, log_segment,
                              segments, d_values, flags, stream);
} HM_API_CATCH("hm_graph_evaluate_segments_dev")

// ---------------------------------------------------------------------------------------------
// The quotient h(X) of a proof in ONE call, from coefficient arrays: every column onto `count` cosets of the n-th roots
// (hm_coeff_to_cosets), the numerator program over count segments of n rows (hm_graph_evaluate_segments), the inverse
// transforms (hm_cosets_to_coeff), and the recombination with the vanishing division on its matrix -- what upstream's
// evaluate_h + divide_by_vanishing_poly + extended_to_coeff make of the extended arrays.
// ---------------------------------------------------------------------------------------------
// host side of the recombination: u_c = shift_c^n, V^-1 (Gauss-Jordan) with 1 / (u_c - 1) on column c -> rows of 4 * count words
static int quotient_matrix(const char* who, const uint64_t* shifts, size_t count, uint32_t log_n, std::vector<std::vector<uint64_t>>* rows) {
  std::vector<host::Fr4> u(count);
  for (size_t c = 0; c < count; ++c) {
    host::Fr4 x = host::fr_load(shifts + 4 * c);
    if (host::fr_is_zero(x)) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": a coset shift is zero");
    for (uint32_t b = 0; b < log_n; ++b) x = host::fr_mul(x, x);
    u[c] = x;
    if (host::fr_eq(u[c], host::FR_ONE)) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": a coset lies in the n-th roots (X^n - 1 vanishes on it)");
    for (size_t b = 0; b < c; ++b)
      if (host::fr_eq(u[b], u[c])) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": two shifts name the same coset");
  }
  const host::Fr4 zero = {{0, 0, 0, 0}};
  std::vector<std::vector<host::Fr4>> m(count, std::vector<host::Fr4>(2 * count, zero));      // [V | I] -> [I | V^-1]
  for (size_t a = 0; a < count; ++a) {
    host::Fr4 p = host::FR_ONE;
    for (size_t t = 0; t < count; ++t) { m[a][t] = p; p = host::fr_mul(p, u[a]); }
    m[a][count + a] = host::FR_ONE;
  }
  for (size_t col = 0; col < count; ++col) {
    size_t piv = col;
    while (piv < count && host::fr_is_zero(m[piv][col])) ++piv;
    if (piv == count) return hm_fail(HM_ERR_INTERNAL, std::string(who) + ": singular coset matrix");
    std::swap(m[col], m[piv]);
    const host::Fr4 inv = host::fr_inv(m[col][col]);
    for (auto& v : m[col]) v = host::fr_mul(v, inv);
    for (size_t row = 0; row < count; ++row) {
      if (row == col || host::fr_is_zero(m[row][col])) continue;
      const host::Fr4 f = m[row][col];
      for (size_t c2 = 0; c2 < 2 * count; ++c2) m[row][c2] = host::fr_sub(m[row][c2], host::fr_mul(f, m[col][c2]));
    }
  }
  rows->assign(count, std::vector<uint64_t>(4 * count));
  for (size_t c = 0; c < count; ++c) {
    const host::Fr4 tinv = host::fr_inv(host::fr_sub(u[c], host::FR_ONE));
    for (size_t t = 0; t < count; ++t) {
      const host::Fr4 v = host::fr_mul(m[t][count + c], tinv);
      std::memcpy(&(*rows)[t][4 * c], v.l, 32);
    }
  }
  return HM_OK;
}

// steps 1 - 3 on this device: the columns onto `count` cosets, the numerator over count segments, the inverse transforms ->
// d_partials (count x n).  ctx.mu held.
static int quotient_partials(const char* who, DeviceCtx& ctx, uint64_t program, const void* const* d_coeff_columns, const void* const* d_on_cosets,
                             size_t n_columns, const uint64_t* dynamic_constants, size_t n_dynamic, uint32_t log_n, const uint64_t omega[4],
                             const uint64_t* shifts, size_t count, void* d_partials, hipStream_t st) {
  std::vector<size_t> todo;
  for (size_t i = 0; i < n_columns; ++i) {
    if (d_on_cosets && d_on_cosets[i]) continue;
    if (!d_coeff_columns || !d_coeff_columns[i]) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": a column has neither coefficients nor coset values");
    todo.push_back(i);
  }
  const uint64_t n = 1ull << log_n;
  std::vector<host::Fr4> shift_inv(count);
  for (size_t c = 0; c < count; ++c) {
    const host::Fr4 x = host::fr_load(shifts + 4 * c);
    if (host::fr_is_zero(x)) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": a coset shift is zero");
    shift_inv[c] = host::fr_inv(x);
  }
  const host::Fr4 zero = {{0, 0, 0, 0}};
  host::Fr4 nn = host::FR_ONE;                                               // n = 2^log_n in Montgomery form: 1 doubled log_n times
  for (uint32_t b = 0; b < log_n; ++b) nn = host::fr_sub(nn, host::fr_sub(zero, nn));
  const host::Fr4 n_inv = host::fr_inv(nn);
  const host::Fr4 om_inv = host::fr_inv(host::fr_load(omega));
  GraphProgram* g = nullptr;
  for (auto& gp : ctx.graphs)
    if (gp->handle == program) g = gp.get();
  if (!g) return hm_fail(HM_ERR_NOT_FOUND, std::string(who) + ": unknown program handle");
  if (n_columns != g->n_columns) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": the program was built for another number of columns");
  AuxSlot* slot = aux_acquire(ctx, st);
  if (!slot) return HM_ERR_HIP;
  // work: [the columns to transform, side by side: T x n] [those columns on the cosets: T x count x n]
  const size_t row = (size_t)n * 32;
  const size_t T = todo.size();
  uint8_t* work = (uint8_t*)slot->work.ensure(row * (T + T * count) + 32);
  if (!work) return hm_fail(HM_ERR_HIP, std::string(who) + ": workspace allocation failed");
  uint8_t* contig = work;
  uint8_t* on_cosets = contig + row * T;
  int rc = HM_OK;
  if (T) {
    for (size_t j = 0; j < T; ++j)
      HM_HIP_CHECK(hipMemcpyAsync(contig + row * j, d_coeff_columns[todo[j]], row, hipMemcpyDeviceToDevice, st));
    rc = ntt_cosets_run(ctx, (const uint32_t*)contig, (uint32_t*)on_cosets, (uint32_t)T, omega, log_n, shifts, (uint32_t)count, true, st);
    if (rc != HM_OK) return rc;
    count_ntt(ctx, log_n, T * count);
  }
  std::vector<const void*> cols(n_columns);
  for (size_t i = 0; i < n_columns; ++i) cols[i] = d_on_cosets ? d_on_cosets[i] : nullptr;
  for (size_t j = 0; j < T; ++j) cols[todo[j]] = on_cosets + row * count * j;
  HM_HIP_CHECK(hipMemsetAsync(d_partials, 0, row * count, st));                // PreviousValue: upstream starts h at zero
  rc = graph_evaluate(ctx, *g, cols.data(), n_columns, dynamic_constants, n_dynamic, log_n, (uint32_t)count, d_partials, HM_GRAPH_COLUMNS_INTERNAL, st);
  if (rc != HM_OK) return rc;
  count_vector(ctx, HM_STAT_GRAPH_EVALUATE, 1, (uint64_t)count << log_n);
  rc = ntt_cosets_inverse_run(ctx, (uint32_t*)d_partials, (uint32_t)count, om_inv.l, log_n, n_inv.l, shift_inv[0].l, st);
  if (rc != HM_OK) return rc;
  count_ntt(ctx, log_n, count);
  return aux_release(ctx, slot, st);
}

static int quotient_check_args(const char* who, const void* out, const uint64_t* omega, const uint64_t* shifts, size_t n_columns,
                               const void* const* d_coeff_columns, const void* const* d_on_cosets, size_t n_dynamic,
                               const uint64_t* dynamic_constants, uint32_t log_n, size_t count) {
  if (!out || !omega || !shifts || (n_columns && !d_coeff_columns && !d_on_cosets) || (n_dynamic && !dynamic_constants))
    return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null argument");
  if (log_n > 28 || log_n == 0) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": log_n must be in 1 .. 28");
  if (count == 0 || count > 16) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": 1 .. 16 cosets per call");
  if (n_columns == 0 || n_columns * count > 65535) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": columns x cosets must be in 1 .. 65535");
  return HM_OK;
}

int hm_quotient_partials_bn256_fr_dev(uint64_t program, const void* const* d_coeff_columns, const void* const* d_on_cosets, size_t n_columns,
                                      const uint64_t* dynamic_constants, size_t n_dynamic, uint32_t log_n, const uint64_t omega[4],
                                      const uint64_t* shifts, size_t count, void* d_partials, void* stream) try {
  const char* who = "hm_quotient_partials_bn256_fr_dev";
  const int arc = quotient_check_args(who, d_partials, omega, shifts, n_columns, d_coeff_columns, d_on_cosets, n_dynamic, dynamic_constants, log_n, count);
  if (arc != HM_OK) return arc;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return quotient_partials(who, *ctx, program, d_coeff_columns, d_on_cosets, n_columns, dynamic_constants, n_dynamic, log_n, omega, shifts, count,
                           d_partials, (hipStream_t)stream);
} HM_API_CATCH("hm_quotient_partials_bn256_fr_dev")

int hm_quotient_combine_bn256_fr_dev(const void* const* d_partials, const uint64_t* shifts, size_t count, uint32_t log_n, size_t pieces, void* d_h,
                                     void* stream) try {
  const char* who = "hm_quotient_combine_bn256_fr_dev";
  if (!d_partials || !shifts || !d_h) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null argument");
  if (log_n > 28 || log_n == 0) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": log_n must be in 1 .. 28");
  if (count == 0 || count > 64 || pieces == 0 || pieces > count) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": 1 <= pieces <= cosets <= 64");
  for (size_t c = 0; c < count; ++c) {
    if (!d_partials[c]) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": null partial");
    // piece t is written before piece t + 1 reads EVERY partial again: h must not share memory with any of them
    if (ranges_overlap(d_h, ((size_t)32 << log_n) * pieces, d_partials[c], (size_t)32 << log_n))
      return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": d_h overlaps a partial (recombining in place is not supported)");
  }
  std::vector<std::vector<uint64_t>> rows;
  const int mrc = quotient_matrix(who, shifts, count, log_n, &rows);
  if (mrc != HM_OK) return mrc;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const uint64_t n = 1ull << log_n;
  for (size_t t = 0; t < pieces; ++t) {
    const int rc = fr_linear_combination_run(d_partials, rows[t].data(), count, n, (uint32_t*)((uint8_t*)d_h + (size_t)n * 32 * t), (hipStream_t)stream);
    if (rc != HM_OK) return rc;
  }
  count_vector(*ctx, HM_STAT_LINEAR_COMBINATION, pieces, (uint64_t)pieces * count * n);
  return HM_OK;
} HM_API_CATCH("hm_quotient_combine_bn256_fr_dev")

int hm_quotient_by_cosets_bn256_fr_dev(uint64_t program, const void* const* d_coeff_columns, const void* const* d_on_cosets, size_t n_columns,
                                       const uint64_t* dynamic_constants, size_t n_dynamic, uint32_t log_n, const uint64_t omega[4],
                                       const uint64_t* shifts, size_t count, size_t pieces, void* d_h, void* stream) try {
  const char* who = "hm_quotient_by_cosets_bn256_fr_dev";
  const int arc = quotient_check_args(who, d_h, omega, shifts, n_columns, d_coeff_columns, d_on_cosets, n_dynamic, dynamic_constants, log_n, count);
  if (arc != HM_OK) return arc;
  if (pieces == 0 || pieces > count) return hm_fail(HM_ERR_BAD_ARG, std::string(who) + ": 1 <= pieces <= cosets <= 16");
  std::vector<std::vector<uint64_t>> rows;
  const int mrc = quotient_matrix(who, shifts, count, log_n, &rows);
  if (mrc != HM_OK) return mrc;
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipStream_t st = (hipStream_t)stream;
  const uint64_t n = 1ull << log_n;
  const size_t row = (size_t)n * 32;
  // the partials live in the stream's NTT scratch slot? no: that is the transforms' ping-pong buffer.  They get the tail of a
  // buffer of their own (AuxSlot::table is the scans' scratch: free between calls on this stream)
  AuxSlot* slot = aux_acquire(*ctx, st);
  if (!slot) return HM_ERR_HIP;
  const size_t keep = (size_t)64 * 15 * 28 * 4;             // never shrink below the fixed-base table (see poly.hip)
  uint8_t* parts_buf = (uint8_t*)slot->table.ensure(row * count > keep ? row * count : keep);
  if (!parts_buf) return hm_fail(HM_ERR_HIP, std::string(who) + ": workspace allocation failed");
  int rc = quotient_partials(who, *ctx, program, d_coeff_columns, d_on_cosets, n_columns, dynamic_constants, n_dynamic, log_n, omega, shifts, count,
                             parts_buf, st);
  if (rc != HM_OK) return rc;
  std::vector<const void*> parts(count);
  for (size_t c = 0; c < count; ++c) parts[c] = parts_buf + row * c;
  for (size_t t = 0; t < pieces; ++t) {
    rc = fr_linear_combination_run(parts.data(), rows[t].data(), count, n, (uint32_t*)((uint8_t*)d_h + row * t), st);
    if (rc != HM_OK) return rc;
  }
  count_vector(*ctx, HM_STAT_LINEAR_COMBINATION, pieces, (uint64_t)pieces * count * n);
  return aux_release(*ctx, slot, st);
} HM_API_CATCH("hm_quotient_by_cosets_bn256_fr_dev")

int hm_graph_destroy(uint64_t handle) try {
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (size_t i = 0; i < ctx->graphs.size(); ++i)
    if (ctx->graphs[i]->handle == handle) {
      (void)hipDeviceSynchronize();          // a launch may still read the program (rare call: once per circuit)
      graph_release(*ctx->graphs[i]);
      ctx->graphs.erase(ctx->graphs.begin() + i);
      return HM_OK;
    }
  return hm_fail(HM_ERR_NOT_FOUND, "hm_graph_destroy: unknown program handle");
} HM_API_CATCH("hm_graph_destroy")

int hm_fr_dot_bn256_dev(const void* d_a, const void* d_b, size_t n, uint64_t out[4], void* stream) try {
  if (!out || (n && (!d_a || !d_b))) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_dot_bn256_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return fr_dot_run(*ctx, (const uint32_t*)d_a, (const uint32_t*)d_b, n, out, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_dot_bn256_dev")

int hm_fr_affine_sequence_dev(void* d_out, size_t n, const uint64_t a[4], const uint64_t b[4], void* stream) try {
  if ((n && !d_out) || !a || !b) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_affine_sequence_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return fr_affine_sequence_run((uint32_t*)d_out, n, a, b, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_affine_sequence_dev")

int hm_fr_random_dev(void* d_out, size_t n, uint64_t seed, void* stream) try {
  if (n && !d_out) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_random_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  return fr_random_run((uint32_t*)d_out, n, seed, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_random_dev")

int hm_fr_scale_dev(void* d_a, size_t n, const uint64_t c[4], void* stream) try {
  if ((n && !d_a) || !c) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_scale_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  if (n == 0) return HM_OK;
  return fr_scale_run((uint32_t*)d_a, c, n, (hipStream_t)stream);     // the constant travels by value: no shared state
} HM_API_CATCH("hm_fr_scale_dev")

int hm_fr_distribute_powers_dev(void* d_a, size_t n, const uint64_t c3[12], void* stream) try {
  if ((n && !d_a) || !c3) return hm_fail(HM_ERR_BAD_ARG, "hm_fr_distribute_powers_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  if (n == 0) return HM_OK;
  return fr_mul_pattern3_run((uint32_t*)d_a, c3, n, (hipStream_t)stream);
} HM_API_CATCH("hm_fr_distribute_powers_dev")

int hm_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, const uint64_t base_xy[8], void* d_out_xy, void* stream) try {
  if ((n && (!d_scalars || !d_out_xy)) || !base_xy) return hm_fail(HM_ERR_BAD_ARG, "hm_g1_fixed_base_mul_dev: null argument");
  DeviceCtx* ctx = ctx_for_current_device();
  if (!ctx) return HM_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return g1_fixed_base_mul_run(*ctx, (const uint32_t*)d_scalars, n, base_xy, (uint32_t*)d_out_xy, (hipStream_t)stream);
} HM_API_CATCH("hm_g1_fixed_base_mul_dev")

#ifdef HM_FAULT_INJECTION
// test build only (libhalo2_mi355x_fi.so; not declared in the public header): the (after + 1)-th passage through the
// named fault point throws std::runtime_error; point == NULL disarms
int hm_test_arm_fault(const char* point, long after) {
  std::lock_guard<std::mutex> lk(g_fault_mu);
  g_fault_name = point ? point : "";
  g_fault_after = point ? after : -1;
  return HM_OK;
}
#endif

}  // extern "C"
