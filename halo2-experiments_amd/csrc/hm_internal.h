// hm_internal.h -- private declarations shared by the translation units of libhalo2_mi355x.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/halo2_mi355x.h"

namespace hm {

int hm_fail(int code, const std::string& what);  // records the message for hm_last_error(), returns code

#define HM_HIP_CHECK(expr)                                                                          \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return ::hm::hm_fail(HM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
  } while (0)
#define HM_HIP_CHECK_PTR(expr)                                                                      \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      ::hm::hm_fail(HM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return nullptr;                                                                               \
    }                                                                                               \
  } while (0)

// ---- the exception barrier of the C ABI ------------------------------------------------------------
// include/halo2_mi355x.h promises "never aborts or throws across the boundary": every extern "C" entry is a
// function-try-block closed by HM_API_CATCH, so a std::bad_alloc / std::system_error / anything else raised under
// it (vector growth, std::string, std::thread creation) comes back as HM_ERR_INTERNAL with the message in
// hm_last_error().  Locks are std::lock_guard (released by the unwinding), helper threads are owned by JoinOnExit.
int hm_guard_fail(const char* entry, const char* what) noexcept;
#define HM_API_CATCH(entry)                                                                         \
  catch (const std::exception& e) { return ::hm::hm_guard_fail(entry, e.what()); }                  \
  catch (...) { return ::hm::hm_guard_fail(entry, nullptr); }

// Fault points: compiled in only for the test build (libhalo2_mi355x_fi.so, -DHM_FAULT_INJECTION), where
// hm_test_arm_fault(point, after) makes the (after + 1)-th passage through the named point throw.  In the product
// build the call is an empty inline function.
#ifdef HM_FAULT_INJECTION
void hm_fault_point(const char* point);
#else
inline void hm_fault_point(const char*) {}
#endif

// Helper threads of a call (digest lanes, per-device workers, the batch waiter): joined on every way out of the
// scope that owns them -- a throw between two spawns must not destroy a joinable std::thread (std::terminate).
struct JoinOnExit {
  std::vector<std::thread> th;
  ~JoinOnExit() {
    for (auto& t : th)
      if (t.joinable()) t.join();
  }
};
// Start `f` on a new thread owned by `pool`; false when no thread can be had (the caller then runs f itself).
template <class F>
inline bool spawn_or_false(JoinOnExit& pool, const char* point, F&& f) {
  try {
    hm_fault_point(point);
    pool.th.emplace_back(std::forward<F>(f));
    return true;
  } catch (...) {
    return false;
  }
}
const std::string& hm_last_error_string();   // the calling thread's message (workers carry theirs back to the caller's thread)

struct NttTables {
  uint32_t log_n = 0;
  uint64_t omega[4] = {0, 0, 0, 0};
  uint32_t log_lb = 0;
  // The table kernels run on the stream of the call that first needed them.  `ready` is recorded behind
  // them; until it has completed (`published`), a caller on another stream waits for it on the device.
  hipStream_t build_stream = nullptr;
  hipEvent_t ready = nullptr;
  bool published = false;
  uint32_t* d_lo = nullptr;
  uint32_t* d_hi = nullptr;
  uint32_t* d_mid = nullptr;   // direct twiddles of the middle pass of a 3-pass plan
  uint32_t* d_stage[16] = {};
  size_t bytes = 0;            // device memory of all of the above
  uint64_t last_use = 0;       // ctx.ntt_table_clock at the last ntt_get_tables hit (LRU)
};

// The powers table of a transform on a coset shift * <omega> (ntt.hip: coset_table_get): n external words holding
// 32 * shift^i (or 1024 * shift^i for outputs in the evaluator's internal column form).  Built on the first caller's
// stream and published to the others behind `ready`, like the twiddle tables.
struct CosetTable {
  uint64_t shift[4] = {0, 0, 0, 0};
  uint32_t log_n = 0;
  uint32_t internal = 0;
  uint32_t* d = nullptr;
  hipStream_t build_stream = nullptr;
  hipEvent_t ready = nullptr;
  bool published = false;
  uint64_t last_use = 0;
};

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* ensure(size_t bytes) {
    if (bytes <= cap) return p;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    cap = bytes;
    return p;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// xfer.hip: host <-> device copies of the host-pointer entry points through pinned staging the library owns (never the caller's
// pages: see the file's head).  One transfer at a time per device (mu); lanes are created on first use and kept.
constexpr int HM_XFER_LANES = 8;
struct XferLane {
  void* pin[2] = {nullptr, nullptr};      // 2 MiB each, carved from HostXfer::block
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  double t_wait_us = 0, t_copy_us = 0, t_issue_us = 0, t_start_us = 0;    // of the lane's last share: waiting for DMAs / host memcpy (HALO2_MI355X_XFER_TRACE)
};
struct HostXfer {
  std::mutex mu;
  XferLane lanes[HM_XFER_LANES];
  int ready = 0;
  void* block = nullptr;                  // ONE pinned allocation for every lane's slots (16 separate ones took 49 ms to make)
  void* d_warm = nullptr;                 // 4 KiB of device memory: a new lane's stream is used once in both directions when it is made
  std::atomic<uint64_t> direct{0}, staged{0};   // copies handed to hipMemcpy on the caller's pointers / moved through the lanes (policy: xfer.hip)
};

struct BasesEntry {          // device-resident, converted base set (hm_register_bases)
  uint64_t handle = 0;
  size_t n = 0;
  uint32_t* d_xy = nullptr;  // n x 16 u32: x, y packed internal form
  uint8_t* d_inf = nullptr;  // n flags: base is the identity
  uint32_t pc_c = 0;         // != 0: d_xy holds pc_W * n points, entry j*n + i = 2^(pc_c * j) * P_i
  uint32_t pc_W = 0;
  size_t xy_bytes = 0, inf_bytes = 0;
};

struct MsmStats {
  double t_digits_ms = 0, t_sort_ms = 0, t_accum_ms = 0, t_reduce_ms = 0, t_total_ms = 0, t_accum_kernel_ms = 0;
  uint64_t pairs = 0, tasks = 0;
  uint32_t c = 0, windows = 0;
};

constexpr int HM_MSM_GROUP = 8;   // scalar arrays one launch chain of the five-launch plan carries (msm_small.hip)
#ifndef HM_MSM_SLOTS_N
#define HM_MSM_SLOTS_N 9
#endif
constexpr int HM_MSM_SLOTS = HM_MSM_SLOTS_N;   // slot 0: synchronous calls; 1..8: asynchronous tickets (workspaces allocated on first use)

struct MsmSlot {            // one in-flight MSM: its workspace, events and host landing buffers
  DevBuf ws;
  hipEvent_t ev[7] = {};
  bool ev_ready = false;
  bool busy = false;        // held by a ticket of hm_msm_submit_dev
  bool awaiting = false;    // a thread is inside hm_msm_wait for this ticket (it waits and folds without the context lock)
  bool phase_timed = false; // the MSM in flight recorded its per-phase events (ev[1..3], ev[5..6]), not only ev[0] / ev[4]
  uint64_t ticket = 0;
  uint64_t bases_handle = 0;   // base set the MSM in flight reads (a released set is freed only after its last ticket)
  size_t n = 0;
  hipStream_t stream = nullptr;
  uint32_t* h_land = nullptr;   // pinned host landing zone: 4 totals words, then the window sums (as on the device)
  uint32_t* totals() { return h_land; }
  uint32_t* win() { return h_land + 4; }
  uint32_t SW = 0, c = 0, W = 0;
  uint32_t group = 1;       // MSMs the chain in flight carries (msm_small.hip groups); their results land res_stride words apart
  uint32_t res_stride = 0;
  const uint32_t* live_ptr = nullptr;   // where the last five-launch chain left its (zeroed) block counters; null after any other use of ws
  const uint32_t* d_win = nullptr;   // device locations of the results of the MSM in flight (inside ws)
  const uint32_t* d_tot = nullptr;
  uint64_t T_max = 0;
  bool balanced = false;             // windows of unequal widths (msm_small.hip): the host fold shifts by win_bits[w]
  uint8_t win_bits[128] = {};
};

// Transient device state of the stream-asynchronous entry points (NTT ping-pong buffer, fixed-base
// table): one slot per stream in use, so that calls on different streams never share a buffer.  A slot
// is handed to a new stream only behind the event recorded after its previous user's launches.
struct AuxSlot {
  DevBuf scratch;                 // NTT ping-pong buffer
  DevBuf table;                   // multiples table of hm_g1_fixed_base_mul_dev
  DevBuf args;                    // per-call argument block of hm_graph_evaluate_dev (column table + per-proof constants)
  DevBuf work;                    // hm_quotient_by_cosets_bn256_fr_dev: the columns on the cosets, the numerator's values
  hipStream_t stream = nullptr;   // stream of the last user
  hipEvent_t done = nullptr;      // recorded behind the last user's launches
  bool used = false;
  uint64_t last_use = 0;
};
constexpr int HM_AUX_SLOTS = 4;

struct GraphVariant {       // the program lowered for one column format (graph.hip: graph_lower)
  void* d_calcs = nullptr;
  uint32_t n_calc = 0, n_slots = 1, result_src = 0, result_prev = 0;
  bool ready = false;
};
struct GraphProgram {       // a compiled evaluate_h program (graph.hip): the validated program + its lowered device forms
  uint64_t handle = 0;
  void* d_blob = nullptr;     // constants + rotations
  uint32_t* d_consts = nullptr;
  int32_t* d_rot = nullptr;
  std::vector<uint32_t> calcs5;      // as given to hm_graph_create
  uint32_t n_intermediates = 0, n_static = 0, n_dynamic = 0;
  size_t n_columns = 0;
  GraphVariant variant[2];           // [0] external-form columns, [1] internal-form columns (HM_GRAPH_COLUMNS_INTERNAL)
};

struct FreeBases {          // buffers of a released base set, kept for the next registration of that size
  uint32_t* d_xy = nullptr;
  uint8_t* d_inf = nullptr;
  size_t xy_bytes = 0, inf_bytes = 0;
};

// per-call counters (hm_get_stats): what the shim of INTEGRATION.md reads to replace call-trace estimates
struct CallStats {
  uint64_t msm_calls = 0, msm_points = 0, ntt_calls = 0, ntt_elements = 0;
  uint64_t msm_by_log[32] = {}, ntt_by_log[32] = {};
  double msm_h2d_us = 0, msm_device_us = 0, msm_host_us = 0, ntt_h2d_us = 0, ntt_device_us = 0, ntt_d2h_us = 0;
  uint64_t h2d_bytes = 0, d2h_bytes = 0;
  uint64_t vector_calls[8] = {}, vector_elements[8] = {};    // HM_STAT_* kinds (include/halo2_mi355x.h)
};

struct DeviceCtx {
  int device = 0;
  std::mutex mu;
  std::vector<std::unique_ptr<NttTables>> ntt_tables;   // LRU, bounded by count and bytes (ntt.hip: kNttTablesMax, kNttTableBytesMax)
  size_t ntt_table_bytes = 0;
  uint64_t ntt_table_clock = 0;
  std::vector<std::unique_ptr<CosetTable>> coset_tables;
  uint64_t coset_clock = 0;
  size_t coset_table_bytes = 0;        // HBM held by coset_tables (capped: ntt.hip kCosetTableBytesMax)
  AuxSlot aux[HM_AUX_SLOTS];
  uint64_t aux_clock = 0;
  HostXfer xfer;          // pinned staging lanes of the host-pointer forms' copies (xfer.hip)
  DevBuf io;              // staging for host-pointer calls (scalars / NTT array)
  DevBuf io_bases;        // staging for raw external bases of host-pointer MSM calls
  DevBuf conv_bases;      // converted bases of un-registered calls
  DevBuf conv_inf;
  MsmSlot msm_slots[HM_MSM_SLOTS];   // MSM workspaces (digits, sort scratch, sorted indices, bucket sums ...)
  uint64_t next_ticket = 1;
  std::vector<BasesEntry> bases;
  std::vector<BasesEntry> zombie_bases;   // released while a ticket still reads them: freed by the last hm_msm_wait
  std::vector<FreeBases> free_bases;      // recycled buffers (no hipFree => no device-wide synchronisation)
  std::vector<std::unique_ptr<GraphProgram>> graphs;
  hipStream_t batch_streams[HM_MSM_SLOTS - 1] = {};   // hm_msm_batch_bn256_g1_dev: one per asynchronous slot
  hipEvent_t batch_event = nullptr;
  bool batch_streams_ready = false;
  DevBuf batch_io[HM_MSM_SLOTS - 1];                  // hm_msm_batch_bn256_g1_h: per-lane staging of host scalar arrays
  DevBuf live_io;                                     // msm_count_live_blocks: column pointers in, block counts out
  std::mutex live_mu;                                 // ... one counting call at a time (never taken under mu)
  std::mutex batch_h_mu;                              // ... which belong to ONE _h batch call at a time (taken before mu, never under it)
  uint64_t next_handle = 1;
  // drop-in MSM: the converted bases of the previous call, keyed by a digest of the WHOLE host array (capi.hip)
  size_t cached_host_n = 0;
  const void* cached_xy = nullptr;
  uint64_t cached_digest[4] = {0, 0, 0, 0};
  MsmStats last_msm;
  CallStats calls;
  bool msm_attr_set = false, ntt_attr_set = false, msm_small_attr_set = false, lookup_attr_set = false;
};

// Slot for a call on `stream` (ctx.mu held): the stream's own slot, else a free or finished one, else the
// least recently used one behind a device-side wait for its event.  aux_release records the event.
AuxSlot* aux_acquire(DeviceCtx& ctx, hipStream_t stream);
int aux_release(DeviceCtx& ctx, AuxSlot* slot, hipStream_t stream);

DeviceCtx* ctx_for_current_device();

// xfer.hip.  Synchronous; the device side must be idle on the range (the callers synchronise their stream first).  May be called
// with or without ctx.mu held (takes ctx.xfer.mu, never ctx.mu).  d2h failures leave `dst` partly written.
int xfer_h2d(DeviceCtx& ctx, void* d_dst, const void* src, size_t bytes, const char* who);
int xfer_d2h(DeviceCtx& ctx, void* dst, const void* d_src, size_t bytes, const char* who);
// `count` arrays as ONE job of the lanes (hm_copy_many_to_device / _to_host): dev[i] <-> host[i], bytes[i] each
int xfer_many(DeviceCtx& ctx, bool up, void* const* dev, void* const* host, const size_t* bytes, size_t count, const char* who);
void xfer_release(DeviceCtx& ctx);
void xfer_prefault(void* p, size_t bytes);      // first-touch a fresh destination from helper threads (contents kept)
int xfer_set_policy(int mode);                  // 0 auto, 1 lanes, 2 direct; -1 on anything else
int xfer_mode(const void* host, size_t bytes);  // which way a copy of this host range goes NOW: 0 = hipMemcpy on the caller's pointers, 1 = the library's pinned lanes
int xfer_host_register(const void* p, size_t bytes);   // hm_host_register / hm_host_unregister: ranges the caller declares long-lived
int xfer_host_unregister(const void* p);
size_t xfer_host_ranges();

// capi.hip: the one-device bodies the multi-device layer runs per part (Jacobian results, so that partials fold)
int msm_h_local(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n, uint64_t jac[12], int* is_id);
std::vector<int> msm_device_list();            // copy of hm_set_msm_devices' list (empty: one device)
constexpr size_t kMinShardPoints = 1 << 14;    // below this many points per device a split only adds latency
constexpr size_t kSliceBasesFrom = (size_t)1 << 22;   // base sets from this size are SLICED over the devices, smaller ones replicated

// multi.hip: single-process multi-GPU layer over the one-device entry points (hm_set_msm_devices).  A handle with
// HM_MULTI_HANDLE_BIT names a base set registered on several devices; every form that takes a handle dispatches on it.
constexpr uint64_t HM_MULTI_HANDLE_BIT = 1ull << 62;
inline bool is_multi_handle(uint64_t h) { return (h & HM_MULTI_HANDLE_BIT) != 0; }
// fn(r) for every part r, part r on device devs[r] from its own host thread (the calling thread runs the parts no thread
// could be had for); the first failure is reported on the caller's thread.  Nothing escapes a worker.
int run_per_device(const std::vector<int>& devs, const std::function<int(size_t)>& fn);
bool& multi_worker_flag();                     // true on a thread that is running a part (registrations there stay on one device)
int multi_register(const uint64_t* bases_host, const void* d_bases, size_t n, void* stream, int layout, const std::vector<int>& devs,
                   uint64_t* out_handle);
int multi_bases_info(uint64_t handle, hm_bases_info* out);   // sums over the parts
int multi_release(uint64_t handle);
void multi_release_touching(int device);       // hm_shutdown: drop every multi handle with a part on `device`
int multi_msm(uint64_t handle, size_t offset, const void* scalars, bool from_host, size_t n, void* stream, uint64_t jac[12], int* is_id);
int multi_msm_batch(uint64_t handle, size_t offset, const void* const* scalars, bool from_host, size_t n, size_t count, void* stream,
                    uint64_t* out_xyz);
int multi_local_part(uint64_t handle, int device, uint64_t* local_handle);   // replicated sets: the copy on `device`

// ntt.hip
// scale / coset: optional external (4 x u64 Montgomery) constants on the HOST; they travel to the kernels
// by value, so no call shares a constants buffer with another.  post3: optional {1, c, c^2}-style pattern
// multiplied into element i of every output array by i % 3 (EvaluationDomain::extended_to_coeff).
struct NttFused {
  const uint64_t* scale = nullptr;   // 4 words: multiply every output (ifft divisor)
  const uint64_t* coset = nullptr;   // 12 words: input element i *= coset[i % 3] (coeff_to_extended)
  const uint64_t* post3 = nullptr;   // 12 words: output element i *= post3[i % 3] (extended_to_coeff)
  const uint32_t* d_in_scale = nullptr;   // DEVICE table, n x 8 words: input element i *= table[i] / 32 (the transform on a coset)
  // several cosets of the SAME inputs in one launch chain: n_in_scales tables; output array (b, c) = number b * n_in_scales + c
  const uint32_t* d_in_scales[16] = {};
  uint32_t n_in_scales = 0;
};
constexpr uint32_t HM_NTT_COSETS_MAX = 16;
int ntt_run(DeviceCtx& ctx, uint32_t* d_a, const uint64_t omega_ext[4], uint32_t log_n, uint32_t batch,
            const NttFused& fused, hipStream_t stream, const uint32_t* d_in = nullptr, uint32_t log_z = 0);
int ntt_plan_first_digit(uint32_t log_n, int* passes);
int fr_scale_run(uint32_t* d_a, const uint64_t c_ext[4], uint64_t n, hipStream_t stream);
int fr_mul_pattern3_run(uint32_t* d_a, const uint64_t c3_ext[12], uint64_t n, hipStream_t stream);
void ntt_tables_release(NttTables& t);
void coset_tables_release(DeviceCtx& ctx);
size_t ntt_caches_give_back(DeviceCtx& ctx);   // every cached twiddle / coset table freed (after a device synchronise); bytes given back
// out_b[t] = sum_i in_b[i] (shift omega^t)^i for `batch` back-to-back n-element arrays, out of place (d_out may be d_in);
// internal: outputs multiplied by 32 (HM_GRAPH_COLUMNS_INTERNAL)
int ntt_coset_run(DeviceCtx& ctx, const uint32_t* d_in, uint32_t* d_out, uint32_t batch, const uint64_t omega_ext[4], uint32_t log_n,
                  const uint64_t shift_ext[4], bool internal, hipStream_t stream);
// the same for `count` cosets at once: d_out holds batch * count arrays, array b * count + c = input b on coset shifts[c]
int ntt_cosets_run(DeviceCtx& ctx, const uint32_t* d_in, uint32_t* d_out, uint32_t batch, const uint64_t omega_ext[4], uint32_t log_n,
                   const uint64_t* shifts_ext, uint32_t count, bool internal, hipStream_t stream);
// in place, `count` arrays: a_c <- divisor * inverse transform of a_c, then a_c[i] *= shift_invs[c]^i
int ntt_cosets_inverse_run(DeviceCtx& ctx, uint32_t* d_a, uint32_t count, const uint64_t omega_inv_ext[4], uint32_t log_n,
                           const uint64_t divisor_ext[4], const uint64_t* shift_invs_ext, hipStream_t stream);
// in place: a_b <- divisor * inverse transform of a_b, then a_b[i] *= shift_inv^i
int ntt_coset_inverse_run(DeviceCtx& ctx, uint32_t* d_a, uint32_t batch, const uint64_t omega_inv_ext[4], uint32_t log_n,
                          const uint64_t divisor_ext[4], const uint64_t shift_inv_ext[4], hipStream_t stream);

// graph.hip
int graph_create(DeviceCtx& ctx, const uint32_t* calcs5, size_t n_calc, const uint64_t* constants_ext, size_t n_const_static,
                 size_t n_dynamic, const int32_t* rotations, size_t n_rot, size_t n_columns, uint32_t n_intermediates,
                 uint64_t* out_handle);
int graph_evaluate(DeviceCtx& ctx, GraphProgram& g, const void* const* d_columns, size_t n_columns, const uint64_t* dyn_ext,
                   size_t n_dyn, uint32_t log_size, uint32_t segments, void* d_values, uint32_t flags, hipStream_t stream);
void graph_release(GraphProgram& g);

// poly.hip
int fr_powers_run(uint32_t* d_out, uint64_t n, const uint64_t x_ext[4], hipStream_t stream);
int fr_dot_run(DeviceCtx& ctx, const uint32_t* d_a, const uint32_t* d_b, uint64_t n, uint64_t out_ext[4], hipStream_t stream);
int fr_affine_sequence_run(uint32_t* d_out, uint64_t n, const uint64_t a_ext[4], const uint64_t b_ext[4], hipStream_t stream);
int fr_random_run(uint32_t* d_out, uint64_t n, uint64_t seed, hipStream_t stream);
int fr_eval_polynomial_run(DeviceCtx& ctx, const uint32_t* d_polys, uint64_t n, const uint32_t* poly_index, const uint64_t* points_ext,
                           size_t q, uint64_t* out_ext, hipStream_t stream);

// polyops.hip
int fr_kate_division_run(DeviceCtx& ctx, const uint32_t* d_a, uint64_t n, const uint64_t z_ext[4], uint32_t* d_q, hipStream_t stream);
int fr_grand_product_run(DeviceCtx& ctx, const uint32_t* d_m, uint64_t n, const uint64_t start_ext[4], uint32_t* d_out, hipStream_t stream);
int fr_kate_division_batch_run(DeviceCtx& ctx, const void* const* d_a, uint64_t n, const uint64_t* z_ext, void* const* d_q, size_t count,
                               hipStream_t stream);
int fr_grand_product_batch_run(DeviceCtx& ctx, const void* const* d_m, uint64_t n, const uint64_t start_ext[4], uint64_t chain_row,
                               void* const* d_out, size_t count, hipStream_t stream);
int fr_batch_invert_run(uint32_t* d_v, uint64_t n, hipStream_t stream);
int fr_mul_periodic_run(uint32_t* d_a, uint64_t n, const uint64_t* pattern_ext, uint32_t period, hipStream_t stream);
int fr_linear_combination_run(const void* const* d_polys, const uint64_t* coeffs_ext, size_t count, uint64_t n, uint32_t* d_out,
                              hipStream_t stream);

// lookup.hip
int lookup_permute_run(DeviceCtx& ctx, const void* const* d_inputs, const void* const* d_tables, size_t pairs, uint64_t rows,
                       void* const* d_out_inputs, void* const* d_out_tables, int* missing, hipStream_t stream);

// msm.hip
int msm_convert_bases(const uint32_t* d_bases_ext, uint32_t* d_xy, uint8_t* d_inf, size_t n, hipStream_t stream);
// out_windows: host buffer of W x 12 u64 external Jacobian + flags
int msm_enqueue(DeviceCtx& ctx, int slot, const uint32_t* d_scalars_ext, const uint32_t* d_xy, const uint8_t* d_inf, size_t n,
                uint32_t precomp_c, hipStream_t stream);
// `group` dense MSMs over one fixed-base table through one chain of the general pipeline (msm.hip); 1 when it does not apply
uint32_t msm_table_group_max(size_t n, uint32_t precomp_c);
int msm_enqueue_table_group(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                            const uint8_t* d_inf, size_t n, uint32_t precomp_c, hipStream_t stream);
int msm_finish(DeviceCtx& ctx, int slot, uint64_t out_jac_ext[12], int* out_is_identity);
// the two halves of msm_finish: the blocking part touches only the slot (callable without ctx.mu while the slot is
// busy), the bookkeeping part runs under ctx.mu
int msm_finish_wait_fold(MsmSlot& sl, uint64_t* out_jac_ext /* 12 words per MSM of the group */, int* out_is_identity /* one per MSM */,
                         double* host_us);
void msm_finish_record(DeviceCtx& ctx, int slot, double host_us);
bool msm_phase_timing(bool small_plan);    // hm_msm_set_phase_timing: -1 auto (general pipeline only), 0 never, 1 always
int msm_slot_prepare(MsmSlot& sl);   // events + pinned landing zone, on first use
int msm_launch_digits(const uint32_t* d_scalars_ext, const uint8_t* d_inf, int32_t* d_digits, size_t n, uint32_t c, uint32_t W,
                      hipStream_t stream);
// msm_small.hip: the five-launch chain for n < 2^19 (plain base sets, c <= 15)
bool msm_small_applies(size_t n, uint32_t c, bool single_set);
int msm_issue_small(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                    const uint8_t* d_inf, size_t n, uint32_t c, hipStream_t stream);
// live_out[i] = how many 256-row blocks of column i hold a row that survives the digits kernel's compaction (a non-zero
// scalar on a non-identity base); synchronises `stream` (one small copy).  Decides the grouping of a phase of commitments.
int msm_count_live_blocks(DeviceCtx& ctx, const void* const* d_columns, size_t count, const uint8_t* d_inf, size_t n, hipStream_t stream,
                          uint32_t* live_out);
// a group of MSMs over the same points through one chain (only where the five-launch plan applies: msm_group_applies)
bool msm_group_applies(size_t n, uint32_t precomp_c);
int msm_enqueue_group(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                      const uint8_t* d_inf, size_t n, hipStream_t stream, size_t live_rows = 0);
int msm_run(DeviceCtx& ctx, const uint32_t* d_scalars_ext, const uint32_t* d_xy, const uint8_t* d_inf, size_t n,
            uint32_t precomp_c, uint64_t out_jac_ext[12], int* out_is_identity, hipStream_t stream);
uint32_t msm_precomp_window(size_t n);
int msm_precompute(uint32_t* d_table, const uint8_t* d_inf, size_t n, uint32_t c, uint32_t W, hipStream_t stream);
void host_sum_points(const uint64_t* pts, size_t count, uint64_t out_jac_ext[12], int* out_is_identity);
int g1_fixed_base_mul_run(DeviceCtx& ctx, const uint32_t* d_scalars_ext, size_t n, const uint64_t base_affine_ext[8],
                          uint32_t* d_out_affine_ext, hipStream_t stream);

}  // namespace hm
