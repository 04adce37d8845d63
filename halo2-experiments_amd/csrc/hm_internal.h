// hm_internal.h -- private declarations shared by the translation units of libhalo2_mi355x.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <string>
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

struct NttTables {
  uint32_t log_n = 0;
  uint64_t omega[4] = {0, 0, 0, 0};
  uint32_t log_lb = 0;
  void* d_omega = nullptr;
  uint32_t* d_lo = nullptr;
  uint32_t* d_hi = nullptr;
  uint32_t* d_mid = nullptr;   // direct twiddles of the middle pass of a 3-pass plan
  uint32_t* d_stage[16] = {};
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

struct BasesEntry {          // device-resident, converted base set (hm_register_bases)
  uint64_t handle = 0;
  size_t n = 0;
  uint32_t* d_xy = nullptr;  // n x 16 u32: x, y packed internal form
  uint8_t* d_inf = nullptr;  // n flags: base is the identity
  uint32_t pc_c = 0;         // != 0: d_xy holds pc_W * n points, entry j*n + i = 2^(pc_c * j) * P_i
  uint32_t pc_W = 0;
};

struct MsmStats {
  double t_digits_ms = 0, t_sort_ms = 0, t_accum_ms = 0, t_reduce_ms = 0, t_total_ms = 0, t_accum_kernel_ms = 0;
  uint64_t pairs = 0, tasks = 0;
  uint32_t c = 0, windows = 0;
};

constexpr int HM_MSM_SLOTS = 9;   // slot 0: synchronous calls; 1..8: asynchronous tickets (workspaces allocated on first use)

struct MsmGraphKey {        // what the captured launch sequence (everything after the digit kernel) depends on
  size_t n = 0;
  const void* d_xy = nullptr;
  uint32_t precomp_c = 0;
  int window_override = 0;
  const void* ws = nullptr;
  bool operator==(const MsmGraphKey& o) const {
    return n == o.n && d_xy == o.d_xy && precomp_c == o.precomp_c && window_override == o.window_override && ws == o.ws;
  }
};

struct MsmSlot {            // one in-flight MSM: its workspace, events and host landing buffers
  DevBuf ws;
  struct Graph {                     // captured launch sequence of one small-MSM shape on this slot
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    MsmGraphKey key;
    uint64_t last_use = 0;
  };
  static constexpr int kGraphs = 4;  // e.g. {g, g_lagrange} x {one or two sizes}; least recently used is replaced
  Graph graphs[kGraphs];
  uint64_t graph_clock = 0;
  bool timed = false;                // per-phase events were recorded for the MSM in flight
  hipEvent_t ev[7] = {};
  bool ev_ready = false;
  bool busy = false;        // held by a ticket of hm_msm_submit_dev
  uint64_t ticket = 0;
  size_t n = 0;
  hipStream_t stream = nullptr;
  uint32_t* h_land = nullptr;   // pinned host landing zone: 4 totals words, then the window sums (as on the device)
  uint32_t* totals() { return h_land; }
  uint32_t* win() { return h_land + 4; }
  uint32_t SW = 0, c = 0, W = 0;
  const uint32_t* d_win = nullptr;   // device locations of the results of the MSM in flight (inside ws)
  const uint32_t* d_tot = nullptr;
  uint64_t T_max = 0;
};

struct DeviceCtx {
  int device = 0;
  std::mutex mu;
  std::vector<std::unique_ptr<NttTables>> ntt_tables;
  DevBuf scratch;         // NTT ping-pong buffer
  DevBuf io;              // staging for host-pointer calls (scalars / NTT array)
  DevBuf io_bases;        // staging for raw external bases of host-pointer MSM calls
  DevBuf conv_bases;      // converted bases of un-registered calls
  DevBuf conv_inf;
  MsmSlot msm_slots[HM_MSM_SLOTS];   // MSM workspaces (digits, sort scratch, sorted indices, bucket sums ...)
  uint64_t next_ticket = 1;
  DevBuf small;           // small constants
  std::vector<BasesEntry> bases;
  uint64_t next_handle = 1;
  // host-pointer MSM: cache of the last converted un-registered base array (keyed by ptr,len,probe)
  const void* cached_host_bases = nullptr;
  size_t cached_host_n = 0;
  uint64_t cached_probe[4] = {0, 0, 0, 0};
  MsmStats last_msm;
  bool msm_attr_set = false, ntt_attr_set = false;
  hipStream_t capture_stream = nullptr;   // launch sequences are captured here, replayed on the caller's stream
  void* ensure_scratch(size_t bytes) { return scratch.ensure(bytes); }
};

DeviceCtx* ctx_for_current_device();

// ntt.hip
int ntt_run(DeviceCtx& ctx, uint32_t* d_a, const uint64_t omega_ext[4], uint32_t log_n, uint32_t batch,
            const uint32_t* d_scale_int, const uint32_t* d_coset_int, hipStream_t stream, const uint32_t* d_in = nullptr,
            uint32_t log_z = 0);
int ntt_plan_first_digit(uint32_t log_n, int* passes);
int fr_scale_run(uint32_t* d_a, const uint32_t* d_c_ext, uint64_t n, hipStream_t stream);
int fr_mul_pattern3_run(uint32_t* d_a, const uint32_t* d_c3_ext, uint64_t n, hipStream_t stream);
int fr_ext_to_int_run(const uint32_t* d_c_ext, uint32_t* d_out, uint32_t count, hipStream_t stream);

// msm.hip
int msm_convert_bases(const uint32_t* d_bases_ext, uint32_t* d_xy, uint8_t* d_inf, size_t n, hipStream_t stream);
// out_windows: host buffer of W x 12 u64 external Jacobian + flags
int msm_enqueue(DeviceCtx& ctx, int slot, const uint32_t* d_scalars_ext, const uint32_t* d_xy, const uint8_t* d_inf, size_t n,
                uint32_t precomp_c, hipStream_t stream);
int msm_finish(DeviceCtx& ctx, int slot, uint64_t out_jac_ext[12], int* out_is_identity);
int msm_run(DeviceCtx& ctx, const uint32_t* d_scalars_ext, const uint32_t* d_xy, const uint8_t* d_inf, size_t n,
            uint32_t precomp_c, uint64_t out_jac_ext[12], int* out_is_identity, hipStream_t stream);
void msm_set_use_graphs(bool on);
void msm_slot_release_graph(MsmSlot& sl);
uint32_t msm_precomp_window(size_t n);
int msm_precompute(uint32_t* d_table, const uint8_t* d_inf, size_t n, uint32_t c, uint32_t W, hipStream_t stream);
void host_sum_points(const uint64_t* pts, size_t count, uint64_t out_jac_ext[12], int* out_is_identity);
int g1_fixed_base_mul_run(DeviceCtx& ctx, const uint32_t* d_scalars_ext, size_t n, const uint64_t base_affine_ext[8],
                          uint32_t* d_out_affine_ext, hipStream_t stream);

}  // namespace hm
