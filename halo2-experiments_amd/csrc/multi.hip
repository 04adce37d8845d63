// multi.hip -- the single-process multi-GPU layer of the C ABI (hm_set_msm_devices).
//
// The reference's prover is ONE process (/root/reference/src/circuits/utils.rs:22-70: full_prover calls create_proof
// at :40-48), so a Rust caller has no torch.distributed to deal commitments with: the split has to live under the
// handle and batch entry points it binds (INTEGRATION.md §3 / §3b route ParamsKZG::commit* to them).  With a device list
// set, hm_register_bases* returns a MULTI handle:
//   n <  2^22  (every circuit of the reference: k = 11 / 17 / 18)   the set is REPLICATED on every listed device; a
//              phase of commitments (hm_msm_batch_bn256_g1_h / _dev) is dealt round-robin as WHOLE commitments, one host
//              thread per device running the ordinary one-device batch (eight in flight there), results in call order
//   n >= 2^22  (the 2^24 .. 2^26 microbenchmark)   the set is SLICED by contiguous index ranges, device r keeps
//              [r n / N, (r + 1) n / N); every MSM runs its range on every device and the 96-byte partials are folded on
//              the host (EC addition is not a reduction operator of any collective library)
// A single hm_msm_bn256_g1_h / _dev call splits by index range in both modes (replicated sets: only when every part
// keeps >= 2^14 points).  The data path has no inter-GPU exchange except where the CALLER's scalars live on one device
// and the work on another (device-pointer forms): those slices cross xGMI once, by hipMemcpyPeer.
// A device may be listed several times (tests drive one GPU as three "devices"): every listing is a part of its own.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>

#include "hm_internal.h"

namespace hm {

namespace {

struct Part {
  int dev = 0;
  uint64_t local = 0;          // the one-device handle on `dev`
  size_t lo = 0, hi = 0;       // the index range of the set this part holds (replicated: the whole set)
  std::mutex mu;               // guards `stage`
  DevBuf stage;                // scalars that arrive as device pointers of ANOTHER device land here (hipMemcpyPeer)
};

struct MultiBases {
  uint64_t handle = 0;
  size_t n = 0;
  bool sliced = false;
  std::vector<std::unique_ptr<Part>> parts;
};

std::mutex g_multi_mu;
std::map<uint64_t, std::shared_ptr<MultiBases>> g_multi;
uint64_t g_multi_next = 1;

std::shared_ptr<MultiBases> find_multi(uint64_t handle) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  auto it = g_multi.find(handle);
  return it == g_multi.end() ? nullptr : it->second;
}

struct DeviceRestore {         // a part run on the calling thread must leave the thread's device as it found it
  int prev = -1;
  DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int device_of_pointer(const void* p, int* dev) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return hm_fail(HM_ERR_BAD_ARG, "multi-device call: the scalar / base pointer is not a device pointer");
  }
  *dev = attr.device;
  return HM_OK;
}

// the bytes [src, src + bytes) of device `src_dev` as a pointer valid on the part's device: the pointer itself when
// that is the same device, else a copy in the part's staging buffer at `stage_off` (part.mu held by the caller)
int bring(Part& part, const void* src, int src_dev, size_t bytes, size_t stage_off, size_t stage_total, const void** out) {
  // HALO2_MI355X_FORCE_PEER_STAGE=1 (tests on a one-GPU box): stage even when source and part share the device
  const char* fs = std::getenv("HALO2_MI355X_FORCE_PEER_STAGE");
  const bool force_stage = fs && *fs == '1';
  if (src_dev == part.dev && !force_stage) {
    *out = src;
    return HM_OK;
  }
  uint8_t* buf = (uint8_t*)part.stage.ensure(stage_total ? stage_total : 32);
  if (!buf) return hm_fail(HM_ERR_HIP, "multi-device call: staging allocation failed");
  HM_HIP_CHECK(hipMemcpyPeer(buf + stage_off, part.dev, src, src_dev, bytes));
  *out = buf + stage_off;
  return HM_OK;
}

}  // namespace

bool& multi_worker_flag() {
  static thread_local bool flag = false;
  return flag;
}

int run_per_device(const std::vector<int>& devs, const std::function<int(size_t)>& fn) {
  const size_t parts = devs.size();
  std::vector<int> rcs(parts, HM_OK);
  std::vector<std::string> errs(parts);
  auto body = [&](size_t r) {
    try {
      if (hipSetDevice(devs[r]) != hipSuccess) {
        rcs[r] = HM_ERR_HIP;
        errs[r] = "multi-device call: hipSetDevice failed for device " + std::to_string(devs[r]);
        return;
      }
      hm_fault_point("worker_body");
      struct InWorker {           // the one-device entry points a part calls must not dispatch on the device list again
        bool prev = multi_worker_flag();
        InWorker() { multi_worker_flag() = true; }
        ~InWorker() { multi_worker_flag() = prev; }
      } in_worker;
      rcs[r] = fn(r);
      if (rcs[r] != HM_OK) errs[r] = hm_last_error_string();   // thread-local: carry it back to the caller's thread
    } catch (const std::exception& e) {
      rcs[r] = HM_ERR_INTERNAL;
      try { errs[r] = std::string("multi-device worker: ") + e.what(); } catch (...) {}
    } catch (...) {
      rcs[r] = HM_ERR_INTERNAL;
    }
  };
  {
    JoinOnExit pool;
    std::vector<char> started(parts, 0);
    for (size_t r = 1; r < parts; ++r) started[r] = spawn_or_false(pool, "worker_spawn", [&body, r] { body(r); }) ? 1 : 0;
    DeviceRestore keep;
    if (parts) body(0);
    for (size_t r = 1; r < parts; ++r)
      if (!started[r]) body(r);                  // no thread to be had: the calling thread takes the part
  }                                              // every worker joined here
  for (size_t r = 0; r < parts; ++r)
    if (rcs[r] != HM_OK) return hm_fail(rcs[r], errs[r].empty() ? std::string("multi-device worker failed") : errs[r]);
  return HM_OK;
}

// layout: 0 the library's default, 1 the fixed-base table, 2 plain (capi.hip: BaseLayout) -- every part through the public
// one-device entry of that layout (on a worker thread those stay on one device: multi_worker_flag)
int multi_register(const uint64_t* bases_host, const void* d_bases, size_t n, void* stream, int layout, const std::vector<int>& devs,
                   uint64_t* out_handle) {
  auto reg_host = [layout](const uint64_t* b, size_t cnt, uint64_t* out) {
    return layout == 1 ? hm_register_bases_precomp(b, cnt, out) : layout == 2 ? hm_register_bases_plain(b, cnt, out) : hm_register_bases(b, cnt, out);
  };
  auto reg_dev = [layout](const void* b, size_t cnt, uint64_t* out) {
    return layout == 1 ? hm_register_bases_precomp_dev(b, cnt, nullptr, out)
                       : layout == 2 ? hm_register_bases_plain_dev(b, cnt, nullptr, out) : hm_register_bases_dev(b, cnt, nullptr, out);
  };
  // HALO2_MI355X_SLICE_FROM_LOG: the size from which sets are sliced instead of replicated (default 22; tests lower it)
  const char* sf = std::getenv("HALO2_MI355X_SLICE_FROM_LOG");
  const size_t slice_from = sf && *sf ? (size_t)1 << std::atoi(sf) : kSliceBasesFrom;
  auto mb = std::make_shared<MultiBases>();
  mb->n = n;
  mb->sliced = n >= slice_from;
  const size_t P = devs.size();
  for (size_t r = 0; r < P; ++r) {
    auto p = std::make_unique<Part>();
    p->dev = devs[r];
    p->lo = mb->sliced ? n * r / P : 0;
    p->hi = mb->sliced ? n * (r + 1) / P : n;
    mb->parts.push_back(std::move(p));
  }
  int src_dev = -1;
  if (d_bases) {
    int rc = device_of_pointer(d_bases, &src_dev);
    if (rc != HM_OK) return rc;
    DeviceRestore keep;
    HM_HIP_CHECK(hipSetDevice(src_dev));
    HM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));     // the bases are produced on `stream` of their own device
  }
  const int rc = run_per_device(devs, [&](size_t r) -> int {
    Part& part = *mb->parts[r];
    const size_t cnt = part.hi - part.lo;
    if (bases_host || n == 0) {
      const uint64_t* b = bases_host ? bases_host + part.lo * 8 : nullptr;
      return reg_host(b, cnt, &part.local);
    }
    const uint8_t* src = (const uint8_t*)d_bases + part.lo * 64;
    void* tmp = nullptr;
    if (src_dev != part.dev) {                    // once per SRS: the raw points cross xGMI, are converted, and the copy is dropped
      HM_HIP_CHECK(hipMalloc(&tmp, cnt ? cnt * 64 : 64));
      const hipError_t e = hipMemcpyPeer(tmp, part.dev, src, src_dev, cnt * 64);
      if (e != hipSuccess) {
        (void)hipFree(tmp);
        return hm_fail(HM_ERR_HIP, std::string("hm_register_bases_dev: hipMemcpyPeer: ") + hipGetErrorString(e));
      }
      src = (const uint8_t*)tmp;
    }
    const int rr = reg_dev(src, cnt, &part.local);
    if (tmp) (void)hipFree(tmp);
    return rr;
  });
  if (rc != HM_OK) {                              // drop what did get registered
    const std::string keep_msg = hm_last_error_string();
    (void)run_per_device(devs, [&](size_t r) -> int {
      if (mb->parts[r]->local) (void)hm_release_bases(mb->parts[r]->local);
      return HM_OK;
    });
    return hm_fail(rc, keep_msg);
  }
  std::lock_guard<std::mutex> lk(g_multi_mu);
  mb->handle = HM_MULTI_HANDLE_BIT | g_multi_next++;
  g_multi[mb->handle] = mb;
  *out_handle = mb->handle;
  return HM_OK;
}

int multi_bases_info(uint64_t handle, hm_bases_info* out) {
  auto mb = find_multi(handle);
  if (!mb) return hm_fail(HM_ERR_NOT_FOUND, "hm_get_bases_info: unknown handle");
  std::vector<int> devs;
  for (auto& p : mb->parts) devs.push_back(p->dev);
  std::vector<hm_bases_info> infos(devs.size());
  const int rc = run_per_device(devs, [&](size_t r) -> int { return hm_get_bases_info(mb->parts[r]->local, &infos[r]); });
  if (rc != HM_OK) return rc;
  out->n = mb->n;
  out->devices = (uint32_t)devs.size();
  out->sliced = mb->sliced ? 1u : 0u;
  out->table_windows = infos.empty() ? 0 : infos[0].table_windows;        // 0 as soon as one part has no table
  out->table_window_bits = infos.empty() ? 0 : infos[0].table_window_bits;
  for (const auto& i : infos) {
    out->device_bytes += i.device_bytes;
    out->parked_bytes += i.parked_bytes;
    if (i.table_windows == 0) out->table_windows = out->table_window_bits = 0;
  }
  return HM_OK;
}

static int release_parts(MultiBases& mb) {
  std::vector<int> devs;
  for (auto& p : mb.parts) devs.push_back(p->dev);
  return run_per_device(devs, [&](size_t r) -> int {
    Part& part = *mb.parts[r];
    {
      std::lock_guard<std::mutex> lk(part.mu);
      part.stage.release();
    }
    return hm_release_bases(part.local);
  });
}

int multi_release(uint64_t handle) {
  std::shared_ptr<MultiBases> mb;
  {
    std::lock_guard<std::mutex> lk(g_multi_mu);
    auto it = g_multi.find(handle);
    if (it == g_multi.end()) return hm_fail(HM_ERR_NOT_FOUND, "hm_release_bases: unknown handle");
    mb = it->second;
    g_multi.erase(it);
  }
  return release_parts(*mb);
}

void multi_release_touching(int device) {
  std::vector<std::shared_ptr<MultiBases>> gone;
  {
    std::lock_guard<std::mutex> lk(g_multi_mu);
    for (auto it = g_multi.begin(); it != g_multi.end();) {
      bool touches = false;
      for (auto& p : it->second->parts) touches = touches || p->dev == device;
      if (touches) {
        gone.push_back(it->second);
        it = g_multi.erase(it);
      } else {
        ++it;
      }
    }
  }
  for (auto& mb : gone) (void)release_parts(*mb);
}

int multi_local_part(uint64_t handle, int device, uint64_t* local_handle) {
  auto mb = find_multi(handle);
  if (!mb) return hm_fail(HM_ERR_NOT_FOUND, "unknown base handle");
  if (mb->sliced)
    return hm_fail(HM_ERR_BAD_ARG, "hm_msm_submit_dev: the set is sliced over several devices; use hm_msm_bn256_g1_dev or the batch forms");
  for (auto& p : mb->parts)
    if (p->dev == device) {
      *local_handle = p->local;
      return HM_OK;
    }
  return hm_fail(HM_ERR_BAD_ARG, "hm_msm_submit_dev: the calling thread's device holds no copy of this base set");
}

// the index ranges [start, end) of one MSM over [offset, offset + n), per part (empty ranges: start == end)
static void split_ranges(const MultiBases& mb, size_t offset, size_t n, std::vector<size_t>& start, std::vector<size_t>& end) {
  const size_t P = mb.parts.size();
  start.assign(P, 0);
  end.assign(P, 0);
  if (mb.sliced) {
    for (size_t r = 0; r < P; ++r) {
      const size_t lo = std::max(mb.parts[r]->lo, offset), hi = std::min(mb.parts[r]->hi, offset + n);
      if (lo < hi) { start[r] = lo; end[r] = hi; }
    }
  } else if (n < P * kMinShardPoints) {          // too small to be worth splitting: the first listed device takes it whole
    start[0] = offset;
    end[0] = offset + n;
  } else {
    for (size_t r = 0; r < P; ++r) {
      start[r] = offset + n * r / P;
      end[r] = offset + n * (r + 1) / P;
    }
  }
}

int multi_msm(uint64_t handle, size_t offset, const void* scalars, bool from_host, size_t n, void* stream, uint64_t jac[12], int* is_id) {
  auto mb = find_multi(handle);
  if (!mb) return hm_fail(HM_ERR_NOT_FOUND, "unknown base handle");
  if (offset > mb->n || n > mb->n - offset) return hm_fail(HM_ERR_BAD_ARG, "offset + n exceeds the base set");
  int src_dev = -1;
  if (!from_host && n) {
    int rc = device_of_pointer(scalars, &src_dev);
    if (rc != HM_OK) return rc;
    DeviceRestore keep;
    HM_HIP_CHECK(hipSetDevice(src_dev));
    HM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  }
  std::vector<size_t> start, end;
  split_ranges(*mb, offset, n, start, end);
  const size_t P = mb->parts.size();
  std::vector<uint64_t> partial(P * 12, 0);
  std::vector<int> devs;
  for (auto& p : mb->parts) devs.push_back(p->dev);
  const int rc = run_per_device(devs, [&](size_t r) -> int {
    const size_t cnt = end[r] - start[r];
    if (cnt == 0) return HM_OK;                   // the partial stays the identity (z = 0)
    Part& part = *mb->parts[r];
    const size_t loff = start[r] - part.lo;
    int id = 0;
    if (from_host) return msm_h_local(part.local, loff, (const uint64_t*)scalars + (start[r] - offset) * 4, cnt, &partial[r * 12], &id);
    std::lock_guard<std::mutex> lk(part.mu);
    const void* src = nullptr;
    int brc = bring(part, (const uint8_t*)scalars + (start[r] - offset) * 32, src_dev, cnt * 32, 0, cnt * 32, &src);
    if (brc != HM_OK) return brc;
    return hm_msm_bn256_g1_dev(part.local, loff, src, cnt, nullptr, &partial[r * 12]);
  });
  if (rc != HM_OK) return rc;
  host_sum_points(partial.data(), P, jac, is_id);
  return HM_OK;
}

int multi_msm_batch(uint64_t handle, size_t offset, const void* const* scalars, bool from_host, size_t n, size_t count, void* stream,
                    uint64_t* out_xyz) {
  auto mb = find_multi(handle);
  if (!mb) return hm_fail(HM_ERR_NOT_FOUND, "unknown base handle");
  if (offset > mb->n || n > mb->n - offset) return hm_fail(HM_ERR_BAD_ARG, "offset + n exceeds the base set");
  if (count == 0) return HM_OK;
  for (size_t i = 0; i < count; ++i)
    if (!scalars[i] && n) return hm_fail(HM_ERR_BAD_ARG, "null scalar array");
  int src_dev = -1;
  if (!from_host && n) {
    int rc = device_of_pointer(scalars[0], &src_dev);
    if (rc != HM_OK) return rc;
    DeviceRestore keep;
    HM_HIP_CHECK(hipSetDevice(src_dev));
    HM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  }
  const size_t P = mb->parts.size();
  std::vector<int> devs;
  for (auto& p : mb->parts) devs.push_back(p->dev);
  if (!mb->sliced) {
    // whole commitments, round-robin: part r computes the commitments r, r + P, r + 2P, ...
    std::vector<std::vector<uint64_t>> res(P);
    const int rc = run_per_device(devs, [&](size_t r) -> int {
      Part& part = *mb->parts[r];
      std::vector<const void*> mine;
      for (size_t i = r; i < count; i += P) mine.push_back(scalars[i]);
      if (mine.empty()) return HM_OK;
      res[r].assign(mine.size() * 12, 0);
      if (from_host)
        return hm_msm_batch_bn256_g1_h(part.local, offset, reinterpret_cast<const uint64_t* const*>(mine.data()), n, mine.size(), res[r].data());
      std::lock_guard<std::mutex> lk(part.mu);
      for (size_t j = 0; j < mine.size(); ++j) {
        int brc = bring(part, mine[j], src_dev, n * 32, j * n * 32, mine.size() * n * 32, &mine[j]);
        if (brc != HM_OK) return brc;
      }
      return hm_msm_batch_bn256_g1_dev(part.local, offset, mine.data(), n, mine.size(), nullptr, res[r].data());
    });
    if (rc != HM_OK) return rc;
    for (size_t i = 0; i < count; ++i) std::memcpy(out_xyz + 12 * i, &res[i % P][12 * (i / P)], 96);
    return HM_OK;
  }
  // sliced set: every device computes its index range of EVERY commitment; the partials are folded per commitment
  std::vector<size_t> start, end;
  split_ranges(*mb, offset, n, start, end);
  std::vector<std::vector<uint64_t>> res(P);
  const int rc = run_per_device(devs, [&](size_t r) -> int {
    const size_t cnt = end[r] - start[r];
    res[r].assign(count * 12, 0);                 // an empty range leaves identities
    if (cnt == 0) return HM_OK;
    Part& part = *mb->parts[r];
    const size_t loff = start[r] - part.lo, skip = (start[r] - offset) * 32;
    std::vector<const void*> mine(count);
    for (size_t i = 0; i < count; ++i) mine[i] = (const uint8_t*)scalars[i] + skip;
    if (from_host)
      return hm_msm_batch_bn256_g1_h(part.local, loff, reinterpret_cast<const uint64_t* const*>(mine.data()), cnt, count, res[r].data());
    std::lock_guard<std::mutex> lk(part.mu);
    for (size_t i = 0; i < count; ++i) {
      int brc = bring(part, mine[i], src_dev, cnt * 32, i * cnt * 32, count * cnt * 32, &mine[i]);
      if (brc != HM_OK) return brc;
    }
    return hm_msm_batch_bn256_g1_dev(part.local, loff, mine.data(), cnt, count, nullptr, res[r].data());
  });
  if (rc != HM_OK) return rc;
  std::vector<uint64_t> pts(P * 12);
  for (size_t i = 0; i < count; ++i) {
    for (size_t r = 0; r < P; ++r) std::memcpy(&pts[r * 12], &res[r][12 * i], 96);
    int id = 0;
    host_sum_points(pts.data(), P, out_xyz + 12 * i, &id);
  }
  return HM_OK;
}

}  // namespace hm
