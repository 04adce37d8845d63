// msm.hip -- Pippenger bucket MSM over BN256 G1 for gfx950 (replaces
// halo2_proofs::arithmetic::best_multiexp, SURVEY.md §3.3: sum_i coeffs[i] * bases[i]).
//
// Pipeline (all device-resident, nothing is read back before the final window sums):
//   K0  digits      scalar (radix-2^256 Montgomery) -> canonical -> W signed c-bit digits
//   K2  sort        two-level counting sort through LDS (coarse partition, then one workgroup per
//                   region: fine histogram -> bucket counts; global scan -> bucket + task offsets;
//                   in-region scatter) -> point indices grouped by bucket
//   K2b task order  counting sort of the tasks by chain length, longest first: the 64 lanes of a
//                   K3 wave run chains of equal length
//   K3  accumulate  one lane per task (a bucket, or a <= L-long slice of a long bucket): a serial
//                   chain of extended-Jacobian mixed additions, the accumulator living in registers
//   K3b finalize    per bucket: sum of its task partials (lane / workgroup / sliced workgroups)
//   K4a/K4b reduce  per window sum_b b*B_b by segmented running sums, then a tree in LDS
//   host            Horner over the W window sums (c doublings each) and affine normalisation
// Differences from the reference's CPU algorithm that do not change the (canonical) result:
// signed digits (half the buckets), a window size chosen for the whole array instead of per
// thread chunk, buckets summed in sorted order.  Long buckets (constant columns, tiny scalars) are
// split into tasks so that one hot bucket cannot serialise the launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "hm_internal.h"
#include "host_fq.h"
#include "msm_dev.h"

namespace hm {


// Every counter an MSM starts from zero is cleared by ONE launch of its own (instead of four
// hipMemsetAsync calls): the bucket counts (the cooperative histogram adds into them), the big-region
// list count, the task-length key histogram, the hot-bucket queue counts.
__global__ void msm_init_counters_kernel(uint32_t* __restrict__ bcnt, size_t nbt, uint32_t* __restrict__ br_count,
                                         uint32_t* __restrict__ khist, uint32_t nkeys, uint32_t* __restrict__ big_count) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = i0; i < nbt; i += stride) bcnt[i] = 0;
  if (i0 < nkeys) khist[i0] = 0;
  if (i0 < 4) {
    if (br_count) br_count[i0] = 0;
    big_count[i0] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// bases: external affine -> internal packed (x, y) + identity flags
// ---------------------------------------------------------------------------------------------
__global__ void msm_convert_bases_kernel(const uint32_t* __restrict__ ext, uint32_t* __restrict__ xy,
                                         uint8_t* __restrict__ inf, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(ext + i * 16);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  const uint32_t wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const uint32_t wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) any |= wx[k] | wy[k];
  uint32_t ox[8], oy[8];
  fe_pack(ox, fe_from_ext<FqParams>(wx));
  fe_pack(oy, fe_from_ext<FqParams>(wy));
  uint4* o = reinterpret_cast<uint4*>(xy + i * 16);
  o[0] = make_uint4(ox[0], ox[1], ox[2], ox[3]);
  o[1] = make_uint4(ox[4], ox[5], ox[6], ox[7]);
  o[2] = make_uint4(oy[0], oy[1], oy[2], oy[3]);
  o[3] = make_uint4(oy[4], oy[5], oy[6], oy[7]);
  inf[i] = any == 0 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// K0: signed window digits.  digits[w*n + i] in [-2^(c-1)+1, 2^(c-1)], 0 = skip.
// ---------------------------------------------------------------------------------------------
// Tasks are cut to at most L pairs.  The host picks L for the worst case (every digit non-zero); the
// first sort level counts the pairs that really exist, and when they are far fewer (zero or repeated
// digits: flag and selector-like columns) every later kernel derives the same shorter L from that
// count, so that K3 still gets >= TARGET_TASKS chains to fill the chip with.
#ifndef HM_TARGET_TASKS      // measurement knob (tools/ab_build.sh)
#define HM_TARGET_TASKS 327680
#endif
constexpr uint32_t TARGET_TASKS = HM_TARGET_TASKS;   // 256 CUs x 4 SIMDs x 5 waves x 64 lanes
__device__ __forceinline__ uint32_t effective_task_len(const uint32_t* __restrict__ pairs, uint32_t L_host) {
  if (pairs == nullptr) return L_host;
  const uint32_t by_fill = pairs[0] / TARGET_TASKS;
  const uint32_t L = L_host < by_fill ? L_host : by_fill;
  return L < 16u ? 16u : L;
}

// Windows [0, n_wide) are c bits wide, the others c - 1 (n_wide = W: every window c bits, the top one implicitly shorter).
// A GROUP of MSMs over the same points (blockIdx.y = the element): element e reads scalars list.s[e] and writes its own
// W x n digit array behind the previous element's (pointer / dword tables of a by-value argument are read with scalar
// loads: graph.hip's note on what must not travel by value does not apply).
struct MsmGroupScalars {
  const uint32_t* s[HM_MSM_GROUP];
};
__global__ void msm_digits_kernel(MsmGroupScalars list, const uint8_t* __restrict__ inf,
                                  int32_t* __restrict__ digits, size_t n, uint32_t c, uint32_t W, uint32_t n_wide) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* __restrict__ scalars = list.s[blockIdx.y];
  digits += (size_t)blockIdx.y * W * n;
  const uint4* q = reinterpret_cast<const uint4*>(scalars + i * 8);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w_in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  // s_mont * 32 * 2^-261 = s_mont * 2^-256 = the canonical scalar (Fr::to_repr)
  Fr k32 = fe_zero<FrParams>();
  k32.l[0] = 32;
  const Fr s = fe_canonical(fe_mul(fe_unpack<FrParams>(w_in), k32));
  uint32_t v[9];
  {
    uint32_t t[8];
    fe_pack(t, s);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = t[k];
    v[8] = 0;
  }
  const bool skip = inf[i] != 0;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t bits = w < n_wide ? c : c - 1;
    const uint32_t mask = (1u << bits) - 1u, half = 1u << (bits - 1);
    uint32_t d = (v[0] & mask) + carry;
    int32_t sd;
    if (d > half) {
      sd = (int32_t)d - (int32_t)(1u << bits);
      carry = 1;
    } else {
      sd = (int32_t)d;
      carry = 0;
    }
    // a narrow window's digit is DOUBLED and its table entry holds half the weight (2^(offset - 1) P): its points then
    // spread over the even buckets of the whole range instead of filling the lower half only (every coarse region of
    // the sort gets the same share: the lower half would hold seven times the upper one at c = 22, W = 12)
    digits[(size_t)w * n + i] = skip ? 0 : (w < n_wide ? sd : 2 * sd);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __funnelshift_r(v[k], v[k + 1], bits);
  }
}

// ---------------------------------------------------------------------------------------------
// K2: grouping (bucket, point index) pairs by bucket -- a two-level counting sort through LDS.
//
// A one-level sort (every chunk scattering straight into 2^(c-1) bucket runs) writes each 32-byte
// HBM sector from ~8 different workgroups at different times: 8x write amplification, measured
// (profiles/r01_pmc_traffic.json: 8.2 GiB written for a 1 GiB index array).  Instead:
//   part 1  partition by the HIGH `cb` bits of the bucket: each (chunk, window) workgroup appends
//           its items to 2^cb coarse bins in ~1 KiB runs; an item is ONE 32-bit word
//           [fine bucket bits | sign | point index].
//   part 2  one workgroup per (coarse bin, window) owns a contiguous region of the final array:
//           fine histogram in LDS -> bucket counts; after the global scan it scatters the items of
//           its own region, so every sector is completed by one workgroup while still in L2.
// When fine bits + sign + index bits fit 32 bits with cb = 0 (n <= 2^19) part 1 disappears and
// part 2 reads the digits directly.
// ---------------------------------------------------------------------------------------------
#ifndef HM_SORT_THREADS
#define HM_SORT_THREADS 1024
#endif
constexpr int SORT_THREADS = HM_SORT_THREADS;

// Apply f(index, word) to base[lo .. hi) with the whole workgroup, 16 bytes per lane per load:
// one 4-byte load in flight per lane keeps only ~1 MB outstanding chip-wide (latency-bound at
// ~0.5 TB/s); four words per load quadruple the bytes in flight.
template <class T, class F>
__device__ __forceinline__ void block_for_each_word(const T* __restrict__ base, size_t lo, size_t hi, F f) {
  size_t a = lo;
  while (a < hi && (reinterpret_cast<uintptr_t>(base + a) & (4 * sizeof(T) - 1))) ++a;   // aligned start
  const size_t nvec = (hi - a) / 4;
  if (threadIdx.x < a - lo) f(lo + threadIdx.x, base[lo + threadIdx.x]);     // <= 3 head words
  const size_t tail = a + nvec * 4;
  if (threadIdx.x < hi - tail) f(tail + threadIdx.x, base[tail + threadIdx.x]);   // <= 3 tail words
  typedef T vec4 __attribute__((ext_vector_type(4)));
  const vec4* vb = reinterpret_cast<const vec4*>(base + a);
  for (size_t v = threadIdx.x; v < nvec; v += SORT_THREADS) {
    const vec4 q = vb[v];
    const size_t i = a + v * 4;
    f(i, q.x);
    f(i + 1, q.y);
    f(i + 2, q.z);
    f(i + 3, q.w);
  }
}

__global__ __launch_bounds__(SORT_THREADS) void msm_part1_hist_kernel(const int32_t* __restrict__ digits,
                                                                      uint32_t* __restrict__ chist, size_t n, size_t chunk,
                                                                      uint32_t fb, uint32_t NC) {
  extern __shared__ uint32_t hist[];
  const uint32_t g = blockIdx.x, w = blockIdx.y, G = gridDim.x;
  for (uint32_t b = threadIdx.x; b < NC; b += SORT_THREADS) hist[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int32_t* dw = digits + (size_t)w * n;
  block_for_each_word(dw, lo, hi, [&](size_t, int32_t d) {
    if (d != 0) (void)lds_inc(hist, ((uint32_t)(d < 0 ? -d : d) - 1u) >> fb);
  });
  __syncthreads();
  uint32_t* out = chist + ((size_t)w * G + g) * NC;
  for (uint32_t b = threadIdx.x; b < NC; b += SORT_THREADS) out[b] = hist[b];
}

// per (window, coarse bin): exclusive prefix over chunks in place, total into ctot.
// One 64-lane workgroup per (window, bin): lanes take contiguous runs of chunks, then a wave scan.
__global__ __launch_bounds__(64) void msm_part1_scan_kernel(uint32_t* __restrict__ chist, uint32_t* __restrict__ ctot, uint32_t G,
                                                            uint32_t NC, uint32_t W) {
  const uint32_t idx = blockIdx.x;
  if (idx >= W * NC) return;
  const uint32_t w = idx / NC, b = idx - w * NC;
  const uint32_t lane = threadIdx.x;
  const uint32_t per = (G + 63) / 64;
  const uint32_t lo = lane * per, hi = lo + per < G ? lo + per : G;
  uint32_t sum = 0;
  for (uint32_t g = lo; g < hi; ++g) sum += chist[((size_t)w * G + g) * NC + b];
  uint32_t incl = sum;                       // inclusive scan across the 64 lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off, 64);
    if (lane >= (uint32_t)off) incl += v;
  }
  uint32_t run = incl - sum;
  for (uint32_t g = lo; g < hi; ++g) {
    uint32_t* p = chist + ((size_t)w * G + g) * NC + b;
    const uint32_t t = *p;
    *p = run;
    run += t;
  }
  if (lane == 63) ctot[idx] = incl;
}

// one block: cstart = exclusive scan of ctot (count <= 16 * 1024 + 1 entries; cstart[count] = total)
__global__ __launch_bounds__(1024) void msm_part1_starts_kernel(const uint32_t* __restrict__ ctot,
                                                                uint32_t* __restrict__ cstart, uint32_t count) {
  __shared__ uint32_t s[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (count + 1023) / 1024;
  const uint32_t lo = t * per, hi = lo + per < count ? lo + per : count;
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += ctot[i];
  s[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    uint32_t a = 0;
    if (t >= off) a = s[t - off];
    __syncthreads();
    s[t] += a;
    __syncthreads();
  }
  uint32_t run = s[t] - sum;
  for (uint32_t i = lo; i < hi; ++i) {
    cstart[i] = run;
    run += ctot[i];
  }
  if (t == 1023) cstart[count] = s[1023];
}

// ITEM = uint32_t when [fine bits | sign | index] fits 32 bits, else uint64_t
template <class ITEM>
__global__ __launch_bounds__(SORT_THREADS) void msm_part1_scatter_kernel(const int32_t* __restrict__ digits,
                                                                         const uint32_t* __restrict__ chist,
                                                                         const uint32_t* __restrict__ cstart,
                                                                         ITEM* __restrict__ tmp, size_t n, size_t chunk,
                                                                         uint32_t fb, uint32_t ib, uint32_t NC) {
  extern __shared__ uint32_t cursor[];
  const uint32_t g = blockIdx.x, w = blockIdx.y, G = gridDim.x;
  const uint32_t* pre = chist + ((size_t)w * G + g) * NC;
  const uint32_t* cs = cstart + (size_t)w * NC;
  for (uint32_t b = threadIdx.x; b < NC; b += SORT_THREADS) cursor[b] = cs[b] + pre[b];
  __syncthreads();
  const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int32_t* dw = digits + (size_t)w * n;
  const uint32_t fmask = (1u << fb) - 1u;
  const ITEM imask = ((ITEM)1 << ib) - 1;     // positional items keep only the low ib bits of the index (see msm_issue)
  block_for_each_word(dw, lo, hi, [&](size_t i, int32_t d) {
    if (d != 0) {
      const uint32_t b1 = (uint32_t)(d < 0 ? -d : d) - 1u;
      const uint32_t pos = lds_inc(cursor, b1 >> fb);
      tmp[pos] = ((ITEM)(b1 & fmask) << (ib + 1)) | ((ITEM)(d < 0 ? 1u : 0u) << ib) | ((ITEM)i & imask);
    }
  });
}

// part 1 scatter, tiled form: the chunk is processed in tiles of 4096 digits that are first
// counting-sorted by coarse bin inside LDS, so that consecutive lanes append to the same coarse run
// (sector-complete stores, as in the tiled part-2 scatter below).  Used while NC <= 4096.
#ifndef HM_P1_IPT
#define HM_P1_IPT 4
#endif
constexpr int P1_IPT_DEFAULT = HM_P1_IPT;
#ifndef HM_P1_BIG_IPT        // tiles for 512 .. 2048 coarse bins (the shared-bucket-set plan, the plain layout from 2^21 points).
#define HM_P1_BIG_IPT 16     // Measured at 2^24 on the table, whole sort: 8 items per lane 3.19 ms, 16 (with part 2's 16) 3.00 ms --
#endif                       // 16 items = a 64-byte run per coarse bin and tile instead of 32 (140 KiB of LDS, one workgroup per CU)
constexpr int P1_BIG_IPT = HM_P1_BIG_IPT;
// P1_IPT items per lane and tile: a tile should bring >= 8 items (one 32-byte sector) to every coarse bin, i.e.
// 4 096 items for <= 512 bins and 8 192 for 1 024 (the shared-bucket-set plan)
template <class ITEM, int P1_IPT = P1_IPT_DEFAULT>
__global__ __launch_bounds__(SORT_THREADS) void msm_part1_scatter_tiled_kernel(const int32_t* __restrict__ digits,
                                                                               const uint32_t* __restrict__ chist,
                                                                               const uint32_t* __restrict__ cstart,
                                                                               ITEM* __restrict__ tmp, size_t n, size_t chunk,
                                                                               uint32_t fb, uint32_t ib, uint32_t NC) {
  extern __shared__ uint32_t sm[];
  constexpr int P1_TILE = SORT_THREADS * P1_IPT;
  const uint32_t g = blockIdx.x, w = blockIdx.y, G = gridDim.x, tid = threadIdx.x;
  uint32_t* gcur = sm;                 // global cursor of every coarse bin for this chunk
  uint32_t* tcnt = gcur + NC;
  uint32_t* tstart = tcnt + NC;
  uint32_t* wsum = tstart + NC;
  uint32_t* st_bin = wsum + 32;
  ITEM* st_item = reinterpret_cast<ITEM*>(st_bin + P1_TILE);
  const uint32_t* pre = chist + ((size_t)w * G + g) * NC;
  const uint32_t* cs = cstart + (size_t)w * NC;
  for (uint32_t b = tid; b < NC; b += SORT_THREADS) { gcur[b] = cs[b] + pre[b]; tcnt[b] = 0; }
  __syncthreads();
  const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int32_t* dw = digits + (size_t)w * n;
  const uint32_t fmask = (1u << fb) - 1u;
  const ITEM imask = ((ITEM)1 << ib) - 1;
  const uint32_t per = (NC + SORT_THREADS - 1) / SORT_THREADS;
  for (size_t t0 = lo; t0 < hi; t0 += P1_TILE) {
    const uint32_t tile_n = hi - t0 < (size_t)P1_TILE ? (uint32_t)(hi - t0) : (uint32_t)P1_TILE;
    ITEM item[P1_IPT];
    uint32_t bin[P1_IPT], rank[P1_IPT];
#pragma unroll
    for (int k = 0; k < P1_IPT; ++k) {
      const uint32_t e = (uint32_t)k * SORT_THREADS + tid;
      bin[k] = 0xffffffffu;
      if (e < tile_n) {
        const int32_t d = dw[t0 + e];
        if (d != 0) {
          const uint32_t b1 = (uint32_t)(d < 0 ? -d : d) - 1u;
          bin[k] = b1 >> fb;
          item[k] = ((ITEM)(b1 & fmask) << (ib + 1)) | ((ITEM)(d < 0 ? 1u : 0u) << ib) | ((ITEM)(t0 + e) & imask);
          rank[k] = lds_inc(tcnt, bin[k]);
        }
      }
    }
    __syncthreads();
    {   // exclusive scan of tcnt[0 .. NC)
      const uint32_t b0 = tid * per;
      uint32_t v[4096 / SORT_THREADS] = {}, sum = 0;
      for (uint32_t k = 0; k < per; ++k)
        if (b0 + k < NC) { v[k] = tcnt[b0 + k]; sum += v[k]; }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if ((tid & 63) >= (uint32_t)off) incl += u;
      }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      if (tid < 64) {
        const uint32_t ws = tid < (SORT_THREADS / 64) ? wsum[tid] : 0;
        uint32_t wi = ws;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t u = __shfl_up(wi, off, 64);
          if (tid >= (uint32_t)off) wi += u;
        }
        if (tid < (SORT_THREADS / 64)) wsum[tid] = wi - ws;
      }
      __syncthreads();
      uint32_t run = wsum[tid >> 6] + incl - sum;
      for (uint32_t k = 0; k < per; ++k)
        if (b0 + k < NC) { tstart[b0 + k] = run; run += v[k]; }
    }
    __syncthreads();
    uint32_t kept = 0;
#pragma unroll
    for (int k = 0; k < P1_IPT; ++k) {
      if (bin[k] != 0xffffffffu) {
        const uint32_t pos = tstart[bin[k]] + rank[k];
        st_item[pos] = item[k];
        st_bin[pos] = bin[k];
      }
    }
    __syncthreads();
    kept = tstart[NC - 1] + tcnt[NC - 1];      // non-zero digits of this tile
    for (uint32_t e = tid; e < kept; e += SORT_THREADS) {
      const uint32_t b = st_bin[e];
      tmp[gcur[b] + (e - tstart[b])] = st_item[e];
    }
    __syncthreads();
    for (uint32_t b = tid; b < NC; b += SORT_THREADS) { gcur[b] += tcnt[b]; tcnt[b] = 0; }
    __syncthreads();
  }
}

// part 2, first half: fine histogram of one (coarse bin, window) region -> bucket counts.
// FROM_DIGITS (cb = 0): the "region" is the whole window and items are read from the digit array.
// Regions larger than `big` items (a hot bucket: constant or flag columns send a whole window to one
// region) are left to the COOP instantiation, which many workgroups run on slices of such a region
// (work list built by msm_big_regions_kernel) and which ADDS its counts to bcnt with global atomics.
template <bool FROM_DIGITS, class ITEM, bool COOP = false>
__global__ __launch_bounds__(SORT_THREADS) void msm_part2_hist_kernel(const ITEM* __restrict__ tmp,
                                                                      const int32_t* __restrict__ digits,
                                                                      const uint32_t* __restrict__ cstart,
                                                                      uint32_t* __restrict__ bcnt, size_t n, uint32_t fb,
                                                                      uint32_t ib, uint32_t NC, uint32_t NBP, uint32_t big,
                                                                      uint32_t slice, const uint2* __restrict__ list,
                                                                      const uint32_t* __restrict__ list_count) {
  extern __shared__ uint32_t fine[];
  uint32_t hb = blockIdx.x, w = blockIdx.y, sl = 0;
  if (COOP) {
    if (blockIdx.x >= *list_count) return;
    const uint2 it = list[blockIdx.x];
    w = it.x / NC;
    hb = it.x - w * NC;
    sl = it.y;
  }
  const uint32_t NF = 1u << fb;
  if (!FROM_DIGITS && !COOP && cstart[w * NC + hb + 1] - cstart[w * NC + hb] > big) return;   // bcnt was zeroed
  for (uint32_t b = threadIdx.x; b < NF; b += SORT_THREADS) fine[b] = 0;
  __syncthreads();
  if (FROM_DIGITS) {
    const int32_t* dw = digits + (size_t)w * n;
    block_for_each_word(dw, 0, n, [&](size_t, int32_t d) {
      if (d != 0) (void)lds_inc(fine, (uint32_t)(d < 0 ? -d : d) - 1u);
    });
  } else {
    uint32_t lo = cstart[w * NC + hb], hi = cstart[w * NC + hb + 1];
    if (COOP) {
      lo += sl * slice;
      hi = lo + slice < hi ? lo + slice : hi;
    }
    block_for_each_word(tmp, lo, hi, [&](size_t, ITEM item) { (void)lds_inc(fine, (uint32_t)(item >> (ib + 1))); });
  }
  __syncthreads();
  uint32_t* out = bcnt + (size_t)w * NBP + 1 + ((size_t)hb << fb);
  if (COOP) {
    for (uint32_t b = threadIdx.x; b < NF; b += SORT_THREADS)
      if (fine[b]) atomicAdd(&out[b], fine[b]);
  } else {
    for (uint32_t b = threadIdx.x; b < NF; b += SORT_THREADS) out[b] = fine[b];
    if (hb == 0 && threadIdx.x == 0) bcnt[(size_t)w * NBP] = 0;   // bucket 0 (digit 0) is never used
  }
}

// work list of (region, slice) pairs for regions with more than `big` items
__global__ void msm_big_regions_kernel(const uint32_t* __restrict__ cstart, uint32_t nregions, uint32_t big, uint32_t slice,
                                       uint2* __restrict__ list, uint32_t* __restrict__ list_count, uint32_t capacity) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nregions) return;
  const uint32_t size = cstart[r + 1] - cstart[r];
  if (size <= big) return;
  const uint32_t nsl = (size + slice - 1) / slice;
  const uint32_t at = atomicAdd(list_count, nsl);
  for (uint32_t i = 0; i < nsl && at + i < capacity; ++i) list[at + i] = make_uint2(r, i);
}

// part 2, second half: scatter the region's items to their final places (boff = global bucket offsets)
template <bool FROM_DIGITS, class ITEM>
__global__ __launch_bounds__(SORT_THREADS) void msm_part2_scatter_kernel(const ITEM* __restrict__ tmp,
                                                                         const int32_t* __restrict__ digits,
                                                                         const uint32_t* __restrict__ cstart,
                                                                         const uint32_t* __restrict__ boff,
                                                                         uint32_t* __restrict__ sorted, size_t n, uint32_t fb,
                                                                         uint32_t ib, uint32_t NC, uint32_t NBP) {
  extern __shared__ uint32_t cursor[];
  const uint32_t hb = blockIdx.x, w = blockIdx.y;
  const uint32_t NF = 1u << fb;
  const uint32_t* bo = boff + (size_t)w * NBP + 1 + ((size_t)hb << fb);
  for (uint32_t b = threadIdx.x; b < NF; b += SORT_THREADS) cursor[b] = bo[b];
  __syncthreads();
  if (FROM_DIGITS) {
    const int32_t* dw = digits + (size_t)w * n;
    block_for_each_word(dw, 0, n, [&](size_t i, int32_t d) {
      if (d != 0) {
        const uint32_t pos = lds_inc(cursor, (uint32_t)(d < 0 ? -d : d) - 1u);
        sorted[pos] = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
      }
    });
  } else {
    const uint32_t lo = cstart[w * NC + hb], hi = cstart[w * NC + hb + 1];
    const ITEM imask = ((ITEM)1 << ib) - 1;
    block_for_each_word(tmp, lo, hi, [&](size_t, ITEM item) {
      const uint32_t pos = lds_inc(cursor, (uint32_t)(item >> (ib + 1)));
      sorted[pos] = (uint32_t)(item & imask) | ((uint32_t)((item >> ib) & 1) << 31);
    });
  }
}

// part 2, second half, tiled form (regions read from tmp): the region is processed in tiles of
// 4096 items that are first counting-sorted by fine bucket inside LDS, so that consecutive lanes write
// consecutive addresses of a bucket run -- every 32-byte sector of the final array is written by one
// wave instruction (or two adjacent tiles) instead of by eight separate 4-byte stores spread over the
// workgroup's lifetime, which is what kept missing L2 once a region had more than ~128 runs.
#ifndef HM_P2_IPT
#define HM_P2_IPT 16         // 16 384-item tiles (8: +0.06 ms, 4: +0.35 ms of sort at 2^24)
#endif
constexpr int P2_IPT = HM_P2_IPT;
constexpr int P2_TILE = SORT_THREADS * P2_IPT;
// LDS of the tiled kernels at their largest plans (part 2: fb = 12; part 1's big tiles: 2 048 coarse bins, 4-byte items):
// gfx950 has 160 KiB per workgroup
static_assert(((size_t)3 * 4096 + 32 + P2_TILE) * 4 + (size_t)P2_TILE * 2 <= 160 * 1024, "part-2 tile does not fit the LDS");
static_assert(((size_t)3 * 2048 + 32 + SORT_THREADS * P1_BIG_IPT) * 4 + (size_t)SORT_THREADS * P1_BIG_IPT * 4 <= 160 * 1024,
              "part-1 big tile does not fit the LDS");
// COOP: slices of big regions (work list); the per-bucket cursors then live in global memory
// (`gcursor`, a copy of boff) and every tile reserves its runs with one atomicAdd per non-empty bucket.
// POSITIONAL (a shared bucket set of n * W > 2^27 items: fine bits + sign + a 28-bit index do not fit one word): an
// item carries only the low `ib` bits of its index.  The rest is its SUPER-CHUNK, 2^ib consecutive indices = `gpc`
// chunks of the first sort level, and the first level lays the runs of a coarse bin down in chunk order -- so the
// super-chunk of an item follows from its POSITION in the region: the boundaries are the (already scanned) per-chunk
// offsets chist[g * NC + bin] at g = 0, gpc, 2 gpc, ...  They go to LDS (at most PS_MAX_SC + 1 words) and every item
// finds its piece by a binary search.  Half the bytes of the 64-bit items this replaces, at every level of the sort.
constexpr uint32_t PS_MAX_SC = 1024;
struct Positional {
  const uint32_t* chist;    // scanned per-chunk offsets: [set][g][NC] (a shared bucket set is one "window"; a group of MSMs has one set each)
  uint32_t G, gpc, nsc;     // chunks, chunks per super-chunk, super-chunks (0: items carry their whole index)
};
template <class ITEM, bool COOP = false>
__global__ __launch_bounds__(SORT_THREADS) void msm_part2_scatter_tiled_kernel(const ITEM* __restrict__ tmp,
                                                                               const uint32_t* __restrict__ cstart,
                                                                               const uint32_t* __restrict__ boff,
                                                                               uint32_t* __restrict__ sorted, uint32_t fb,
                                                                               uint32_t ib, uint32_t NC, uint32_t NBP, uint32_t big,
                                                                               uint32_t slice, const uint2* __restrict__ list,
                                                                               const uint32_t* __restrict__ list_count,
                                                                               uint32_t* __restrict__ gcursor, Positional ps) {
  extern __shared__ uint32_t sm[];
  __shared__ uint32_t sc_start[PS_MAX_SC + 1];
  uint32_t hb = blockIdx.x, w = blockIdx.y, sl = 0;
  const uint32_t tid = threadIdx.x;
  if (COOP) {
    if (blockIdx.x >= *list_count) return;
    const uint2 it = list[blockIdx.x];
    w = it.x / NC;
    hb = it.x - w * NC;
    sl = it.y;
  } else if (cstart[w * NC + hb + 1] - cstart[w * NC + hb] > big) {
    return;
  }
  const uint32_t NF = 1u << fb;                       // <= 4096
  uint32_t* gcur = sm;                                // global cursor of every fine bucket
  uint32_t* tcnt = gcur + NF;                         // items of the current tile per bucket
  uint32_t* tstart = tcnt + NF;                       // exclusive prefix of tcnt
  uint32_t* wsum = tstart + NF;                       // 32 wave totals of the scan
  uint32_t* st_pay = wsum + 32;                       // tile items in bucket order
  uint16_t* st_bin = reinterpret_cast<uint16_t*>(st_pay + P2_TILE);   // their fine buckets (< 2^12)
  const uint32_t* bo = boff + (size_t)w * NBP + 1 + ((size_t)hb << fb);
  uint32_t* gc = gcursor + (size_t)w * NBP + 1 + ((size_t)hb << fb);
  for (uint32_t b = tid; b < NF; b += SORT_THREADS) { gcur[b] = bo[b]; tcnt[b] = 0; }
  __syncthreads();
  uint32_t lo = cstart[w * NC + hb], hi = cstart[w * NC + hb + 1];
  const uint32_t region_lo = lo;
  if (ps.nsc) {                                        // region-relative start of every super-chunk's runs
    for (uint32_t sc = tid; sc <= ps.nsc; sc += SORT_THREADS)
      sc_start[sc] = sc < ps.nsc ? ps.chist[((size_t)w * ps.G + (size_t)sc * ps.gpc) * NC + hb] : hi - lo;
    __syncthreads();
  }
  if (COOP) {
    lo += sl * slice;
    hi = lo + slice < hi ? lo + slice : hi;
  }
  const ITEM imask = ((ITEM)1 << ib) - 1;
  const uint32_t per = (NF + SORT_THREADS - 1) / SORT_THREADS;   // scan entries per lane (1 .. 4)
  for (uint32_t t0 = lo; t0 < hi; t0 += P2_TILE) {
    const uint32_t tile_n = hi - t0 < (uint32_t)P2_TILE ? hi - t0 : (uint32_t)P2_TILE;
    uint32_t pay[P2_IPT], bin[P2_IPT], rank[P2_IPT];
    uint32_t sc_tile = 0;
    if (ps.nsc) {                                      // super-chunk of the tile's first position (the same words for every lane:
      const uint32_t pos0 = t0 - region_lo;            // broadcast LDS reads); a tile of 4 096 positions crosses few boundaries
      uint32_t a = 0, b = ps.nsc;                      // invariant: sc_start[a] <= pos0 < sc_start[b]
      while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (sc_start[m] <= pos0) a = m; else b = m;
      }
      sc_tile = a;
    }
#pragma unroll
    for (int k = 0; k < P2_IPT; ++k) {
      const uint32_t e = (uint32_t)k * SORT_THREADS + tid;
      if (e < tile_n) {
        const ITEM item = tmp[t0 + e];
        bin[k] = (uint32_t)(item >> (ib + 1));
        pay[k] = (uint32_t)(item & imask) | ((uint32_t)((item >> ib) & 1) << 31);
        if (ps.nsc) {                                  // the largest sc with sc_start[sc] <= position: a few steps from the tile's
          const uint32_t pos = t0 + e - region_lo;
          uint32_t a = sc_tile;
          while (sc_start[a + 1] <= pos) ++a;          // (pos < sc_start[nsc] = the region's size: the walk ends)
          pay[k] |= a << ib;
        }
        rank[k] = lds_inc(tcnt, bin[k]);
      }
    }
    __syncthreads();
    // exclusive scan of tcnt[0 .. NF): lane-serial over `per` entries, wave scan, then wave totals
    {
      const uint32_t b0 = tid * per;
      uint32_t v[4096 / SORT_THREADS] = {}, sum = 0;
      for (uint32_t k = 0; k < per; ++k)
        if (b0 + k < NF) { v[k] = tcnt[b0 + k]; sum += v[k]; }
      uint32_t incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if ((tid & 63) >= (uint32_t)off) incl += u;
      }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      if (tid < 64) {
        const uint32_t ws = tid < (SORT_THREADS / 64) ? wsum[tid] : 0;
        uint32_t wi = ws;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t u = __shfl_up(wi, off, 64);
          if (tid >= (uint32_t)off) wi += u;
        }
        if (tid < (SORT_THREADS / 64)) wsum[tid] = wi - ws;   // exclusive wave offsets
      }
      __syncthreads();
      uint32_t run = wsum[tid >> 6] + incl - sum;
      for (uint32_t k = 0; k < per; ++k)
        if (b0 + k < NF) { tstart[b0 + k] = run; run += v[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P2_IPT; ++k) {
      const uint32_t e = (uint32_t)k * SORT_THREADS + tid;
      if (e < tile_n) {
        const uint32_t pos = tstart[bin[k]] + rank[k];
        st_pay[pos] = pay[k];
        st_bin[pos] = (uint16_t)bin[k];
      }
    }
    if (COOP)      // reserve this tile's run in every non-empty bucket
      for (uint32_t b = tid; b < NF; b += SORT_THREADS)
        if (tcnt[b]) gcur[b] = atomicAdd(&gc[b], tcnt[b]);
    __syncthreads();
    for (uint32_t e = tid; e < tile_n; e += SORT_THREADS) {
      const uint32_t b = st_bin[e];
      sorted[gcur[b] + (e - tstart[b])] = st_pay[e];
    }
    __syncthreads();
    for (uint32_t b = tid; b < NF; b += SORT_THREADS) {
      if (!COOP) gcur[b] += tcnt[b];
      tcnt[b] = 0;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The tiled scatters, second form ("v2", 32-bit items; round 4).  SQ counters on the kernels above at 2^24 on the table
// (profiles/r04_b_side_measurements.txt) showed both of them bound by instruction issue and latency, not by HBM: 2.3
// wave-instructions per item (147 lane-instructions), and in the ISA every one of a lane's 16 loads of a tile sits in its
// own branch region behind its own `s_waitcnt vmcnt(0)` -- sixteen serial HBM round trips per tile (a tile took 32 us) --
// and the positional walk is a divergent loop of dependent LDS reads per item.  Here a lane takes FOUR consecutive items
// per 16-byte load, all loads of a tile are issued before anything waits, the super-chunk of an item comes from a per-tile
// table (one entry per 64 positions, filled by a few lanes while the loads are in flight) plus a compare against the next
// boundary, and the wave-uniformity test that guards the LDS counters is made once per four items.
// ---------------------------------------------------------------------------------------------
// A workgroup barrier that orders LDS accesses only: __syncthreads() also waits for every outstanding global access
// (s_waitcnt vmcnt(0)), which would drain a tile's scattered stores -- and the next tile's prefetched loads -- at each of
// the six barriers of a tile.  The v2 scatters never read back what they store to global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Exclusive scan of cnt[0 .. NB) (NB <= 4096) into start[], by the whole workgroup; returns the total to every lane.
// Three barriers; cnt is left untouched.
__device__ __forceinline__ uint32_t tile_exclusive_scan(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ start,
                                                        uint32_t* __restrict__ wsum, uint32_t NB, uint32_t tid) {
  const uint32_t per = (NB + SORT_THREADS - 1) / SORT_THREADS, b0 = tid * per;
  uint32_t v[4096 / SORT_THREADS] = {}, sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4096 / SORT_THREADS; ++k)
    if (k < per && b0 + k < NB) { v[k] = cnt[b0 + k]; sum += v[k]; }
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(incl, off, 64);
    if ((tid & 63) >= (uint32_t)off) incl += u;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  lds_barrier();
  if (tid < 64) {
    const uint32_t ws = tid < (SORT_THREADS / 64) ? wsum[tid] : 0;
    uint32_t wi = ws;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(wi, off, 64);
      if (tid >= (uint32_t)off) wi += u;
    }
    if (tid < (SORT_THREADS / 64)) wsum[tid] = wi - ws;   // exclusive wave offsets
    if (tid == SORT_THREADS / 64 - 1) wsum[SORT_THREADS / 64] = wi;   // the total
  }
  lds_barrier();
  uint32_t run = wsum[tid >> 6] + incl - sum;
#pragma unroll
  for (uint32_t k = 0; k < 4096 / SORT_THREADS; ++k)
    if (k < per && b0 + k < NB) { start[b0 + k] = run; run += v[k]; }
  const uint32_t total = wsum[SORT_THREADS / 64];
  lds_barrier();
  return total;
}

// rank[k] = cnt[bin[k]]++ for the valid ones of a lane's four items (FULL: all four are).  When every item of the WAVE
// goes to one counter (constant and flag columns: lds_inc's reason) one lane adds for all of them.  Called with the whole
// wave converged.
template <bool FULL>
__device__ __forceinline__ void lds_rank4(uint32_t* cnt, const uint32_t bin[4], uint32_t valid_mask, uint32_t rank[4]) {
  if (FULL) valid_mask = 0xfu;
  const uint64_t active = FULL ? ~0ull : __ballot(valid_mask != 0);
  if (!FULL && active == 0) return;                            // wave-uniform
  uint32_t mine = bin[0];
  if (!FULL) {
#pragma unroll
    for (int k = 3; k >= 0; --k)
      if (valid_mask & (1u << k)) mine = bin[k];               // a valid bin of this lane (its first)
  }
  const uint32_t v = FULL ? (uint32_t)__builtin_amdgcn_readfirstlane((int)mine)
                          : (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(active));
  bool same = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) same = same && (!(valid_mask & (1u << k)) || bin[k] == v);
  if (__ballot(same) == ~0ull) {                               // every valid item of the wave wants counter v
    const uint32_t mycount = (uint32_t)__popc(valid_mask);
    uint32_t incl = mycount;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(incl, off, 64);
      if ((threadIdx.x & 63) >= (uint32_t)off) incl += u;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(&cnt[v], total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    uint32_t r = base + incl - mycount;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (valid_mask & (1u << k)) rank[k] = r++;
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (FULL || (valid_mask & (1u << k))) rank[k] = atomicAdd(&cnt[bin[k]], 1u);
}

// One tile of the first-level scatter.  FULL: every position of the tile is a digit of the chunk (all but the first and
// last tile of a chunk): no masks, no predication.  `base + r` is digit lo + r - mis.
template <int IPT>
__device__ __forceinline__ void p1v2_load(const int32_t* __restrict__ base, uint32_t rb, uint32_t end, uint32_t tid, int4 (&q)[IPT / 4]) {
#pragma unroll
  for (int j = 0; j < IPT / 4; ++j) {
    const uint32_t r = rb + 4u * ((uint32_t)j * SORT_THREADS + tid);
    q[j] = *reinterpret_cast<const int4*>(base + (r < end ? r : 0u));      // (a vector past the end reads the first one again)
  }
}
template <int IPT, bool FULL>
__device__ __forceinline__ void p1v2_tile(const int4 (&q)[IPT / 4], uint32_t rb, uint32_t mis, uint32_t end, uint32_t lo32,
                                          uint32_t fb, uint32_t ib, uint32_t NC, uint32_t* __restrict__ tmp, uint32_t* gcur, uint32_t* tcnt,
                                          uint32_t* tstart, uint32_t* wsum, uint32_t* st_bin, uint32_t* st_item, uint32_t tid) {
  constexpr int NV = IPT / 4;
  const uint32_t fmask = (1u << fb) - 1u, imask = (1u << ib) - 1u;
  uint32_t rank[IPT], vmask[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint32_t r = rb + 4u * ((uint32_t)j * SORT_THREADS + tid);
    const int32_t d4[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
    uint32_t m = 0, bin[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int32_t d = d4[c];
      const bool ok = (FULL || (r + c >= mis && r + c < end)) && d != 0;      // zero digits are not items
      bin[c] = ((uint32_t)(d < 0 ? -d : d) - 1u) >> fb;
      m |= ok ? 1u << c : 0u;
    }
    vmask[j] = m;
    if (__ballot(m != 0xfu) == 0) lds_rank4<true>(tcnt, bin, m, &rank[4 * j]);     // no zero digit in the wave's 256: the usual case
    else lds_rank4<false>(tcnt, bin, m, &rank[4 * j]);
  }
  lds_barrier();
  const uint32_t kept = tile_exclusive_scan(tcnt, tstart, wsum, NC, tid);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint32_t r = rb + 4u * ((uint32_t)j * SORT_THREADS + tid);
    const int32_t d4[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (vmask[j] & (1u << c)) {
        const int32_t d = d4[c];
        const uint32_t b1 = (uint32_t)(d < 0 ? -d : d) - 1u, bin = b1 >> fb;
        const uint32_t pos = lo32 + (r + c - mis);                              // the item's index in its bucket set (low `ib` bits kept)
        const uint32_t dst = tstart[bin] + rank[4 * j + c];
        st_item[dst] = ((b1 & fmask) << (ib + 1)) | ((d < 0 ? 1u : 0u) << ib) | (pos & imask);
        st_bin[dst] = bin;
      }
  }
  lds_barrier();
  if (kept == (uint32_t)(SORT_THREADS * IPT)) {
#pragma unroll 4
    for (int k = 0; k < IPT; ++k) {
      const uint32_t e = (uint32_t)k * SORT_THREADS + tid, b = st_bin[e];
      tmp[gcur[b] + (e - tstart[b])] = st_item[e];
    }
  } else {
    for (uint32_t e = tid; e < kept; e += SORT_THREADS) {
      const uint32_t b = st_bin[e];
      tmp[gcur[b] + (e - tstart[b])] = st_item[e];
    }
  }
  lds_barrier();
  for (uint32_t b = tid; b < NC; b += SORT_THREADS) { gcur[b] += tcnt[b]; tcnt[b] = 0; }
  lds_barrier();
}

template <int IPT>
__global__ __launch_bounds__(SORT_THREADS) void msm_part1_scatter_v2_kernel(const int32_t* __restrict__ digits,
                                                                            const uint32_t* __restrict__ chist,
                                                                            const uint32_t* __restrict__ cstart,
                                                                            uint32_t* __restrict__ tmp, size_t n, size_t chunk,
                                                                            uint32_t fb, uint32_t ib, uint32_t NC) {
  extern __shared__ uint32_t sm[];
  constexpr uint32_t TILE = SORT_THREADS * IPT;
  const uint32_t g = blockIdx.x, w = blockIdx.y, G = gridDim.x, tid = threadIdx.x;
  uint32_t* gcur = sm;                 // global cursor of every coarse bin for this chunk
  uint32_t* tcnt = gcur + NC;
  uint32_t* tstart = tcnt + NC;
  uint32_t* wsum = tstart + NC;        // 32 words
  uint32_t* st_bin = wsum + 32;
  uint32_t* st_item = st_bin + TILE;
  const uint32_t* pre = chist + ((size_t)w * G + g) * NC;
  const uint32_t* cs = cstart + (size_t)w * NC;
  for (uint32_t b = tid; b < NC; b += SORT_THREADS) { gcur[b] = cs[b] + pre[b]; tcnt[b] = 0; }
  lds_barrier();
  const size_t lo = (size_t)g * chunk, hi = lo + chunk < n ? lo + chunk : n;
  if (lo >= hi) return;
  // tiles start where the ADDRESS is 16-byte aligned: `mis` <= 3 positions before lo (masked off; for w > 0 they are the
  // previous window's last digits, for w = 0 the array is 256-byte aligned and mis = lo & 3).  Positions are counted from
  // that address: r in [mis, mis + len) is digit lo + r - mis.
  const int32_t* dw = digits + (size_t)w * n + lo;
  const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(dw) >> 2) & 3);
  const int32_t* base = dw - mis;
  const uint32_t end = mis + (uint32_t)(hi - lo);              // chunks are far below 2^32 items
  int4 q[IPT / 4], qn[IPT / 4];
  p1v2_load<IPT>(base, 0, end, tid, q);
  for (uint32_t rb = 0; rb < end; rb += TILE) {
    const bool more = rb + TILE < end;
    if (more) p1v2_load<IPT>(base, rb + TILE, end, tid, qn);     // the next tile's digits travel while this one is ranked and staged
    if (rb >= mis && rb + TILE <= end)
      p1v2_tile<IPT, true>(q, rb, mis, end, (uint32_t)lo, fb, ib, NC, tmp, gcur, tcnt, tstart, wsum, st_bin, st_item, tid);
    else
      p1v2_tile<IPT, false>(q, rb, mis, end, (uint32_t)lo, fb, ib, NC, tmp, gcur, tcnt, tstart, wsum, st_bin, st_item, tid);
    if (more) {
#pragma unroll
      for (int j = 0; j < IPT / 4; ++j) q[j] = qn[j];
    }
  }
}

__device__ __forceinline__ uint32_t dst_of(const uint32_t* tstart, uint32_t bin, uint32_t rank) { return tstart[bin] + rank; }

// One tile of the second-level scatter.  FULL: every position of the tile is an item of the region / slice.
__device__ __forceinline__ void p2v2_load(const uint32_t* __restrict__ tmp, uint32_t tb, uint32_t lo, uint32_t hi, uint32_t tid,
                                          uint4 (&q)[P2_IPT / 4]) {
#pragma unroll
  for (int j = 0; j < P2_IPT / 4; ++j) {
    const uint32_t p = tb + 4u * ((uint32_t)j * SORT_THREADS + tid);
    q[j] = *reinterpret_cast<const uint4*>(tmp + (p < hi ? p : (lo & ~3u)));
  }
}
template <bool COOP, bool FULL>
__device__ __forceinline__ void p2v2_tile(const uint4 (&q)[P2_IPT / 4], uint32_t* __restrict__ sorted, uint32_t tb, uint32_t lo,
                                          uint32_t hi, uint32_t region_lo, uint32_t ib, uint32_t NF, const Positional& ps, uint32_t* gcur,
                                          uint32_t* tcnt, uint32_t* tstart, uint32_t* wsum, uint32_t* st_pay, uint16_t* st_bin,
                                          const uint32_t* sc_start, uint32_t* tile_sc, uint32_t* __restrict__ gc, uint32_t tid) {
  constexpr int NV = P2_IPT / 4;
  const uint32_t imask = (1u << ib) - 1u;
  if (ps.nsc && tid < P2_TILE / 64) {                  // while the loads are in flight: the largest sc with sc_start[sc] <= position
    const uint32_t p0 = tb + 64u * tid;
    const uint32_t rel = p0 > region_lo ? p0 - region_lo : 0u;
    uint32_t a = 0, b = ps.nsc;                        // invariant: sc_start[a] <= rel (sc_start[0] = 0), and rel < sc_start[b] or b = nsc
    while (b - a > 1) {
      const uint32_t m = (a + b) >> 1;
      if (sc_start[m] <= rel) a = m; else b = m;
    }
    tile_sc[tid] = a;
  }
  uint32_t rank[P2_IPT], vmask[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint32_t p = tb + 4u * ((uint32_t)j * SORT_THREADS + tid);
    const uint32_t i4[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
    uint32_t m = 0, bin[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bin[c] = i4[c] >> (ib + 1);
      m |= (FULL || (p + c >= lo && p + c < hi)) ? 1u << c : 0u;
    }
    vmask[j] = m;
    lds_rank4<FULL>(tcnt, bin, m, &rank[4 * j]);
  }
  lds_barrier();
  const uint32_t kept = tile_exclusive_scan(tcnt, tstart, wsum, NF, tid);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const uint32_t p = tb + 4u * ((uint32_t)j * SORT_THREADS + tid);
    const uint32_t i4[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
    uint32_t a = 0, nb = 0xffffffffu;
    if (ps.nsc && (FULL || vmask[j])) {                // super-chunk of the vector's first position: the tile table, then a short walk
      const uint32_t rel = (p > region_lo ? p - region_lo : 0u);
      a = tile_sc[(p - tb) >> 6];
      while (sc_start[a + 1] <= rel && a + 1 < ps.nsc) ++a;
      nb = sc_start[a + 1];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (FULL || (vmask[j] & (1u << c))) {
        uint32_t pay = (i4[c] & imask) | (((i4[c] >> ib) & 1u) << 31);
        if (ps.nsc) {
          const uint32_t rel = p + c - region_lo;
          while (nb <= rel && a + 1 < ps.nsc) { ++a; nb = sc_start[a + 1]; }   // almost never: runs are ~2 000 items long
          pay |= a << ib;
        }
        const uint32_t bin = i4[c] >> (ib + 1);
        st_pay[dst_of(tstart, bin, rank[4 * j + c])] = pay;
        st_bin[dst_of(tstart, bin, rank[4 * j + c])] = (uint16_t)bin;
      }
  }
  if (COOP)      // reserve this tile's run in every non-empty bucket
    for (uint32_t b = tid; b < NF; b += SORT_THREADS)
      if (tcnt[b]) gcur[b] = atomicAdd(&gc[b], tcnt[b]);
  lds_barrier();
  if (FULL) {
#pragma unroll 4
    for (int k = 0; k < P2_IPT; ++k) {
      const uint32_t e = (uint32_t)k * SORT_THREADS + tid, b = st_bin[e];
      sorted[gcur[b] + (e - tstart[b])] = st_pay[e];
    }
  } else {
    for (uint32_t e = tid; e < kept; e += SORT_THREADS) {
      const uint32_t b = st_bin[e];
      sorted[gcur[b] + (e - tstart[b])] = st_pay[e];
    }
  }
  lds_barrier();
  for (uint32_t b = tid; b < NF; b += SORT_THREADS) {
    if (!COOP) gcur[b] += tcnt[b];
    tcnt[b] = 0;
  }
  lds_barrier();
}

template <bool COOP>
__global__ __launch_bounds__(SORT_THREADS) void msm_part2_scatter_v2_kernel(const uint32_t* __restrict__ tmp,
                                                                            const uint32_t* __restrict__ cstart,
                                                                            const uint32_t* __restrict__ boff,
                                                                            uint32_t* __restrict__ sorted, uint32_t fb,
                                                                            uint32_t ib, uint32_t NC, uint32_t NBP, uint32_t big,
                                                                            uint32_t slice, const uint2* __restrict__ list,
                                                                            const uint32_t* __restrict__ list_count,
                                                                            uint32_t* __restrict__ gcursor, Positional ps) {
  extern __shared__ uint32_t sm[];
  __shared__ uint32_t sc_start[PS_MAX_SC + 2];
  __shared__ uint32_t tile_sc[P2_TILE / 64];                  // super-chunk of every 64th position of the current tile
  uint32_t hb = blockIdx.x, w = blockIdx.y, sl = 0;
  const uint32_t tid = threadIdx.x;
  if (COOP) {
    if (blockIdx.x >= *list_count) return;
    const uint2 it = list[blockIdx.x];
    w = it.x / NC;
    hb = it.x - w * NC;
    sl = it.y;
  } else if (cstart[w * NC + hb + 1] - cstart[w * NC + hb] > big) {
    return;
  }
  const uint32_t NF = 1u << fb;                       // <= 4096
  uint32_t* gcur = sm;                                // global cursor of every fine bucket
  uint32_t* tcnt = gcur + NF;                         // items of the current tile per bucket
  uint32_t* tstart = tcnt + NF;                       // exclusive prefix of tcnt
  uint32_t* wsum = tstart + NF;                       // 32 words
  uint32_t* st_pay = wsum + 32;                       // tile items in bucket order
  uint16_t* st_bin = reinterpret_cast<uint16_t*>(st_pay + P2_TILE);   // their fine buckets (< 2^12)
  const uint32_t* bo = boff + (size_t)w * NBP + 1 + ((size_t)hb << fb);
  uint32_t* gc = gcursor + (size_t)w * NBP + 1 + ((size_t)hb << fb);
  for (uint32_t b = tid; b < NF; b += SORT_THREADS) { gcur[b] = bo[b]; tcnt[b] = 0; }
  uint32_t lo = cstart[w * NC + hb], hi = cstart[w * NC + hb + 1];
  const uint32_t region_lo = lo, region_n = hi - lo;
  if (ps.nsc) {                                        // region-relative start of every super-chunk's runs; a sentinel behind them
    for (uint32_t sc = tid; sc <= ps.nsc + 1; sc += SORT_THREADS)
      sc_start[sc] = sc < ps.nsc ? ps.chist[((size_t)w * ps.G + (size_t)sc * ps.gpc) * NC + hb] : (sc == ps.nsc ? region_n : 0xffffffffu);
  }
  lds_barrier();
  if (COOP) {
    lo += sl * slice;
    hi = lo + slice < hi ? lo + slice : hi;
  }
  if (lo >= hi) return;
  uint4 q[P2_IPT / 4], qn[P2_IPT / 4];
  p2v2_load(tmp, lo & ~3u, lo, hi, tid, q);
  for (uint32_t tb = lo & ~3u; tb < hi; tb += P2_TILE) {   // tmp is 256-byte aligned: positions that are multiples of 4 are 16-byte addresses
    const bool more = tb + P2_TILE < hi;
    if (more) p2v2_load(tmp, tb + P2_TILE, lo, hi, tid, qn);     // the next tile's items travel while this one is ranked and staged
    if (tb >= lo && tb + P2_TILE <= hi)
      p2v2_tile<COOP, true>(q, sorted, tb, lo, hi, region_lo, ib, NF, ps, gcur, tcnt, tstart, wsum, st_pay, st_bin, sc_start, tile_sc, gc, tid);
    else
      p2v2_tile<COOP, false>(q, sorted, tb, lo, hi, region_lo, ib, NF, ps, gcur, tcnt, tstart, wsum, st_pay, st_bin, sc_start, tile_sc, gc, tid);
    if (more) {
#pragma unroll
      for (int j = 0; j < P2_IPT / 4; ++j) q[j] = qn[j];
    }
  }
}

// Measured negative in round 6 (profiles/r06_sort_ab.txt): the second-level scatter writes 4-item = 16-byte runs at 2^24 on the table
// (2.1 x its bytes in WRITE_SIZE; with every tile stored where it was read the kernel takes 0.48 ms instead of 0.83 -- the ceiling).
// A form with carry slots in LDS (8 words per bucket, items leave in whole aligned 32-byte sectors; needs fb <= 11, so one more coarse
// bit and half-size tiles) was correct and SLOWER than the plain scatter on the same plan, 906 against 716 us: two more per-bucket
// passes and a barrier per tile cost more than the full sectors save; ten coarse bits alone only move 0.13 ms from this level to the
// first.  The plan stays 9 + 12 bits.

// bucket offsets (exclusive scan of counts) and task offsets (exclusive scan of ceil(count / L)):
// three small launches -- per-block sums, a one-block scan of those, per-block rescan + offset.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;                       // items per lane; a block covers 2048 buckets
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ void block_scan_pair(uint32_t& p, uint32_t& t, uint32_t* s_p, uint32_t* s_t) {
  // inclusive scan over the block of (p, t); returns inclusive values in p, t
  const uint32_t tid = threadIdx.x;
  s_p[tid] = p;
  s_t[tid] = t;
  __syncthreads();
  for (uint32_t off = 1; off < SCAN_THREADS; off <<= 1) {
    uint32_t a = 0, b = 0;
    if (tid >= off) { a = s_p[tid - off]; b = s_t[tid - off]; }
    __syncthreads();
    s_p[tid] += a;
    s_t[tid] += b;
    __syncthreads();
  }
  p = s_p[tid];
  t = s_t[tid];
}

__global__ __launch_bounds__(SCAN_THREADS) void msm_scan_partial_kernel(const uint32_t* __restrict__ bcnt,
                                                                        uint32_t* __restrict__ blocksums, uint32_t NBT,
                                                                        const uint32_t* __restrict__ pairs, uint32_t L_host) {
  __shared__ uint32_t s_p[SCAN_THREADS], s_t[SCAN_THREADS];
  const uint32_t L = effective_task_len(pairs, L_host);
  const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
  uint32_t sp = 0, st = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const uint32_t c = base + k < NBT ? bcnt[base + k] : 0u;
    sp += c;
    st += (c + L - 1) / L;
  }
  block_scan_pair(sp, st, s_p, s_t);
  if (threadIdx.x == SCAN_THREADS - 1) {
    blocksums[2 * blockIdx.x] = sp;
    blocksums[2 * blockIdx.x + 1] = st;
  }
}

// one block: exclusive scan of the per-block sums in place; totals[0] = pairs, totals[1] = tasks.
// totals[2] = 1 and totals[1] = 0 if the task count exceeds the capacity of the task arrays (a host-side
// bound that cannot be reached; if it ever were, every later kernel sees the flag and touches nothing,
// and msm_finish reports the error -- no out-of-bounds write happens first).
__global__ __launch_bounds__(SCAN_THREADS) void msm_scan_blocksums_kernel(uint32_t* __restrict__ blocksums, uint32_t nblocks,
                                                                          uint32_t* __restrict__ totals, uint32_t task_capacity) {
  __shared__ uint32_t s_p[SCAN_THREADS], s_t[SCAN_THREADS];
  uint32_t carry_p = 0, carry_t = 0;
  for (uint32_t start = 0; start < nblocks; start += SCAN_THREADS) {
    const uint32_t i = start + threadIdx.x;
    uint32_t p = i < nblocks ? blocksums[2 * i] : 0u, t = i < nblocks ? blocksums[2 * i + 1] : 0u;
    const uint32_t p0 = p, t0 = t;
    block_scan_pair(p, t, s_p, s_t);
    if (i < nblocks) {
      blocksums[2 * i] = carry_p + p - p0;
      blocksums[2 * i + 1] = carry_t + t - t0;
    }
    carry_p += s_p[SCAN_THREADS - 1];
    carry_t += s_t[SCAN_THREADS - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool over = carry_t > task_capacity;
    totals[0] = carry_p;
    totals[1] = over ? 0u : carry_t;
    totals[2] = over ? 1u : 0u;
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void msm_scan_final_kernel(const uint32_t* __restrict__ bcnt,
                                                                      const uint32_t* __restrict__ blocksums,
                                                                      uint32_t* __restrict__ boff, uint32_t* __restrict__ toff,
                                                                      const uint32_t* __restrict__ totals, uint32_t NBT,
                                                                      const uint32_t* __restrict__ pairs, uint32_t L_host,
                                                                      uint32_t* __restrict__ gcursor) {
  __shared__ uint32_t s_p[SCAN_THREADS], s_t[SCAN_THREADS];
  const uint32_t L = effective_task_len(pairs, L_host);
  const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
  uint32_t c[SCAN_ITEMS];
  uint32_t sp = 0, st = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    c[k] = base + k < NBT ? bcnt[base + k] : 0u;
    sp += c[k];
    st += (c[k] + L - 1) / L;
  }
  const uint32_t sp0 = sp, st0 = st;
  block_scan_pair(sp, st, s_p, s_t);
  uint32_t rp = blocksums[2 * blockIdx.x] + sp - sp0, rt = blocksums[2 * blockIdx.x + 1] + st - st0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < NBT) {
      boff[base + k] = rp;
      toff[base + k] = rt;
      if (gcursor) gcursor[base + k] = rp;   // the cooperative scatter's global per-bucket cursors start at the offsets
    }
    rp += c[k];
    rt += (c[k] + L - 1) / L;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) toff[NBT] = totals[1];
}
// (with the overflow flag set toff keeps the unclamped offsets, so every consumer of toff checks totals[2] first)

// bucket order (small inputs): task t runs in slot t
__global__ void msm_task_fill_kernel(const uint32_t* __restrict__ toff, uint32_t* __restrict__ task_bucket,
                                     uint32_t* __restrict__ task_order, uint32_t NBT, const uint32_t* __restrict__ totals) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= NBT || totals[2]) return;
  const uint32_t lo = toff[b], hi = toff[b + 1];
  for (uint32_t t = lo; t < hi; ++t) {
    task_bucket[t] = b;
    task_order[t] = t;
  }
}

// Task list in order of decreasing chain length.  The 64 lanes of a K3 wave run in lockstep until the
// longest of their chains ends, so a wave of tasks taken in bucket order costs max-of-64 bucket sizes
// (for uniform scalars: mean + ~2.4 sigma, 12 % of the lane-cycles idle).  A counting sort of the tasks
// by length (1024 keys, descending) puts equal-length chains side by side, and starting with the longest
// also shortens the tail of the launch.  Only the ORDER of execution changes: task t still owns
// partial[t], so nothing downstream depends on the (atomic, run-to-run varying) order within one key.
constexpr uint32_t TASK_KEYS = 1024;
constexpr int ORDER_THREADS = 1024;   // == TASK_KEYS: one LDS counter per lane
constexpr int ORDER_ITEMS = 4;        // buckets per lane
constexpr uint32_t ORDER_HOT = 32;    // hot buckets a workgroup fills cooperatively (more fall back to their lane)

__device__ __forceinline__ uint32_t task_key(uint32_t len, uint32_t L) {   // len in [1, L]; key 0 = longest
  const uint32_t k = L < TASK_KEYS ? len : (uint32_t)(((uint64_t)len * (TASK_KEYS - 1)) / L);
  return (TASK_KEYS - 1) - k;
}

__global__ __launch_bounds__(ORDER_THREADS) void msm_task_hist_kernel(const uint32_t* __restrict__ bcnt, uint32_t NBT,
                                                                      const uint32_t* __restrict__ pairs, uint32_t L_host,
                                                                      uint32_t* __restrict__ khist) {
  __shared__ uint32_t h[TASK_KEYS];
  const uint32_t L = effective_task_len(pairs, L_host);
  h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ORDER_ITEMS; ++k) {
    const uint32_t b = (blockIdx.x * ORDER_ITEMS + k) * ORDER_THREADS + threadIdx.x;
    if (b < NBT) {
      const uint32_t c = bcnt[b], full = c / L, rem = c - full * L;
      if (full) atomicAdd(&h[task_key(L, L)], full);
      if (rem) atomicAdd(&h[task_key(rem, L)], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&khist[threadIdx.x], h[threadIdx.x]);
}

// one block: khist -> exclusive start of every key, left in kcursor for the scatter to advance
__global__ __launch_bounds__(ORDER_THREADS) void msm_task_keyscan_kernel(const uint32_t* __restrict__ khist,
                                                                         uint32_t* __restrict__ kcursor) {
  __shared__ uint32_t s[TASK_KEYS];
  const uint32_t tid = threadIdx.x, v = khist[tid];
  s[tid] = v;
  __syncthreads();
  for (uint32_t off = 1; off < TASK_KEYS; off <<= 1) {
    const uint32_t a = tid >= off ? s[tid - off] : 0u;
    __syncthreads();
    s[tid] += a;
    __syncthreads();
  }
  kcursor[tid] = s[tid] - v;
}

__global__ __launch_bounds__(ORDER_THREADS) void msm_task_order_kernel(const uint32_t* __restrict__ bcnt,
                                                                       const uint32_t* __restrict__ toff, uint32_t NBT,
                                                                       const uint32_t* __restrict__ pairs, uint32_t L_host,
                                                                       uint32_t* __restrict__ kcursor,
                                                                       uint32_t* __restrict__ task_bucket,
                                                                       uint32_t* __restrict__ task_order,
                                                                       const uint32_t* __restrict__ totals) {
  __shared__ uint32_t h[TASK_KEYS], base[TASK_KEYS];
  if (totals[2]) return;                   // task arrays too small (see msm_scan_blocksums_kernel): write nothing
  const uint32_t L = effective_task_len(pairs, L_host);
  h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t cnt[ORDER_ITEMS];
#pragma unroll
  for (int k = 0; k < ORDER_ITEMS; ++k) {
    const uint32_t b = (blockIdx.x * ORDER_ITEMS + k) * ORDER_THREADS + threadIdx.x;
    cnt[k] = b < NBT ? bcnt[b] : 0u;
    const uint32_t full = cnt[k] / L, rem = cnt[k] - full * L;
    if (full) atomicAdd(&h[task_key(L, L)], full);
    if (rem) atomicAdd(&h[task_key(rem, L)], 1u);
  }
  __syncthreads();
  {   // one global reservation per key and workgroup; h becomes the local cursor
    const uint32_t mine = h[threadIdx.x];
    base[threadIdx.x] = mine ? atomicAdd(&kcursor[threadIdx.x], mine) : 0u;
    h[threadIdx.x] = 0;
  }
  __syncthreads();
  // a bucket cut into many full-length tasks (a hot bucket) is written by the whole workgroup
  __shared__ uint32_t hot_n, hot[ORDER_HOT][4];
  if (threadIdx.x == 0) hot_n = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ORDER_ITEMS; ++k) {
    const uint32_t b = (blockIdx.x * ORDER_ITEMS + k) * ORDER_THREADS + threadIdx.x;
    if (cnt[k] == 0) continue;
    const uint32_t full = cnt[k] / L, rem = cnt[k] - full * L;
    const uint32_t t0 = toff[b];
    if (full) {
      const uint32_t key = task_key(L, L), at = base[key] + atomicAdd(&h[key], full);
      const uint32_t slot = full > 64 ? atomicAdd(&hot_n, 1u) : ORDER_HOT;
      if (slot < ORDER_HOT) {
        hot[slot][0] = b; hot[slot][1] = t0; hot[slot][2] = full; hot[slot][3] = at;
      } else {
        for (uint32_t i = 0; i < full; ++i) {
          task_bucket[t0 + i] = b;
          task_order[at + i] = t0 + i;
        }
      }
    }
    if (rem) {
      const uint32_t key = task_key(rem, L);
      task_bucket[t0 + full] = b;
      task_order[base[key] + atomicAdd(&h[key], 1u)] = t0 + full;
    }
  }
  __syncthreads();
  const uint32_t nh = hot_n < ORDER_HOT ? hot_n : ORDER_HOT;
  for (uint32_t e = 0; e < nh; ++e) {
    const uint32_t b = hot[e][0], t0 = hot[e][1], full = hot[e][2], at = hot[e][3];
    for (uint32_t i = threadIdx.x; i < full; i += ORDER_THREADS) {
      task_bucket[t0 + i] = b;
      task_order[at + i] = t0 + i;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3: bucket accumulation -- one lane per task, serial chain of mixed additions
// ---------------------------------------------------------------------------------------------
#ifdef HM_K3_WAVES            // A/B only (tools/ab_build.sh): force an occupancy, i.e. a register budget, on the hot kernel
#define HM_K3_OCC __attribute__((amdgpu_waves_per_eu(HM_K3_WAVES, HM_K3_WAVES)))
#else
#define HM_K3_OCC
#endif
__global__ __launch_bounds__(ACC_THREADS) HM_K3_OCC void msm_accumulate_kernel(const uint32_t* __restrict__ sorted,
                                                                     const uint32_t* __restrict__ task_bucket,
                                                                     const uint32_t* __restrict__ task_order,
                                                                     const uint32_t* __restrict__ boff,
                                                                     const uint32_t* __restrict__ bcnt,
                                                                     const uint32_t* __restrict__ toff,
                                                                     const uint32_t* __restrict__ xy,
                                                                     uint32_t* __restrict__ partial,
                                                                     const uint32_t* __restrict__ totals,
                                                                     const uint32_t* __restrict__ pairs, uint32_t L_host) {
  __shared__ uint4 stage[4][ACC_THREADS];
  const uint32_t L = effective_task_len(pairs, L_host);
  const uint32_t lane = threadIdx.x;
  const uint32_t slot = blockIdx.x * ACC_THREADS + lane;
  const uint32_t T = totals[1];   // task count from the scan: the grid is sized by its host-side upper bound
  uint32_t start = 0, end = 0, t = 0;
  if (slot < T) {
    t = task_order[slot];       // longest chains first, equal lengths side by side
    const uint32_t b = task_bucket[t];
    const uint32_t k = t - toff[b];
    start = boff[b] + k * L;
    const uint32_t bucket_end = boff[b] + bcnt[b];
    end = start + L < bucket_end ? start + L : bucket_end;
  }
  if (start >= end) return;       // no task for this lane (the kernel has no barriers)
  const G1Jac res = accumulate_chain(sorted, xy, start, end, stage, lane);
  store_jac(partial + (size_t)t * PT_WORDS, res);
}


// K3b: bucket = sum of its task partials.  Buckets with at most FINALIZE_SERIAL partials are
// summed by one lane; longer ones (hot buckets that K3 split into many tasks) are queued for
// the workgroup-per-bucket kernel below so that no single lane walks a long chain.
constexpr uint32_t FINALIZE_SERIAL = 8;
constexpr uint32_t FINALIZE_SLICE = 2048;   // partials one workgroup sums in the first level of a very hot bucket
__global__ __launch_bounds__(ACC_THREADS) void msm_bucket_finalize_kernel(const uint32_t* __restrict__ partial,
                                                                          const uint32_t* __restrict__ toff,
                                                                          uint32_t* __restrict__ bucket, uint32_t NBT,
                                                                          uint32_t* __restrict__ big_count,
                                                                          uint32_t* __restrict__ big_list,
                                                                          uint2* __restrict__ slice_list,
                                                                          const uint32_t* __restrict__ totals) {
  const uint32_t b = blockIdx.x * ACC_THREADS + threadIdx.x;
  if (b >= NBT) return;
  if (totals[2]) {                         // overflow flag: no partial was written; leave an identity in every bucket
    store_jac(bucket + (size_t)b * PT_WORDS, g1_identity());
    return;
  }
  const uint32_t lo = toff[b], hi = toff[b + 1];
  if (hi - lo > FINALIZE_SERIAL) {
    big_list[atomicAdd(big_count, 1u)] = b;
    if (hi - lo > FINALIZE_SLICE) {   // very hot: its partials are first summed slice by slice, many workgroups
      const uint32_t nsl = (hi - lo + FINALIZE_SLICE - 1) / FINALIZE_SLICE;
      const uint32_t at = atomicAdd(big_count + 1, nsl);
      for (uint32_t k = 0; k < nsl; ++k) slice_list[at + k] = make_uint2(b, k);
    }
    return;
  }
  G1Jac acc = g1_identity();
  for (uint32_t t = lo; t < hi; ++t) acc = g1_add(acc, load_jac(partial + (size_t)t * PT_WORDS));
  store_jac(bucket + (size_t)b * PT_WORDS, acc);
}

// first level for very hot buckets: slice k of bucket b = partials [lo + k*SLICE, +SLICE) -> their sum,
// written over the slice's first partial (each slice belongs to one workgroup)
__global__ __launch_bounds__(WIN_THREADS) void msm_bucket_finalize_slices_kernel(uint32_t* __restrict__ partial,
                                                                                 const uint32_t* __restrict__ toff,
                                                                                 const uint32_t* __restrict__ big_count,
                                                                                 const uint2* __restrict__ slice_list) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  const uint32_t count = big_count[1];
  for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    const uint2 e = slice_list[item];
    const uint32_t lo = toff[e.x] + e.y * FINALIZE_SLICE;
    const uint32_t end = toff[e.x + 1], hi = lo + FINALIZE_SLICE < end ? lo + FINALIZE_SLICE : end;
    G1Jac acc = g1_identity();
    for (uint32_t t = lo + threadIdx.x; t < hi; t += WIN_THREADS) acc = g1_add(acc, load_jac(partial + (size_t)t * PT_WORDS));
    const G1Jac r = block_sum_points(tree, acc);   // ends with a barrier: every read of the slice is done
    if (threadIdx.x == 0) store_jac(partial + (size_t)lo * PT_WORDS, r);
    __syncthreads();
  }
}

// one workgroup per hot bucket: all of its partials, or the slice sums the first level left behind
__global__ __launch_bounds__(WIN_THREADS) void msm_bucket_finalize_big_kernel(const uint32_t* __restrict__ partial,
                                                                              const uint32_t* __restrict__ toff,
                                                                              uint32_t* __restrict__ bucket,
                                                                              const uint32_t* __restrict__ big_count,
                                                                              const uint32_t* __restrict__ big_list) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  const uint32_t count = *big_count;
  for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    const uint32_t b = big_list[item];
    const uint32_t lo = toff[b], hi = toff[b + 1];
    const uint32_t step = hi - lo > FINALIZE_SLICE ? FINALIZE_SLICE : 1u;
    G1Jac acc = g1_identity();
    for (uint32_t t = lo + threadIdx.x * step; t < hi; t += WIN_THREADS * step)
      acc = g1_add(acc, load_jac(partial + (size_t)t * PT_WORDS));
    const G1Jac r = block_sum_points(tree, acc);
    if (threadIdx.x == 0) store_jac(bucket + (size_t)b * PT_WORDS, r);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// K4: per-window sum_b b * B_b
// ---------------------------------------------------------------------------------------------
// K4a: one lane per segment of SEG consecutive buckets [lo, lo+SEG): running sums give
// sum (b - lo + 1) B_b; the remaining (lo - 1) * (sum B_b) is a short double-and-add.
__global__ __launch_bounds__(ACC_THREADS) void msm_reduce_segments_kernel(const uint32_t* __restrict__ bucket,
                                                                          uint32_t* __restrict__ segres, uint32_t W,
                                                                          uint32_t NB, uint32_t NBP, uint32_t SEG,
                                                                          uint32_t nseg) {
  const uint32_t idx = blockIdx.x * ACC_THREADS + threadIdx.x;
  if (idx >= W * nseg) return;
  const uint32_t w = idx / nseg, sgi = idx - w * nseg;
  const uint32_t lo = sgi * SEG + 1;
  const uint32_t hi = lo + SEG - 1 < NB ? lo + SEG - 1 : NB;
  const uint32_t* bw = bucket + (size_t)w * NBP * PT_WORDS;
  // one wave per SIMD runs this kernel (the work is what costs, so SEG is long): nothing hides a bucket's load but the two
  // additions of the bucket before it -- the next record is requested before they start
  G1Jac run = g1_identity(), acc = g1_identity();
  G1Jac cur = hi >= lo ? load_jac(bw + (size_t)hi * PT_WORDS) : g1_identity();
  for (uint32_t b = hi; b >= lo; --b) {
    const G1Jac nxt = b > lo ? load_jac(bw + (size_t)(b - 1) * PT_WORDS) : g1_identity();
    run = g1_add(run, cur);
    acc = g1_add(acc, run);
    cur = nxt;
  }
  uint32_t m = lo - 1;  // < NB
  if (m != 0 && !run.inf) {
    G1Jac r = g1_identity();
    for (int bit = 31 - __clz(m); bit >= 0; --bit) {
      r = g1_double(r);
      if ((m >> bit) & 1) r = g1_add(r, run);
    }
    acc = g1_add(acc, r);
  }
  store_jac(segres + (size_t)idx * PT_WORDS, acc);
}

// K4b: sum `count` points per window down to ceil(count / SUM_SPAN) (one workgroup per span: strided
// serial sums then an LDS tree); applied until one point per window is left.
#ifndef HM_SUM_PER_LANE
#define HM_SUM_PER_LANE 4     // measured: 4 beats 16 by 0.06-0.07 ms at every size (more workgroups, shorter serial sums)
#endif
constexpr uint32_t SUM_SPAN = WIN_THREADS * HM_SUM_PER_LANE;
__global__ __launch_bounds__(WIN_THREADS) void msm_sum_points_kernel(const uint32_t* __restrict__ in, uint32_t count,
                                                                     uint32_t* __restrict__ out, uint32_t out_count) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  const uint32_t w = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const uint32_t* sw = in + (size_t)w * count * PT_WORDS;
  const uint32_t lo = blk * SUM_SPAN, hi = lo + SUM_SPAN < count ? lo + SUM_SPAN : count;
  G1Jac acc = g1_identity();
  for (uint32_t s = lo + t; s < hi; s += WIN_THREADS) acc = g1_add(acc, load_jac(sw + (size_t)s * PT_WORDS));
  const G1Jac r = block_sum_points(tree, acc);
  if (t == 0) store_jac(out + ((size_t)w * out_count + blk) * PT_WORDS, r);
}

// K4 for a SHARED bucket set (one window, up to 2^21 buckets), two launches deep instead of a running-sum chain.
// With j = b - 1 = hi * M + lo (M * H = NB, both powers of two):
//     sum_b b B_b = sum_lo lo * C_lo + sum_hi (M * hi + 1) * R_hi,     C_lo = sum_hi B, R_hi = sum_lo B
// so every bucket enters two PLAIN sums (a column and a row: any order, lanes then an LDS tree), the same two additions
// per bucket the running sums cost, without their serial chain of 2 * SEG additions and the log2(NB)-bit multiple behind
// it.  The H + M sums are then split by weight BIT: S_p = sum of the sums whose weight has bit p, one workgroup per
// bit, and the host fold, which doubles its way down 255 bits anyway, reads the c - 1 records S_p as windows of one bit.
// Launch 1: workgroup `blk` < H sums row hi = blk, the others column lo = blk - H.
// blockIdx.y = the bucket set (a group of MSMs over one table: one set each, NB + 1 bucket records and H + M sums apart)
__global__ __launch_bounds__(WIN_THREADS) void msm_reduce_rowcol_kernel(const uint32_t* __restrict__ bucket, uint32_t* __restrict__ sums,
                                                                        uint32_t m, uint32_t h) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  const uint32_t M = 1u << m, H = 1u << h, blk = blockIdx.x, t = threadIdx.x;
  bucket += (size_t)blockIdx.y * ((size_t)M * H + 1) * PT_WORDS;
  sums += (size_t)blockIdx.y * ((size_t)M + H) * PT_WORDS;
  const bool row = blk < H;
  const uint32_t count = row ? M : H;                                  // elements of this sum
  const uint32_t first = row ? blk * M : blk - H, stride = row ? 1u : M;
  const uint32_t* b1 = bucket + PT_WORDS;                              // bucket b = j + 1
  G1Jac acc = g1_identity();
  if (t < count) {
    G1Jac cur = load_jac(b1 + (size_t)(first + t * stride) * PT_WORDS);
    for (uint32_t e = t; e < count; e += WIN_THREADS) {                // the next record is requested before the addition starts
      const uint32_t en = e + WIN_THREADS;
      const G1Jac nxt = en < count ? load_jac(b1 + (size_t)(first + en * stride) * PT_WORDS) : g1_identity();
      acc = g1_add(acc, cur);
      cur = nxt;
    }
  }
  const G1Jac r = block_sum_points_upto(tree, acc, count < WIN_THREADS ? count : WIN_THREADS);
  if (t == 0) store_jac(sums + (size_t)blk * PT_WORDS, r);
}

// Launch 2: workgroup p sums the rows / columns whose weight has bit p and writes the record the host fold reads.
// Columns: weight lo (bits 0 .. m - 1).  Rows: weight M * hi + 1 (bit 0, and bits m .. m + h - 1).
// blockIdx.y = the bucket set; its records go res_stride words behind the previous set's, each block led by a copy of the
// chain's four totals words (what the host reads per element: msm_finish_wait_fold)
__global__ __launch_bounds__(WIN_THREADS) void msm_reduce_bits_kernel(const uint32_t* __restrict__ sums, uint32_t m, uint32_t h,
                                                                      uint32_t* __restrict__ winres, uint32_t res_stride) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  const uint32_t M = 1u << m, H = 1u << h, p = blockIdx.x, t = threadIdx.x;
  sums += (size_t)blockIdx.y * ((size_t)M + H) * PT_WORDS;
  if (blockIdx.y != 0 && p == 0 && t < 4) (winres - 4)[(size_t)blockIdx.y * res_stride + t] = (winres - 4)[t];
  winres += (size_t)blockIdx.y * res_stride;
  const uint32_t* rows = sums;
  const uint32_t* cols = sums + (size_t)H * PT_WORDS;
  G1Jac acc = g1_identity();
  uint32_t widest = 0;                                                 // lanes holding anything: the tree's width
  if (p < m) {                                                         // the M / 2 columns with bit p of lo set
    const uint32_t half = M >> 1, low = (1u << p) - 1u;
    for (uint32_t i = t; i < half; i += WIN_THREADS) {
      const uint32_t lo = ((i & ~low) << 1) | (1u << p) | (i & low);
      acc = g1_add(acc, load_jac(cols + (size_t)lo * PT_WORDS));
    }
    widest = half;
    if (p == 0) {                                                      // every row: the + 1 of its weight
      for (uint32_t hi = t; hi < H; hi += WIN_THREADS) acc = g1_add(acc, load_jac(rows + (size_t)hi * PT_WORDS));
      if (H > widest) widest = H;
    }
  } else {                                                             // the H / 2 rows with bit p - m of hi set
    const uint32_t q = p - m, half = H >> 1, low = (1u << q) - 1u;
    for (uint32_t i = t; i < half; i += WIN_THREADS) {
      const uint32_t hi = ((i & ~low) << 1) | (1u << q) | (i & low);
      acc = g1_add(acc, load_jac(rows + (size_t)hi * PT_WORDS));
    }
    widest = half;
  }
  if (widest > WIN_THREADS) widest = WIN_THREADS;
  if (widest < 1) widest = 1;
  const G1Jac r = block_sum_points_upto(tree, acc, widest);
  if (t == 0) store_window_ext(winres + (size_t)p * 32, r);
}

// window sums -> the external Jacobian format (12 x u64 + flag word) the host fold reads
__global__ void msm_windows_to_ext_kernel(const uint32_t* __restrict__ in, uint32_t W, uint32_t* __restrict__ winres) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  store_window_ext(winres + (size_t)w * 32, load_jac(in + (size_t)w * PT_WORDS));
}

// ---------------------------------------------------------------------------------------------
// fixed-base scalar multiplication (ParamsKZG::setup's G1 work; also builds the bench SRS)
// out[i] = [k_i] B, affine external.  One lane per scalar: 4-bit fixed windows over a 64 x 15
// table of multiples of B built by the first kernel; final inversion by Fermat per lane.
// ---------------------------------------------------------------------------------------------
struct G1AffineWords {     // one external affine point, passed to the table kernel by value
  uint32_t w[16];
};
__global__ void g1_fixed_table_kernel(G1AffineWords base_ext, uint32_t* __restrict__ table) {
  // table[w][d-1] = [d * 16^w] B as Jacobian records, w < 64, d in 1..15.  One lane per window.
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 64) return;
  uint32_t wx[8], wy[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { wx[k] = base_ext.w[k]; wy[k] = base_ext.w[8 + k]; }
  G1Aff b;
  b.x = fe_from_ext<FqParams>(wx);
  b.y = fe_from_ext<FqParams>(wy);
  G1Jac p = g1_from_affine(b);
  for (uint32_t k = 0; k < 4 * w; ++k) p = g1_double(p);
  G1Jac acc = p;
  for (uint32_t d = 1; d <= 15; ++d) {
    store_jac(table + ((size_t)w * 15 + (d - 1)) * PT_WORDS, acc);
    acc = g1_add(acc, p);
  }
}

__device__ __forceinline__ Fq fq_inverse(const Fq& a) {  // a^(p-2), a a product output
  // exponent p - 2 in 29-bit limbs
  uint32_t e[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) e[i] = FqParams::MOD[i];
  e[0] -= 2;  // MOD[0] is odd and > 2: no borrow
  Fq acc = fe_one<FqParams>();
  for (int i = 8; i >= 0; --i) {
    const int top = i == 8 ? 24 : 28;
    for (int bit = top; bit >= 0; --bit) {
      acc = fe_sqr(acc);
      if ((e[i] >> bit) & 1) acc = fe_mul(acc, a);
    }
  }
  return acc;
}

// Precomputed multiples for a fixed base set: table[j*n + i] = 2^(c*j) * P_i (affine, internal
// packed), j < W.  With them every window's digit indexes its own copy of the point and ALL windows
// share one set of 2^(c-1) buckets, so the window size is no longer capped by W bucket sets (c = 22,
// W = 12 at 2^24 instead of c = 16, W = 16: a quarter fewer mixed additions) and the final Horner
// disappears.  One lane per point: c doublings, then a Fermat inversion per stored multiple.
__global__ __launch_bounds__(ACC_THREADS) void msm_precompute_kernel(uint32_t* __restrict__ table,
                                                                     const uint8_t* __restrict__ inf, size_t n, uint32_t c,
                                                                     uint32_t W, uint32_t n_wide) {
  const size_t i = (size_t)blockIdx.x * ACC_THREADS + threadIdx.x;
  if (i >= n) return;
  if (inf[i]) {
    for (uint32_t j = 1; j < W; ++j) {
      uint4* o = reinterpret_cast<uint4*>(table + ((size_t)j * n + i) * 16);
      o[0] = o[1] = o[2] = o[3] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  G1Jac p = g1_from_affine(load_base(table, (uint32_t)i));
  for (uint32_t j = 1; j < W; ++j) {
    const uint32_t bits = j < n_wide ? c : c - 1;           // multiple j = 2^(offset of window j) P, HALF of it for a narrow window (see msm_digits_kernel)
    for (uint32_t k = 0; k < bits; ++k) p = g1_double_nz(p);   // a point of odd prime order never doubles to the identity
    const Fq zi = fq_inverse(p.z);
    const Fq zi2 = fe_sqr(zi);
    const Fq zi3 = fe_mul(zi2, zi);
    uint32_t ox[8], oy[8];
    fe_pack(ox, fe_mul(p.x, zi2));
    fe_pack(oy, fe_mul(p.y, zi3));
    uint4* o = reinterpret_cast<uint4*>(table + ((size_t)j * n + i) * 16);
    o[0] = make_uint4(ox[0], ox[1], ox[2], ox[3]);
    o[1] = make_uint4(ox[4], ox[5], ox[6], ox[7]);
    o[2] = make_uint4(oy[0], oy[1], oy[2], oy[3]);
    o[3] = make_uint4(oy[4], oy[5], oy[6], oy[7]);
  }
}

// The same table in two passes with ONE inversion per point instead of one per stored multiple (Montgomery's trick
// over the W - 1 Z coordinates of a point's chain): a Fermat inversion costs ~380 products, the c doublings between
// two multiples ~130, so the single-pass kernel above spends three quarters of its time inverting (674 ms for 2^24
// points, W = 12).  Pass A runs the doubling chain and leaves every multiple in JACOBIAN form -- (X, Y) packed in its
// table slot, Z and the running product Z_1 ... Z_j in a scratch array; pass B inverts the last running product and
// walks back, turning every slot into the affine point.
__device__ __forceinline__ void store_packed(uint32_t* dst, const Fq& v) {
  uint32_t w[8];
  fe_pack(w, v);
  uint4* o = reinterpret_cast<uint4*>(dst);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ Fq load_packed(const uint32_t* src) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_unpack<FqParams>(w);
}
// ztmp: [2][W - 1][n][8 words]: half 0 the Z coordinates, half 1 their running products
// Both passes work on a SLAB of m points starting at `first` (the scratch is per slab: [2][W - 1][m][8 words]).
__global__ __launch_bounds__(ACC_THREADS) void msm_precompute_chain_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ ztmp,
                                                                           const uint8_t* __restrict__ inf, size_t n, uint32_t c,
                                                                           uint32_t W, uint32_t n_wide, size_t first, size_t m) {
  const size_t r = (size_t)blockIdx.x * ACC_THREADS + threadIdx.x, i = first + r;
  if (r >= m || i >= n || inf[i]) return;           // identity bases: pass B zeroes their slots
  G1Jac p = g1_from_affine(load_base(table, (uint32_t)i));
  Fq run = fe_one<FqParams>();
  for (uint32_t j = 1; j < W; ++j) {
    const uint32_t bits = j < n_wide ? c : c - 1;           // as in msm_precompute_kernel
    for (uint32_t k = 0; k < bits; ++k) p = g1_double_nz(p);   // a point of odd prime order never doubles to the identity
    uint32_t* slot = table + ((size_t)j * n + i) * 16;
    store_packed(slot, p.x);
    store_packed(slot + 8, p.y);
    run = j == 1 ? p.z : fe_mul(run, p.z);
    store_packed(ztmp + ((size_t)(j - 1) * m + r) * 8, p.z);
    store_packed(ztmp + ((size_t)(W - 1 + j - 1) * m + r) * 8, run);
  }
}
__global__ __launch_bounds__(ACC_THREADS) void msm_precompute_normalise_kernel(uint32_t* __restrict__ table,
                                                                               const uint32_t* __restrict__ ztmp,
                                                                               const uint8_t* __restrict__ inf, size_t n, uint32_t W,
                                                                               size_t first, size_t m) {
  const size_t r = (size_t)blockIdx.x * ACC_THREADS + threadIdx.x, i = first + r;
  if (r >= m || i >= n) return;
  if (inf[i]) {
    for (uint32_t j = 1; j < W; ++j) {
      uint4* o = reinterpret_cast<uint4*>(table + ((size_t)j * n + i) * 16);
      o[0] = o[1] = o[2] = o[3] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  Fq inv = fq_inverse(load_packed(ztmp + ((size_t)(W - 1 + W - 2) * m + r) * 8));      // 1 / (Z_1 ... Z_{W-1})
  for (uint32_t j = W - 1; j >= 1; --j) {
    const Fq zj = load_packed(ztmp + ((size_t)(j - 1) * m + r) * 8);
    const Fq zi = j > 1 ? fe_mul(inv, load_packed(ztmp + ((size_t)(W - 1 + j - 2) * m + r) * 8)) : inv;   // 1 / Z_j
    inv = fe_mul(inv, zj);                                                                // 1 / (Z_1 ... Z_{j-1})
    const Fq zi2 = fe_sqr(zi);
    const Fq zi3 = fe_mul(zi2, zi);
    uint32_t* slot = table + ((size_t)j * n + i) * 16;
    const Fq ax = fe_mul(load_packed(slot), zi2), ay = fe_mul(load_packed(slot + 8), zi3);
    store_packed(slot, ax);
    store_packed(slot + 8, ay);
  }
}

__global__ __launch_bounds__(ACC_THREADS) void g1_fixed_base_mul_kernel(const uint32_t* __restrict__ scalars,
                                                                        const uint32_t* __restrict__ table,
                                                                        uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * ACC_THREADS + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(scalars + i * 8);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w_in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Fr k32 = fe_zero<FrParams>();
  k32.l[0] = 32;
  uint32_t v[8];
  fe_pack(v, fe_canonical(fe_mul(fe_unpack<FrParams>(w_in), k32)));
  G1Jac acc = g1_identity();
  for (uint32_t w = 0; w < 64; ++w) {
    const uint32_t d = (v[w >> 3] >> ((w & 7) * 4)) & 15u;
    if (d != 0) acc = g1_add(acc, load_jac(table + ((size_t)w * 15 + (d - 1)) * PT_WORDS));
  }
  uint32_t ox[8], oy[8];
  if (acc.inf) {
#pragma unroll
    for (int k = 0; k < 8; ++k) ox[k] = oy[k] = 0;
  } else {
    const Fq zi = fq_inverse(acc.z);
    const Fq zi2 = fe_sqr(zi);
    const Fq zi3 = fe_mul(zi2, zi);
    fe_to_ext(ox, fe_mul(acc.x, zi2));
    fe_to_ext(oy, fe_mul(acc.y, zi3));
  }
  uint4* o = reinterpret_cast<uint4*>(out + i * 16);
  o[0] = make_uint4(ox[0], ox[1], ox[2], ox[3]);
  o[1] = make_uint4(ox[4], ox[5], ox[6], ox[7]);
  o[2] = make_uint4(oy[0], oy[1], oy[2], oy[3]);
  o[3] = make_uint4(oy[4], oy[5], oy[6], oy[7]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int msm_convert_bases(const uint32_t* d_bases_ext, uint32_t* d_xy, uint8_t* d_inf, size_t n, hipStream_t stream) {
  if (n == 0) return HM_OK;
  hipLaunchKernelGGL(msm_convert_bases_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_bases_ext, d_xy,
                     d_inf, n);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

static uint32_t ilog2(size_t n) {
  uint32_t l = 0;
  while ((n >> (l + 1)) != 0) ++l;
  return l;
}

// host-side Horner over the window sums and affine normalisation (host_fq.h: native 4 x u64
// arithmetic on the external-format words the reduction kernel wrote)
// result = sum_w 2^(offset of window w) * S_w: Horner from the top window, shifting by the width of window w
// before its sum is added (widths == nullptr: every window is c bits wide)
static void host_fold(const uint32_t* winres, uint32_t W, uint32_t c, uint64_t out_jac_ext[12], int* out_is_identity,
                      const uint8_t* widths = nullptr) {
  host::G1J acc = host::g1_identity();
  for (int w = (int)W - 1; w >= 0; --w) {
    const uint32_t shift = widths ? widths[w] : c;
    for (uint32_t k = 0; k < shift; ++k) acc = host::g1_double(acc);
    const uint32_t* o = winres + (size_t)w * 32;
    if (o[24] == 0) {
      host::G1J p;
      std::memcpy(p.x.l, o, 32);
      std::memcpy(p.y.l, o + 8, 32);
      std::memcpy(p.z.l, o + 16, 32);
      acc = host::g1_add(acc, p);
    }
  }
  host::g1_normalise(acc, out_jac_ext, out_is_identity);
}

// sum of `count` external Jacobian points (12 u64 each, z = 0 for the identity) on the host:
// the fold of per-GPU partial MSM results after the RCCL all-gather (a few points, latency only)
void host_sum_points(const uint64_t* pts, size_t count, uint64_t out_jac_ext[12], int* out_is_identity) {
  std::vector<uint32_t> win(count * 32, 0);
  for (size_t i = 0; i < count; ++i) {
    const uint64_t* p = pts + i * 12;
    std::memcpy(&win[i * 32], p, 96);
    win[i * 32 + 24] = (p[8] | p[9] | p[10] | p[11]) == 0 ? 1u : 0u;
  }
  host_fold(win.data(), (uint32_t)count, 0, out_jac_ext, out_is_identity);
}

static std::atomic<int> g_window_override{0};   // process-wide knob, read once per MSM
// Per-phase events (digits / sort / accumulate / reduce, and the accumulate kernel alone) cost seven event records per
// MSM -- nothing against a 20 ms MSM, a fifth of the host time of a prover-sized one.  -1: the general pipeline records
// them, the five-launch plan does not; 0 / 1: never / always (hm_msm_set_phase_timing).
static std::atomic<int> g_phase_timing{-1};
void msm_set_phase_timing(int mode) { g_phase_timing.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); }
bool msm_phase_timing(bool small_plan) {
  const int m = g_phase_timing.load(std::memory_order_relaxed);
  return m < 0 ? !small_plan : m != 0;
}
void msm_set_window_override(int c) { g_window_override.store(c, std::memory_order_relaxed); }

// Window size for a precomputed (single bucket set) base set of n points: minimise
// n * W(c) mixed additions + ~3 * 2^(c-1) addition-equivalents of bucket reduction.
// The 255 digit bits over W windows as evenly as possible: windows [0, n_wide) are `wide` bits, the others wide - 1.
// (W - 1 full windows and a short top one would send every point's top digit into the lowest 2^(short - 1) buckets:
// at 2^24 points, c = 22, 4 096 buckets with 4 192 points each beside a mean of 96.)
static void balanced_windows(uint32_t W, uint32_t* wide, uint32_t* n_wide) {
  const uint32_t base = 255 / W, rest = 255 - base * W;
  *wide = rest ? base + 1 : base;
  *n_wide = rest ? rest : W;
}

uint32_t msm_precomp_window(size_t n) {
  uint32_t best = 8;
  double best_cost = 1e300;
  // (capped at 22: the model knows additions only.  Measured at 2^25 / 2^26, c = 24 against 22: K3 26.6 / 51.4 against
  // 28.6 / 56.0 ms, but the sort 8.8 / 16.6 against 6.0 / 12.0 -- 2^23 buckets need 4 096 coarse bins, two items per bin
  // and first-level tile -- and the bucket reduction 2.8 against 1.3: whole MSM 39.2 / 72.2 against 36.5 / 70.5 ms)
  for (uint32_t c = 8; c <= 22; ++c) {
    const double W = (double)((255 + c - 1) / c);
    const double cost = (double)n * W + 3.0 * (double)(1ull << (c - 1));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  static const int forced = [] { const char* v = std::getenv("HALO2_MI355X_PRECOMP_C"); return v && *v ? std::atoi(v) : 0; }();   // experiments
  if (forced >= 8 && forced <= 24) best = (uint32_t)forced;
  uint32_t wide, n_wide;
  balanced_windows((255 + best - 1) / best, &wide, &n_wide);     // the widest window of the balanced split (<= best)
  return wide;
}

int msm_precompute(uint32_t* d_table, const uint8_t* d_inf, size_t n, uint32_t c, uint32_t W, hipStream_t stream) {
  if (n == 0 || W <= 1) return HM_OK;
  uint32_t wide, n_wide;
  balanced_windows(W, &wide, &n_wide);
  if (wide != c) return hm_fail(HM_ERR_INTERNAL, "msm: precomputation window is not the balanced width");
  const dim3 grid((uint32_t)((n + ACC_THREADS - 1) / ACC_THREADS)), block(ACC_THREADS);
  // two passes with one inversion per point (scratch: 64 B per stored multiple, freed before returning); the one-pass
  // kernel when that scratch cannot be had
  uint32_t* d_z = nullptr;
  static const bool one_pass = [] { const char* v = std::getenv("HALO2_MI355X_PRECOMP_ONE_PASS"); return v && *v == '1'; }();
  const size_t slab = n < ((size_t)1 << 20) ? n : ((size_t)1 << 20);      // <= 1 GiB of scratch whatever the set's size
  if (!one_pass && hipMalloc((void**)&d_z, (size_t)2 * (W - 1) * slab * 32) == hipSuccess) {
    const dim3 sgrid((uint32_t)((slab + ACC_THREADS - 1) / ACC_THREADS));
    for (size_t first = 0; first < n; first += slab) {
      hipLaunchKernelGGL(msm_precompute_chain_kernel, sgrid, block, 0, stream, d_table, d_z, d_inf, n, c, W, n_wide, first, slab);
      hipLaunchKernelGGL(msm_precompute_normalise_kernel, sgrid, block, 0, stream, d_table, (const uint32_t*)d_z, d_inf, n, W, first, slab);
    }
    const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(stream);
    (void)hipFree(d_z);
    if (e1 != hipSuccess || e2 != hipSuccess) return hm_fail(HM_ERR_HIP, "msm: precomputation failed");
    return HM_OK;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(msm_precompute_kernel, grid, block, 0, stream, d_table, d_inf, n, c, W, n_wide);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

// HALO2_MI355X_SORT_V1=1: the round-1..3 tiled scatters (one 4-byte load per item) instead of the vector-load form (A/B)
static bool sort_v1() {
  static const bool v = [] { const char* e = std::getenv("HALO2_MI355X_SORT_V1"); return e && *e == '1'; }();
  return v;
}

struct BigRegionPlan {      // cooperative handling of sort regions that hold far more than their share
  uint32_t big = 0;         // regions above this many items are split
  uint32_t slice = 0;       // ... into slices of this many items, one workgroup each
  uint32_t capacity = 0;    // entries of the work list
  uint2* list = nullptr;
  uint32_t* count = nullptr;
  uint32_t* gcursor = nullptr;   // copy of boff: global per-bucket cursors of the cooperative scatter
};

template <class ITEM>
static int launch_sort(const int32_t* d_digits, uint32_t* d_chist, uint32_t* d_ctot, uint32_t* d_cstart, void* d_tmp,
                       uint32_t* d_bcnt, size_t sn, size_t chunk, uint32_t G, uint32_t SW, uint32_t fb, uint32_t ib, uint32_t cb,
                       uint32_t NC, uint32_t NBP, const BigRegionPlan& br, hipStream_t stream) {
  const size_t lds_fine = (size_t)4 << fb;
  if (cb) {
    const size_t lds_coarse = (size_t)NC * 4;
    hipLaunchKernelGGL(msm_part1_hist_kernel, dim3(G, SW), dim3(SORT_THREADS), lds_coarse, stream, d_digits, d_chist, sn, chunk,
                       fb, NC);
    hipLaunchKernelGGL(msm_part1_scan_kernel, dim3(SW * NC), dim3(64), 0, stream, d_chist, d_ctot, G, NC, SW);
    hipLaunchKernelGGL(msm_part1_starts_kernel, dim3(1), dim3(1024), 0, stream, (const uint32_t*)d_ctot, d_cstart, SW * NC);
    static const bool big_tiles = [] { const char* v = std::getenv("HALO2_MI355X_P1_BIG_TILES"); return !(v && *v == '0'); }();
    static const uint32_t p1_big_from = [] { const char* v = std::getenv("HALO2_MI355X_P1_BIG_FROM"); return (uint32_t)(v && *v ? std::atoi(v) : 512); }();
    if (sizeof(ITEM) == 4 && NC <= 4096 && !sort_v1()) {      // the vector-load form (32-bit items)
      if (NC >= (p1_big_from) && NC <= 2048 && big_tiles) {
        constexpr int TILE = SORT_THREADS * P1_BIG_IPT;
        hipLaunchKernelGGL((msm_part1_scatter_v2_kernel<P1_BIG_IPT>), dim3(G, SW), dim3(SORT_THREADS), ((size_t)3 * NC + 32 + 2 * TILE) * 4, stream,
                           d_digits, (const uint32_t*)d_chist, (const uint32_t*)d_cstart, (uint32_t*)d_tmp, sn, chunk, fb, ib, NC);
      } else {
        constexpr int TILE = SORT_THREADS * P1_IPT_DEFAULT;
        hipLaunchKernelGGL((msm_part1_scatter_v2_kernel<P1_IPT_DEFAULT>), dim3(G, SW), dim3(SORT_THREADS), ((size_t)3 * NC + 32 + 2 * TILE) * 4,
                           stream, d_digits, (const uint32_t*)d_chist, (const uint32_t*)d_cstart, (uint32_t*)d_tmp, sn, chunk, fb, ib, NC);
      }
    } else if (NC >= (p1_big_from) && NC <= 2048 && sizeof(ITEM) == 4 && big_tiles) {
      constexpr int TILE = SORT_THREADS * P1_BIG_IPT;
      const size_t lds_p1 = ((size_t)3 * NC + 32 + TILE) * 4 + (size_t)TILE * sizeof(ITEM);
      hipLaunchKernelGGL((msm_part1_scatter_tiled_kernel<ITEM, P1_BIG_IPT>), dim3(G, SW), dim3(SORT_THREADS), lds_p1, stream, d_digits,
                         (const uint32_t*)d_chist, (const uint32_t*)d_cstart, (ITEM*)d_tmp, sn, chunk, fb, ib, NC);
    } else if (NC <= 4096) {
      constexpr int TILE = SORT_THREADS * P1_IPT_DEFAULT;
      const size_t lds_p1 = ((size_t)3 * NC + 32 + TILE) * 4 + (size_t)TILE * sizeof(ITEM);
      hipLaunchKernelGGL((msm_part1_scatter_tiled_kernel<ITEM, P1_IPT_DEFAULT>), dim3(G, SW), dim3(SORT_THREADS), lds_p1, stream, d_digits,
                         (const uint32_t*)d_chist, (const uint32_t*)d_cstart, (ITEM*)d_tmp, sn, chunk, fb, ib, NC);
    } else {
      hipLaunchKernelGGL(msm_part1_scatter_kernel<ITEM>, dim3(G, SW), dim3(SORT_THREADS), lds_coarse, stream, d_digits,
                         (const uint32_t*)d_chist, (const uint32_t*)d_cstart, (ITEM*)d_tmp, sn, chunk, fb, ib, NC);
    }
    // regions far above their share go to the cooperative kernels: list them, then count in both forms
    hipLaunchKernelGGL(msm_big_regions_kernel, dim3((SW * NC + 255) / 256), dim3(256), 0, stream, (const uint32_t*)d_cstart,
                       SW * NC, br.big, br.slice, br.list, br.count, br.capacity);
    hipLaunchKernelGGL((msm_part2_hist_kernel<false, ITEM, false>), dim3(NC, SW), dim3(SORT_THREADS), lds_fine, stream,
                       (const ITEM*)d_tmp, d_digits, (const uint32_t*)d_cstart, d_bcnt, sn, fb, ib, NC, NBP, br.big, br.slice,
                       (const uint2*)br.list, (const uint32_t*)br.count);
    hipLaunchKernelGGL((msm_part2_hist_kernel<false, ITEM, true>), dim3(br.capacity), dim3(SORT_THREADS), lds_fine, stream,
                       (const ITEM*)d_tmp, d_digits, (const uint32_t*)d_cstart, d_bcnt, sn, fb, ib, NC, NBP, br.big, br.slice,
                       (const uint2*)br.list, (const uint32_t*)br.count);
  } else {
    hipLaunchKernelGGL((msm_part2_hist_kernel<true, ITEM, false>), dim3(1, SW), dim3(SORT_THREADS), lds_fine, stream,
                       (const ITEM*)d_tmp, d_digits, (const uint32_t*)d_cstart, d_bcnt, sn, fb, ib, NC, NBP, 0xffffffffu, 0u,
                       (const uint2*)nullptr, (const uint32_t*)nullptr);
  }
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

template <class ITEM>
static int launch_sort_scatter(const int32_t* d_digits, const uint32_t* d_cstart, const void* d_tmp, const uint32_t* d_boff,
                               uint32_t* d_sorted, size_t sn, uint32_t SW, uint32_t fb, uint32_t ib, uint32_t cb, uint32_t NC,
                               uint32_t NBP, uint32_t NBT, const BigRegionPlan& br, const Positional& ps, hipStream_t stream) {
  const size_t lds_fine = (size_t)4 << fb;
  if (cb && (fb <= 11 || (ps.nsc != 0 && fb <= 12)) && sizeof(ITEM) == 4 && !sort_v1()) {
    const size_t lds_tiled = ((size_t)3 * (1u << fb) + 32 + P2_TILE) * 4 + (size_t)P2_TILE * 2;
    hipLaunchKernelGGL((msm_part2_scatter_v2_kernel<false>), dim3(NC, SW), dim3(SORT_THREADS), lds_tiled, stream, (const uint32_t*)d_tmp,
                       d_cstart, d_boff, d_sorted, fb, ib, NC, NBP, br.big, br.slice, (const uint2*)br.list, (const uint32_t*)br.count,
                       br.gcursor, ps);
    hipLaunchKernelGGL((msm_part2_scatter_v2_kernel<true>), dim3(br.capacity), dim3(SORT_THREADS), lds_tiled, stream, (const uint32_t*)d_tmp,
                       d_cstart, d_boff, d_sorted, fb, ib, NC, NBP, br.big, br.slice, (const uint2*)br.list, (const uint32_t*)br.count,
                       br.gcursor, ps);
  } else if (cb && (fb <= 11 || (ps.nsc != 0 && fb <= 12))) {      // 2^12 fine counters only where the positional plan asks for them
    const size_t lds_tiled = ((size_t)3 * (1u << fb) + 32 + P2_TILE) * 4 + (size_t)P2_TILE * 2;
    hipLaunchKernelGGL((msm_part2_scatter_tiled_kernel<ITEM, false>), dim3(NC, SW), dim3(SORT_THREADS), lds_tiled, stream,
                       (const ITEM*)d_tmp, d_cstart, d_boff, d_sorted, fb, ib, NC, NBP, br.big, br.slice, (const uint2*)br.list,
                       (const uint32_t*)br.count, br.gcursor, ps);
    hipLaunchKernelGGL((msm_part2_scatter_tiled_kernel<ITEM, true>), dim3(br.capacity), dim3(SORT_THREADS), lds_tiled, stream,
                       (const ITEM*)d_tmp, d_cstart, d_boff, d_sorted, fb, ib, NC, NBP, br.big, br.slice, (const uint2*)br.list,
                       (const uint32_t*)br.count, br.gcursor, ps);
  } else if (cb) {
    hipLaunchKernelGGL((msm_part2_scatter_kernel<false, ITEM>), dim3(NC, SW), dim3(SORT_THREADS), lds_fine, stream,
                       (const ITEM*)d_tmp, d_digits, d_cstart, d_boff, d_sorted, sn, fb, ib, NC, NBP);
  } else {
    hipLaunchKernelGGL((msm_part2_scatter_kernel<true, ITEM>), dim3(1, SW), dim3(SORT_THREADS), lds_fine, stream,
                       (const ITEM*)d_tmp, d_digits, d_cstart, d_boff, d_sorted, sn, fb, ib, NC, NBP);
  }
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int msm_launch_digits(const uint32_t* d_scalars_ext, const uint8_t* d_inf, int32_t* d_digits, size_t n, uint32_t c, uint32_t W,
                      hipStream_t stream) {
  MsmGroupScalars one;
  for (auto& p : one.s) p = d_scalars_ext;
  hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, one, d_inf, d_digits, n, c, W, W);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int msm_slot_prepare(MsmSlot& sl) {
  if (!sl.ev_ready) {
    for (int i = 0; i < 7; ++i) HM_HIP_CHECK(hipEventCreate(&sl.ev[i]));
    HM_HIP_CHECK(hipHostMalloc((void**)&sl.h_land, (size_t)HM_MSM_GROUP * (128 * 32 + 4) * sizeof(uint32_t), hipHostMallocDefault));
    sl.ev_ready = true;
  }
  return HM_OK;
}

// Enqueue every kernel of one MSM on `stream` using workspace slot `slot`; nothing here waits for
// the device.  d_xy: n points (plain) or the precomputed table of precomp_W * n points
// (precomp_c != 0).  msm_finish() later waits for the slot, folds the window sums on the host and
// fills the statistics.
// `group` MSMs over the same points in ONE launch chain (group > 1: only on a fixed-base table set below the positional
// plan's sizes -- msm_table_group_applies -- where every element is one bucket set of the same sort and the same K3 launch).
static int msm_issue(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                     const uint8_t* d_inf, size_t n, uint32_t precomp_c, hipStream_t stream) {
  MsmSlot& sl = ctx.msm_slots[slot];
  const uint32_t* d_scalars_ext = d_scalars_list[0];
  if (n == 0) return HM_OK;
  if (n >= (1ull << 31)) return hm_fail(HM_ERR_BAD_ARG, "msm: n must be < 2^31");
  if (group == 0 || group > (uint32_t)HM_MSM_GROUP) return hm_fail(HM_ERR_INTERNAL, "msm: group size out of range");
  // ---- plan ---------------------------------------------------------------------------------
  const bool single_set = precomp_c != 0;     // all windows accumulate into ONE bucket set
  const int window_override = g_window_override.load(std::memory_order_relaxed);
  uint32_t c;
  if (single_set) {
    c = precomp_c;
  } else if (window_override > 0) {
    c = (uint32_t)window_override;
  } else {
    // Rule of thumb: log2(n) - 3 (mean bucket load ~16: short chains, almost no bucket splitting, few
    // (point, bucket) pairs), capped at c = 17, whose W = 15 windows cover the 255 digit bits exactly
    // (18, 19 and 21 leave a top window of 3-8 bits whose few buckets receive all n points, and c = 20's
    // 6.8 M buckets cost more to reduce than they save).  The table is the measured optimum per size
    // (tools/msm_sweep.py <k> <c,c,...>): between 2^15 and 2^17 the fixed costs of the sort and of the
    // bucket reduction move it up to c = 15.
    static const int8_t kWindow[21] = {4, 4, 4, 4, 4, 4, 4, 4, 5, 6, 7, 8, 9, 10, 10, 13, 15, 15, 15, 16, 17};
    // the short chain of msm_small.hip (n < 2^19) has cheaper fixed costs per bucket set: its measured optimum is
    // log2(n) - 3 up to 2^16 (tools/small_tune.sh), then c = 15, whose 17 windows cover the 255 bits exactly
    static const int8_t kWindowSmall[19] = {4, 4, 4, 4, 4, 4, 4, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 15};
    const uint32_t lg = ilog2(n);
    int ci = lg <= 20 ? kWindow[lg] : 17;
    if (lg <= 18 && msm_small_applies(n, (uint32_t)kWindowSmall[lg], false)) ci = kWindowSmall[lg];
    c = (uint32_t)ci;
  }
  if (c < 2 || c > 24) return hm_fail(HM_ERR_BAD_ARG, "msm: window size out of range");
  if (msm_small_applies(n, c, single_set)) {
    if (group != 1) return hm_fail(HM_ERR_INTERNAL, "msm: a table group reached the five-launch plan");
    return msm_issue_small(ctx, slot, &d_scalars_ext, 1, d_xy, d_inf, n, c, stream);
  }
  if (group > 1 && !single_set) return hm_fail(HM_ERR_INTERNAL, "msm: a group needs the fixed-base table's shared bucket set");
  const uint32_t W = (255 + c - 1) / c;
  const uint32_t SW = single_set ? group : W;              // bucket sets ("sort windows"): one per MSM on a table, one per window else
  const size_t sn = single_set ? n * W : n;                // items per bucket set
  const uint32_t NB = 1u << (c - 1), NBP = NB + 1, NBT = SW * NBP;
  const uint64_t pairs_max = (uint64_t)n * W * group;      // of the whole chain
  if (pairs_max >= (1ull << 31)) return hm_fail(HM_ERR_BAD_ARG, "msm: n * windows must be < 2^31");
  double mean = (double)sn / (double)NB;
  if (single_set) {
    // narrow windows fill the even buckets only (msm_digits_kernel): size the tasks for THOSE buckets' load, so that a
    // bucket stays one task (at c = 22, W = 12: 168 points in an even bucket, 24 in an odd one, 96 on average)
    uint32_t wide_b, n_wide_b;
    balanced_windows(W, &wide_b, &n_wide_b);
    mean = (double)n * n_wide_b / (double)NB + (double)n * (W - n_wide_b) / ((double)NB / 2.0);
  }
#ifndef HM_L_MEAN_SCALE      // measurement knobs (tools/ab_build.sh); the defaults are the tuned values
#define HM_L_MEAN_SCALE 1.0
#endif
#ifndef HM_L_SIGMAS
#define HM_L_SIGMAS 4.0
#endif
  uint32_t L = (uint32_t)(HM_L_MEAN_SCALE * mean + HM_L_SIGMAS * std::sqrt(mean) + 8.0);
  {
    // small inputs: if one task per bucket would leave the chip (256 CUs x 4 SIMDs x ~5 waves x 64
    // lanes) mostly idle, cut the tasks shorter so that the launch still fills it (measured: a wave per
    // SIMD is not enough -- K3 at 2^16..2^18 is 15-40 % slower with one task per bucket)
    const double target_tasks = (double)TARGET_TASKS;
    if ((double)pairs_max / (double)L < target_tasks) L = (uint32_t)((double)pairs_max / target_tasks);
  }
  if (L < 16) L = 16;
  // chunks of the first sort level: >= 16 Ki items each, ~64 Ki at scale; G * SW workgroups should
  // cover the chip several times over (a shared bucket set has SW = 1, so it needs many chunks)
  uint32_t G = (uint32_t)((sn + 16383) / 16384);
  if (G < 1) G = 1;
  const uint32_t g_cap = SW >= 8 ? 256u : 4096u;
  if (G > g_cap) {
    G = (uint32_t)((sn + 65535) / 65536);
    if (G > g_cap) G = g_cap;
    if (G < 256) G = 256;
  }
  size_t chunk = (sn + G - 1) / G;
  // the device may shorten L (effective_task_len): then tasks <= 2 * TARGET_TASKS + one per bucket
  // (and never more than one per pair)
  const uint64_t T_max = std::min<uint64_t>(pairs_max, std::max<uint64_t>(pairs_max / L, 2ull * TARGET_TASKS) + NBT) + 1;
  // K4a: running-sum chain of 2*SEG additions per lane + a scalar multiple of <= log2(NB) bits
  // Segment length of K4a, measured per bucket count (the kernel is latency-bound: a lane's chain is
  // 2*SEG additions plus a log2(NB/SEG)-bit multiple, and the number of lanes decides how well the chip hides it)
#ifdef HM_SEG_SMALL
  uint32_t SEG = NB > (1u << 16) ? 32u : HM_SEG_SMALL;
#else
  uint32_t SEG = NB > (1u << 16) ? 32u : NB == (1u << 15) ? 16u : NB <= (1u << 13) ? 4u : 8u;
#endif
  if (SEG > NB) SEG = NB;
  const uint32_t nseg = (NB + SEG - 1) / SEG;
  // a shared bucket set: row / column sums and one record per weight bit instead (msm_reduce_rowcol_kernel)
  static const bool k4_chain_only = [] { const char* v = std::getenv("HALO2_MI355X_K4_CHAIN"); return v && *v == '1'; }();   // A/B
  const bool k4_2d = single_set && NB >= 16 && !k4_chain_only;
  const uint32_t k4_m = (c - 1 + 1) / 2, k4_h = (c - 1) - k4_m;        // M = 2^m columns, H = 2^h rows, M * H = NB
  if (group > 1 && !k4_2d) return hm_fail(HM_ERR_INTERNAL, "msm: a table group needs the two-launch bucket reduction");
  const uint32_t RW = k4_2d ? c - 1 : SW;                               // records the host fold reads (per bucket set of a table)
  const uint32_t res_stride = 4 + RW * 32;                              // words per element of a group: the totals, then its records
  // sort plan: item = [fine bucket bits | sign | item index]
  uint32_t ib = ilog2(sn) + ((sn & (sn - 1)) ? 1u : 0u);
  if (ib == 0) ib = 1;
  bool wide_items = false;
  Positional ps{nullptr, 0, 0, 0};
  uint32_t fb = c - 1 < 31 - ib ? c - 1 : 31 - ib;
  if (fb > 15) fb = 15;                  // 2^15 LDS counters = 128 KiB is what a workgroup may hold
#ifndef HM_NO_POSITIONAL
  if (single_set && c >= 14 && c <= 24 && (c - 1) - fb > 10) {
    // A shared bucket set has n W items (2^27.6 at 2^24 points) and 2^(c-1) buckets (2^21): fine bits + sign + the item index
    // do not fit one word with few enough coarse bins (<= 2^12) for the first level's LDS-sorted tiles.  POSITIONAL 32-bit
    // items: an item keeps the low `sbits` bits of its index, fb fine bucket bits and the sign; its super-chunk (the index
    // bits above sbits) follows from where it lies in its coarse region (msm_part2_scatter_tiled_kernel).  Chunks of the
    // first level are 2^16 (or more) consecutive indices, so every chunk lies inside one super-chunk.
    static const uint32_t cb_first = [] { const char* v = std::getenv("HALO2_MI355X_CB_FIRST"); return (uint32_t)(v && *v ? std::atoi(v) : 9); }();   // A/B: 9 against 10: sort 0.69 / 0.74 ms at 2^22 (c = 20), the same at c = 22; 8 is slower
    for (uint32_t cb_try = cb_first; cb_try <= 12 && ps.nsc == 0; ++cb_try) {
      if (c - 1 < cb_try + 5) break;
      const uint32_t fb_p = c - 1 - cb_try, sbits = 31 - fb_p;
      if (fb_p > 12 || sbits < 16) continue;                   // the tiled second level holds <= 2^12 fine counters
      size_t chunk_p = (size_t)1 << 16;
      while ((sn + chunk_p - 1) / chunk_p > 4096 && chunk_p < ((size_t)1 << sbits)) chunk_p <<= 1;
      const size_t nsc = (sn + ((size_t)1 << sbits) - 1) >> sbits;
      if (nsc > PS_MAX_SC || (sn + chunk_p - 1) / chunk_p > 4096) continue;
      ib = sbits;
      fb = fb_p;
      G = (uint32_t)((sn + chunk_p - 1) / chunk_p);
      chunk = chunk_p;
      ps.G = G;
      ps.gpc = (uint32_t)(((size_t)1 << sbits) / chunk_p);
      ps.nsc = (uint32_t)nsc;
    }
  }
#endif
  if (ps.nsc == 0 && fb < 5 && c - 1 > fb) {   // too few fine bits left in 32: switch to 64-bit items
    wide_items = true;
    fb = c - 1 < 9 ? c - 1 : 9;      // few fine runs per region: the region's write sectors must stay in L2
  }
  if (ps.nsc == 0 && sn >= (1u << 15) && (c - 1) - fb < 5) fb = c - 1 > 5 ? c - 1 - 5 : 0;   // >= 32 regions per set: enough workgroups
  const uint32_t cb = (c - 1) - fb, NC = 1u << cb;

  // ---- workspace ----------------------------------------------------------------------------
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += align(bytes); return o; };
  const size_t o_digits = carve((size_t)W * n * 4 * group);
  const size_t o_chist = carve((size_t)SW * G * NC * 4);
  const size_t o_ctot = carve((size_t)SW * NC * 4);
  const size_t o_cstart = carve(((size_t)SW * NC + 1) * 4);
  const size_t o_tmp = carve(cb ? pairs_max * (wide_items ? 8 : 4) : 4);
  const size_t o_bcnt = carve((size_t)NBT * 4);
  const size_t o_boff = carve((size_t)NBT * 4);
  const size_t o_toff = carve(((size_t)NBT + 1) * 4);
  const size_t o_bsum = carve(((size_t)NBT / SCAN_BLOCK + 2) * 8);
  const size_t o_sorted = carve(pairs_max * 4);
  const size_t o_tb = carve(T_max * 4);
  const size_t o_torder = carve(T_max * 4);
  const size_t o_tkeys = carve(2 * TASK_KEYS * 4);
  const size_t o_partial = carve(T_max * PT_WORDS * 4);
  const size_t o_bucket = carve((size_t)NBT * PT_WORDS * 4);
  const size_t o_seg = carve(k4_2d ? (((size_t)1 << k4_m) + ((size_t)1 << k4_h)) * PT_WORDS * 4 * SW : (size_t)SW * nseg * PT_WORDS * 4);
  const size_t o_seg2 = carve(((size_t)SW * (nseg / SUM_SPAN + 1)) * PT_WORDS * 4);
  const size_t o_res = carve((size_t)(single_set ? group : 1u) * res_stride * 4);   // per element: 4 totals words, then its records (one D2H copy)
  const size_t o_big = carve(((size_t)NBT + 4) * 4);
  const size_t o_slices = carve((T_max / FINALIZE_SLICE + T_max / (FINALIZE_SERIAL + 1) + 2) * 8);
  // cooperative sort of oversized regions (hot buckets): split anything above 4x the mean region
  BigRegionPlan br;
  {
    const uint64_t mean_region = pairs_max / ((uint64_t)NC * SW) + 1;
    uint64_t slice = (mean_region + P2_TILE - 1) / P2_TILE * P2_TILE;
    if (slice < 4 * (uint64_t)P2_TILE) slice = 4 * (uint64_t)P2_TILE;
    br.slice = (uint32_t)slice;
    br.big = (uint32_t)(4 * slice);
    br.capacity = (uint32_t)(pairs_max / slice + pairs_max / br.big + 2);
  }
  const bool coop_sort = cb != 0 && (fb <= 11 || (ps.nsc != 0 && fb <= 12));
  const size_t o_brlist = carve(coop_sort ? (size_t)br.capacity * 8 : 8);
  const size_t o_brcount = carve(16);
  const size_t o_gcur = carve(coop_sort ? (size_t)NBT * 4 : 4);
  uint8_t* ws = (uint8_t*)sl.ws.ensure(off);
  if (!ws) return hm_fail(HM_ERR_HIP, "msm: workspace allocation failed");
  int32_t* d_digits = (int32_t*)(ws + o_digits);
  uint32_t* d_chist = (uint32_t*)(ws + o_chist);
  uint32_t* d_ctot = (uint32_t*)(ws + o_ctot);
  uint32_t* d_cstart = (uint32_t*)(ws + o_cstart);
  void* d_tmp = (void*)(ws + o_tmp);
  uint32_t* d_bcnt = (uint32_t*)(ws + o_bcnt);
  uint32_t* d_boff = (uint32_t*)(ws + o_boff);
  uint32_t* d_toff = (uint32_t*)(ws + o_toff);
  uint32_t* d_bsum = (uint32_t*)(ws + o_bsum);
  uint32_t* d_sorted = (uint32_t*)(ws + o_sorted);
  uint32_t* d_tb = (uint32_t*)(ws + o_tb);
  uint32_t* d_torder = (uint32_t*)(ws + o_torder);
  uint32_t* d_khist = (uint32_t*)(ws + o_tkeys);
  uint32_t* d_kcursor = d_khist + TASK_KEYS;
  uint32_t* d_partial = (uint32_t*)(ws + o_partial);
  uint32_t* d_bucket = (uint32_t*)(ws + o_bucket);
  uint32_t* d_seg = (uint32_t*)(ws + o_seg);
  uint32_t* d_seg2 = (uint32_t*)(ws + o_seg2);
  uint32_t* d_tot = (uint32_t*)(ws + o_res);
  uint32_t* d_win = d_tot + 4;
  uint32_t* d_big_count = (uint32_t*)(ws + o_big);
  uint32_t* d_big_list = d_big_count + 4;
  uint2* d_slices = (uint2*)(ws + o_slices);
  br.list = (uint2*)(ws + o_brlist);
  br.count = (uint32_t*)(ws + o_brcount);
  br.gcursor = (uint32_t*)(ws + o_gcur);
  if (!coop_sort) br.big = 0xffffffffu;    // the non-tiled scatter has no cooperative form: nothing is "big"

  if (!ctx.msm_attr_set) {   // per device: a process may drive several GPUs
    const int lds_max = 32768 * 4;
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_hist_kernel<true, uint32_t, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_hist_kernel<false, uint32_t, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_hist_kernel<false, uint64_t, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_hist_kernel<false, uint32_t, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_hist_kernel<false, uint64_t, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_scatter_kernel<true, uint32_t>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_scatter_kernel<false, uint32_t>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_scatter_kernel<false, uint64_t>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part1_scatter_tiled_kernel<uint32_t, P1_IPT_DEFAULT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part1_scatter_tiled_kernel<uint64_t, P1_IPT_DEFAULT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part1_scatter_tiled_kernel<uint32_t, P1_BIG_IPT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
    const int lds_all = 160 * 1024 - 6 * 1024;     // what a workgroup may ask for dynamically beside the v2 kernels' static tables
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part1_scatter_v2_kernel<P1_IPT_DEFAULT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_all));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part1_scatter_v2_kernel<P1_BIG_IPT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_all));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_scatter_v2_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_all));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_part2_scatter_v2_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_all));
    ctx.msm_attr_set = true;
  }
  hipEvent_t* ev = sl.ev;
  {
    const int rc = msm_slot_prepare(sl);
    if (rc != HM_OK) return rc;
  }
  sl.SW = RW;                                     // what the host fold reads: window sums, or one-bit records of a shared set
  sl.c = c;
  sl.W = W;
  sl.balanced = k4_2d;                            // ... which weigh one bit each
  if (k4_2d) std::memset(sl.win_bits, 1, RW);
  sl.group = group;
  sl.res_stride = res_stride;
  sl.live_ptr = nullptr;                          // the general pipeline writes all over the workspace
  sl.phase_timed = msm_phase_timing(false);
  const bool pt = sl.phase_timed;
  sl.T_max = T_max;
  sl.d_win = d_win;
  sl.d_tot = d_tot;
  HM_HIP_CHECK(hipEventRecord(ev[0], stream));

  // ---- K0 ------------------------------------------------------------------------------------
  uint32_t n_wide = W;                                       // a shared bucket set: the balanced split its table was built for
  if (single_set) {
    uint32_t wide;
    balanced_windows(W, &wide, &n_wide);
    if (wide != c) return hm_fail(HM_ERR_INTERNAL, "msm: the base set's table was built for another window split");
  }
  {
    MsmGroupScalars list;
    for (uint32_t e = 0; e < (uint32_t)HM_MSM_GROUP; ++e) list.s[e] = d_scalars_list[e < group ? e : 0];
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n + 255) / 256), group), dim3(256), 0, stream, list, d_inf, d_digits, n, c, W,
                       n_wide);
  }
  HM_HIP_CHECK(hipGetLastError());
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[1], stream));

  // ---- K2 ------------------------------------------------------------------------------------
  {
    {
      const uint32_t blocks = (uint32_t)std::min<size_t>(((size_t)NBT + 255) / 256, 2048);
      hipLaunchKernelGGL(msm_init_counters_kernel, dim3(blocks < 4 ? 4 : blocks), dim3(256), 0, stream, d_bcnt, (size_t)NBT,
                         cb ? br.count : (uint32_t*)nullptr, d_khist, TASK_KEYS, d_big_count);
      HM_HIP_CHECK(hipGetLastError());
    }
    const int rc = wide_items
                       ? launch_sort<uint64_t>(d_digits, d_chist, d_ctot, d_cstart, d_tmp, d_bcnt, sn, chunk, G, SW, fb, ib, cb,
                                               NC, NBP, br, stream)
                       : launch_sort<uint32_t>(d_digits, d_chist, d_ctot, d_cstart, d_tmp, d_bcnt, sn, chunk, G, SW, fb, ib, cb,
                                               NC, NBP, br, stream);
    if (rc != HM_OK) return rc;
  }
  // pair count for effective_task_len: the first sort level leaves it behind its region starts; an
  // input small enough to need no first level keeps the host's L
  const uint32_t* d_pairs = cb ? (const uint32_t*)(d_cstart + (size_t)SW * NC) : (const uint32_t*)nullptr;
  {
    const uint32_t nblocks = (NBT + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(msm_scan_partial_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, stream, (const uint32_t*)d_bcnt,
                       d_bsum, NBT, d_pairs, L);
    hipLaunchKernelGGL(msm_scan_blocksums_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, d_bsum, nblocks, d_tot,
                       (uint32_t)std::min<uint64_t>(T_max, 0xffffffffull));
    hipLaunchKernelGGL(msm_scan_final_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, stream, (const uint32_t*)d_bcnt,
                       (const uint32_t*)d_bsum, d_boff, d_toff, (const uint32_t*)d_tot, NBT, d_pairs, L,
                       coop_sort ? br.gcursor : (uint32_t*)nullptr);
    HM_HIP_CHECK(hipGetLastError());
  }
  {
    ps.chist = d_chist;
    const int rc = wide_items ? launch_sort_scatter<uint64_t>(d_digits, d_cstart, d_tmp, d_boff, d_sorted, sn, SW, fb, ib, cb, NC,
                                                              NBP, NBT, br, ps, stream)
                              : launch_sort_scatter<uint32_t>(d_digits, d_cstart, d_tmp, d_boff, d_sorted, sn, SW, fb, ib, cb, NC,
                                                              NBP, NBT, br, ps, stream);
    if (rc != HM_OK) return rc;
  }
  if (pairs_max < (1u << 19)) {
    // small inputs: three more launches cost more than the idle lanes they would save
    hipLaunchKernelGGL(msm_task_fill_kernel, dim3((NBT + 255) / 256), dim3(256), 0, stream, (const uint32_t*)d_toff, d_tb, d_torder,
                       NBT, (const uint32_t*)d_tot);
  } else {
    const uint32_t og = (NBT + ORDER_THREADS * ORDER_ITEMS - 1) / (ORDER_THREADS * ORDER_ITEMS);
    hipLaunchKernelGGL(msm_task_hist_kernel, dim3(og), dim3(ORDER_THREADS), 0, stream, (const uint32_t*)d_bcnt, NBT,
                       d_pairs, L, d_khist);
    hipLaunchKernelGGL(msm_task_keyscan_kernel, dim3(1), dim3(ORDER_THREADS), 0, stream, (const uint32_t*)d_khist, d_kcursor);
    hipLaunchKernelGGL(msm_task_order_kernel, dim3(og), dim3(ORDER_THREADS), 0, stream, (const uint32_t*)d_bcnt,
                       (const uint32_t*)d_toff, NBT, d_pairs, L, d_kcursor, d_tb, d_torder, (const uint32_t*)d_tot);
  }
  HM_HIP_CHECK(hipGetLastError());
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[2], stream));

  // ---- K3 ------------------------------------------------------------------------------------
  // The exact task count T stays on the device (d_tot[1]); the grid covers its host-side bound: for
  // uniform scalars every bucket holds one task (T ~ NBT), else at most pairs / L more.  Surplus
  // single-wave workgroups exit at once.
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[5], stream));
  {
    const uint64_t t_grid = T_max;
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3((uint32_t)((t_grid + ACC_THREADS - 1) / ACC_THREADS)), dim3(ACC_THREADS), 0,
                       stream, (const uint32_t*)d_sorted, (const uint32_t*)d_tb, (const uint32_t*)d_torder, (const uint32_t*)d_boff,
                       (const uint32_t*)d_bcnt, (const uint32_t*)d_toff, d_xy, d_partial, (const uint32_t*)d_tot, d_pairs, L);
    HM_HIP_CHECK(hipGetLastError());
  }
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[6], stream));
  hipLaunchKernelGGL(msm_bucket_finalize_kernel, dim3((NBT + ACC_THREADS - 1) / ACC_THREADS), dim3(ACC_THREADS), 0, stream,
                     (const uint32_t*)d_partial, (const uint32_t*)d_toff, d_bucket, NBT, d_big_count, d_big_list, d_slices,
                     (const uint32_t*)d_tot);
  HM_HIP_CHECK(hipGetLastError());
  {
    uint32_t slice_grid = (uint32_t)(T_max / FINALIZE_SLICE + 1);       // upper bound on the number of full slices
    if (slice_grid > 2048) slice_grid = 2048;                           // both kernels stride over their queues
    hipLaunchKernelGGL(msm_bucket_finalize_slices_kernel, dim3(slice_grid), dim3(WIN_THREADS), 0, stream, d_partial,
                       (const uint32_t*)d_toff, (const uint32_t*)d_big_count, (const uint2*)d_slices);
    uint32_t big_grid = (uint32_t)(T_max / (FINALIZE_SERIAL + 1) + 1);  // upper bound on the number of queued buckets
    if (big_grid > 2048) big_grid = 2048;
    hipLaunchKernelGGL(msm_bucket_finalize_big_kernel, dim3(big_grid), dim3(WIN_THREADS), 0, stream,
                       (const uint32_t*)d_partial, (const uint32_t*)d_toff, d_bucket, (const uint32_t*)d_big_count,
                       (const uint32_t*)d_big_list);
    HM_HIP_CHECK(hipGetLastError());
  }
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[3], stream));

  // ---- K4 ------------------------------------------------------------------------------------
  if (k4_2d) {
    hipLaunchKernelGGL(msm_reduce_rowcol_kernel, dim3((1u << k4_m) + (1u << k4_h), SW), dim3(WIN_THREADS), 0, stream,
                       (const uint32_t*)d_bucket, d_seg, k4_m, k4_h);
    hipLaunchKernelGGL(msm_reduce_bits_kernel, dim3(RW, SW), dim3(WIN_THREADS), 0, stream, (const uint32_t*)d_seg, k4_m, k4_h, d_win,
                       res_stride);
    HM_HIP_CHECK(hipGetLastError());
  } else {
  hipLaunchKernelGGL(msm_reduce_segments_kernel, dim3((SW * nseg + ACC_THREADS - 1) / ACC_THREADS), dim3(ACC_THREADS), 0,
                     stream, (const uint32_t*)d_bucket, d_seg, SW, NB, NBP, SEG, nseg);
  HM_HIP_CHECK(hipGetLastError());
  {
    uint32_t* cur = d_seg;
    uint32_t* nxt = d_seg2;
    uint32_t count = nseg;
    while (count > 1) {
      const uint32_t out_count = (count + SUM_SPAN - 1) / SUM_SPAN;
      hipLaunchKernelGGL(msm_sum_points_kernel, dim3(out_count, SW), dim3(WIN_THREADS), 0, stream, (const uint32_t*)cur, count, nxt,
                         out_count);
      HM_HIP_CHECK(hipGetLastError());
      uint32_t* t = cur; cur = nxt; nxt = t;
      count = out_count;
    }
    hipLaunchKernelGGL(msm_windows_to_ext_kernel, dim3((SW + 63) / 64), dim3(64), 0, stream, (const uint32_t*)cur, SW, d_win);
    HM_HIP_CHECK(hipGetLastError());
  }
  }
  if (RW > 128) return hm_fail(HM_ERR_INTERNAL, "msm: more than 128 windows");
  // pinned landing zone, so that this copy (and therefore msm_enqueue) does not wait for the device
  HM_HIP_CHECK(hipMemcpyAsync(sl.totals(), d_tot, (size_t)(single_set ? group : 1u) * res_stride * 4, hipMemcpyDeviceToHost, stream));
  HM_HIP_CHECK(hipEventRecord(ev[4], stream));
  return HM_OK;
}

int msm_enqueue(DeviceCtx& ctx, int slot, const uint32_t* d_scalars_ext, const uint32_t* d_xy, const uint8_t* d_inf, size_t n,
                uint32_t precomp_c, hipStream_t stream) {
  MsmSlot& sl = ctx.msm_slots[slot];
  sl.n = n;
  sl.stream = stream;
  sl.group = 1;
  if (n == 0) return HM_OK;
  return msm_issue(ctx, slot, &d_scalars_ext, 1, d_xy, d_inf, n, precomp_c, stream);
}

// Several DENSE commitments over one fixed-base table in one chain of the general pipeline: every element is a bucket set
// of the same sort launches, the same K3 launch and the same two reduction launches (HALO2_MI355X_GROUP_DENSE: how many,
// default 8; 1 = a chain each).  Applies where a single such MSM runs the general pipeline on the table (not the five-launch
// plan) and the whole chain's (point, bucket) pairs stay below 2^31.
uint32_t msm_table_group_max(size_t n, uint32_t precomp_c) {
  static const uint32_t want = [] {
    const char* v = std::getenv("HALO2_MI355X_GROUP_DENSE");
    const int g = v && *v ? std::atoi(v) : 8;
    return (uint32_t)(g < 1 ? 1 : g > HM_MSM_GROUP ? HM_MSM_GROUP : g);
  }();
  static const bool k4_chain_only = [] { const char* v = std::getenv("HALO2_MI355X_K4_CHAIN"); return v && *v == '1'; }();
  // (from 2^19 points a commitment fills the chip by itself, and the positional sort plan of the larger sizes has only been
  //  exercised one bucket set at a time)
  if (precomp_c < 6 || n == 0 || n >= ((size_t)1 << 19) || k4_chain_only || msm_small_applies(n, precomp_c, true)) return 1;
  const uint64_t per = (uint64_t)n * ((255 + precomp_c - 1) / precomp_c);
  uint32_t g = want;
  while (g > 1 && per * g >= (1ull << 31)) --g;
  while (g > 1 && per * g * 4 > ((uint64_t)3 << 30)) --g;       // keep a chain's item arrays below 3 GiB each
  return g;
}
int msm_enqueue_table_group(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                            const uint8_t* d_inf, size_t n, uint32_t precomp_c, hipStream_t stream) {
  MsmSlot& sl = ctx.msm_slots[slot];
  if (group == 0 || group > msm_table_group_max(n, precomp_c)) return hm_fail(HM_ERR_INTERNAL, "msm: table group larger than the plan allows");
  sl.n = n;
  sl.stream = stream;
  sl.group = 1;
  if (n == 0) return HM_OK;
  return msm_issue(ctx, slot, d_scalars_list, group, d_xy, d_inf, n, precomp_c, stream);
}

// The window plan of the five-launch chain for n points, or 0 when it does not apply (n >= 2^19, a precomputed set,
// an override outside its range): the same choice msm_issue makes for a single MSM.
// `live_rows` (0 = unknown): an upper bound on the rows that survive the digits kernel's compaction -- the window is
// picked for THOSE (an advice column with 1 100 used rows of 2^18 is a 2^10-point MSM: c = 7, not 15: its buckets are
// then dense, and the reduction's per-bucket scalar multiple has 7 bits instead of 14)
static uint32_t small_plan_window(size_t n, uint32_t precomp_c, size_t live_rows = 0) {
  if (precomp_c != 0 || n == 0 || n >= (1u << 19)) return 0;
  static const int8_t kWindowSmallPlan[19] = {4, 4, 4, 4, 4, 4, 4, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 15};
  const int window_override = g_window_override.load(std::memory_order_relaxed);
  const size_t sized_for = live_rows != 0 && live_rows < n ? live_rows : n;
  const uint32_t c = window_override ? (uint32_t)window_override : (uint32_t)kWindowSmallPlan[ilog2(sized_for)];
  return msm_small_applies(n, c, false) ? c : 0;
}
bool msm_group_applies(size_t n, uint32_t precomp_c) { return small_plan_window(n, precomp_c) != 0; }

// `group` MSMs over the same n points through ONE launch chain (msm_small.hip); the slot then carries `group` results.
int msm_enqueue_group(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                      const uint8_t* d_inf, size_t n, hipStream_t stream, size_t live_rows) {
  MsmSlot& sl = ctx.msm_slots[slot];
  const uint32_t c = small_plan_window(n, 0, live_rows);
  if (!c) return hm_fail(HM_ERR_INTERNAL, "msm: a group needs the five-launch plan");
  sl.n = n;
  sl.stream = stream;
  return msm_issue_small(ctx, slot, d_scalars_list, group, d_xy, d_inf, n, c, stream);
}

// Wait for the slot's chain and fold the window sums of each of its MSMs (host Horner + affine normalisation).  Touches
// only the slot: a busy slot is not handed out again, so this part runs without the context lock.
int msm_finish_wait_fold(MsmSlot& sl, uint64_t* out_jac_ext, int* out_is_identity, double* host_us) {
  *host_us = 0;
  if (sl.n == 0) {
    std::memset(out_jac_ext, 0, 96);
    *out_is_identity = 1;
    return HM_OK;
  }
  HM_HIP_CHECK(hipEventSynchronize(sl.ev[4]));
  const auto f0 = std::chrono::steady_clock::now();
  for (uint32_t e = 0; e < sl.group; ++e) {
    const uint32_t* land = sl.h_land + (size_t)e * sl.res_stride;      // group == 1: res_stride is not used
    if (land[2] != 0) return hm_fail(HM_ERR_INTERNAL, "msm: task count exceeds its bound");
    host_fold(land + 4, sl.SW, sl.c, out_jac_ext + 12 * e, out_is_identity + e, sl.balanced ? sl.win_bits : nullptr);   // SW == 1: just the normalisation
  }
  *host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f0).count();
  return HM_OK;
}

// statistics of the MSM just finished on `slot` (ctx.mu held)
void msm_finish_record(DeviceCtx& ctx, int slot, double host_us) {
  MsmSlot& sl = ctx.msm_slots[slot];
  if (sl.n == 0) return;
  ctx.calls.msm_host_us += host_us;
  hipEvent_t* ev = sl.ev;
  float ms[4] = {0, 0, 0, 0}, total = 0, acc_kernel = 0;
  (void)hipEventElapsedTime(&total, ev[0], ev[4]);
  if (sl.phase_timed) {
    for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    (void)hipEventElapsedTime(&acc_kernel, ev[5], ev[6]);
  }
  ctx.last_msm.t_accum_kernel_ms = acc_kernel;
  ctx.last_msm.t_digits_ms = ms[0];
  ctx.last_msm.t_sort_ms = ms[1];
  ctx.last_msm.t_accum_ms = ms[2];
  ctx.last_msm.t_reduce_ms = ms[3];
  ctx.last_msm.t_total_ms = total;
  ctx.last_msm.pairs = sl.totals()[0];
  ctx.last_msm.tasks = sl.totals()[1];
  ctx.last_msm.c = sl.c;
  ctx.last_msm.windows = sl.W;
}

int msm_finish(DeviceCtx& ctx, int slot, uint64_t out_jac_ext[12], int* out_is_identity) {
  double host_us = 0;
  const int rc = msm_finish_wait_fold(ctx.msm_slots[slot], out_jac_ext, out_is_identity, &host_us);
  if (rc == HM_OK) msm_finish_record(ctx, slot, host_us);
  else ctx.msm_slots[slot].live_ptr = nullptr;     // a chain that failed may have left its block counters anywhere
  return rc;
}

// the synchronous form: slot 0, enqueue then finish
int msm_run(DeviceCtx& ctx, const uint32_t* d_scalars_ext, const uint32_t* d_xy, const uint8_t* d_inf, size_t n,
            uint32_t precomp_c, uint64_t out_jac_ext[12], int* out_is_identity, hipStream_t stream) {
  if (ctx.msm_slots[0].busy) return hm_fail(HM_ERR_BAD_ARG, "msm: slot 0 is held by an un-awaited hm_msm_submit_dev ticket");
  const int rc = msm_enqueue(ctx, 0, d_scalars_ext, d_xy, d_inf, n, precomp_c, stream);
  if (rc != HM_OK) return rc;
  return msm_finish(ctx, 0, out_jac_ext, out_is_identity);
}

int g1_fixed_base_mul_run(DeviceCtx& ctx, const uint32_t* d_scalars_ext, size_t n, const uint64_t base_affine_ext[8],
                          uint32_t* d_out_affine_ext, hipStream_t stream) {
  if (n == 0) return HM_OK;
  AuxSlot* slot = aux_acquire(ctx, stream);          // the multiples table is per stream in use
  if (!slot) return HM_ERR_HIP;
  uint32_t* d_table = (uint32_t*)slot->table.ensure((size_t)64 * 15 * PT_WORDS * 4);
  if (!d_table) return hm_fail(HM_ERR_HIP, "fixed-base: table allocation failed");
  G1AffineWords base;
  std::memcpy(base.w, base_affine_ext, 64);
  hipLaunchKernelGGL(g1_fixed_table_kernel, dim3(1), dim3(64), 0, stream, base, d_table);
  HM_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(g1_fixed_base_mul_kernel, dim3((uint32_t)((n + ACC_THREADS - 1) / ACC_THREADS)), dim3(ACC_THREADS), 0,
                     stream, d_scalars_ext, (const uint32_t*)d_table, d_out_affine_ext, n);
  HM_HIP_CHECK(hipGetLastError());
  return aux_release(ctx, slot, stream);
}

}  // namespace hm
