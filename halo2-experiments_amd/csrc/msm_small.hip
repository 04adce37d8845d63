// msm_small.hip -- the short launch chain for prover-sized MSMs (n < 2^19: the sizes of every circuit
// in the reference, /root/reference/src/circuits/utils.rs:22-70 -- k = 9 in its own test, k = 11 / 17 / 18
// in BASELINE.json's configs).
//
// At these sizes the general pipeline of msm.hip (~26 launches: digit array, two-level LDS sort, three-
// launch scan, task sort, three finalize kernels, segment sums, two tree passes) is bound by launch
// gaps and by the DEPTH of its latency-bound phases, not by throughput: a lone wave issues one VALU
// instruction per ~4.2 cycles, i.e. ~7 us per Jacobian addition, and the bucket reduction of a 2^18
// MSM ran ~45 serial additions / doublings per lane in 544 lone waves (0.45 ms of a 1.0 ms call).
// Here the whole MSM is FIVE launches and one D2H copy, nothing is read back in between:
//   A  digits    one lane per scalar: canonical form, W signed digits.  The 255 digit bits are spread over
//                the W windows as evenly as possible (widths c and c - 1) instead of W - 1 full windows and a
//                short top one, whose few buckets would each receive n / 2^(few bits) points
//   B  sort      one workgroup per (window, bucket range) does the whole counting sort of its share in LDS:
//                histogram, scan -> bucket offsets, task offsets and the task list in order of decreasing
//                chain length, then the scatter with LDS cursors.  Its runs in the global arrays are
//                reserved with ONE atomic per workgroup, so there is no global scan, no per-item global
//                atomic (tried first: 4.4 M of them at 2^18 cost 0.17 ms per pass) and nothing to clear
//   C  K3        the same accumulation chain as the general path (msm_dev.h: accumulate_chain)
//   D  reduce 1  per (window, range): bucket = sum of its task partials (buckets cut into many tasks: by groups of
//                lanes, all such buckets of the workgroup at once); then a workgroup takes the form its range needs:
//                DENSE ranges -- running sums over SEG buckets per lane, short scalar multiple, LDS tree; SPARSE ranges
//                (few non-empty buckets: advice columns) -- one lane per NON-EMPTY bucket, its sum times its weight,
//                LDS tree, so the work follows the data instead of the bucket count
//   E  reduce 2  per window: tree over D's outputs -> external format
// SEG is chosen so that D runs at most one wave per SIMD (the depth of a lane's chain is what costs, not the work).
//
// GROUPS.  The commitments of one prover phase share n, the window plan and the base set, and each of them alone is
// bound by launch gaps and chain depth.  So the five launches take a group dimension: up to HM_MSM_GROUP scalar arrays
// run through ONE chain (grid.y / grid.z = the group element, every workspace pointer advanced by e * ws_stride bytes,
// results landing back to back for one D2H copy).  A single MSM is the group of one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "hm_internal.h"
#include "msm_dev.h"

namespace hm {

constexpr int SA_THREADS = 256;       // A: one lane per scalar
constexpr int SS_THREADS = 1024;      // B: one workgroup per (window, bucket range)
constexpr uint32_t S_TASK_KEYS = 1024;
constexpr uint32_t S_HOT = 32;
constexpr uint32_t S_FINALIZE_SERIAL = 12;   // partials of one bucket its own lane sums; more go to a group of lanes
constexpr uint32_t S_HOT_PER_LANE = 4;       // partials per lane of such a group
constexpr uint32_t S_HOT_LIST = 64;
constexpr uint32_t S_MAX_L = 1023;

// ---- A: digits, balanced window widths ----------------------------------------------------------
// Window w is `wide` bits for w < n_wide, else wide - 1 bits (n_wide * wide + (W - n_wide) * (wide - 1) = 255).
// The top window holds bit 254, which is zero for every canonical scalar, so its digit never carries out.
struct SmallGroupScalars {       // the scalar arrays of one group, by value
  const uint32_t* s[HM_MSM_GROUP];
};
template <class T>
__device__ __forceinline__ T* ws_at(T* p, size_t ws_stride, uint32_t e) {   // element e's copy of a workspace array
  return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (size_t)e * ws_stride);
}

__global__ __launch_bounds__(SA_THREADS) void msm_s_digits_kernel(SmallGroupScalars group, const uint8_t* __restrict__ inf,
                                                                  int32_t* __restrict__ digits, uint32_t n, uint32_t wide,
                                                                  uint32_t n_wide, uint32_t W, uint32_t* __restrict__ gctr,
                                                                  uint32_t* __restrict__ blkidx, uint32_t* __restrict__ live_counts,
                                                                  size_t ws_stride) {
  const uint32_t e = blockIdx.y;
  const uint32_t* __restrict__ scalars = group.s[e];
  digits = ws_at(digits, ws_stride, e);
  gctr = ws_at(gctr, ws_stride, e);
  blkidx = ws_at(blkidx, ws_stride, e);
  uint32_t* live_count = live_counts + e;       // the group's counters lie together (cleared by one memset before this launch)
  const uint32_t n_pad = (n + SA_THREADS - 1) / SA_THREADS * SA_THREADS;     // row stride of the digit array
  const uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x;
  if (i == 0) {                 // the two run-reservation counters of B start from zero (stream order: before B)
    gctr[0] = 0;
    gctr[1] = 0;
  }
  const bool in_range = i < n;
  const uint4* q = reinterpret_cast<const uint4*>(scalars + (size_t)(in_range ? i : 0) * 8);
  const uint4 lo = q[0], hi = q[1];
  const uint32_t w_in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Fr k32 = fe_zero<FrParams>();
  k32.l[0] = 32;                // s_mont * 32 * 2^-261 = s_mont * 2^-256 = the canonical scalar (Fr::to_repr)
  const Fr s = fe_canonical(fe_mul(fe_unpack<FrParams>(w_in), k32));
  uint32_t v[9];
  {
    uint32_t t[8];
    fe_pack(t, s);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = t[k];
    v[8] = 0;
  }
  // Rows with a zero scalar (most rows of an advice column) or an identity base contribute nothing.  The digit array is
  // written COMPACTED BY BLOCKS of SA_THREADS rows: a workgroup without a surviving row writes nothing, the others take
  // the next free block (one atomic per workgroup) and note which rows it holds in blkidx -- so the sort streams
  // m = 256 x (blocks with a survivor) rows per window instead of n, and a dense column pays one table lookup per block.
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) any |= v[k];
  const bool live = in_range && inf[i] == 0 && any != 0;
  __shared__ uint32_t wg_block;
  if (!__syncthreads_or(live ? 1 : 0)) return;
  if (threadIdx.x == 0) {
    wg_block = atomicAdd(live_count, 1u);
    blkidx[wg_block] = blockIdx.x;
  }
  __syncthreads();
  const uint32_t j = wg_block * SA_THREADS + threadIdx.x;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < W; ++w) {
    const uint32_t bits = w < n_wide ? wide : wide - 1;
    const uint32_t mask_w = (1u << bits) - 1u, half = 1u << (bits - 1);
    const uint32_t d = (v[0] & mask_w) + carry;
    int32_t sd;
    if (d > half) {
      sd = (int32_t)d - (int32_t)(1u << bits);
      carry = 1;
    } else {
      sd = (int32_t)d;
      carry = 0;
    }
    digits[(size_t)w * n_pad + j] = live ? sd : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __funnelshift_r(v[k], v[k + 1], bits);
  }
}

// ---- B: sort of one (window, bucket range): histogram, scan, task list, scatter -- in one workgroup's LDS ---
__device__ __forceinline__ uint32_t s_task_key(uint32_t len) { return (S_TASK_KEYS - 1) - len; }   // len <= S_MAX_L; key 0 = longest

// exclusive scan of one value per thread over the 1024-thread workgroup; `total` gets the sum
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* wsum, uint32_t& total) {
  const uint32_t tid = threadIdx.x;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t u = __shfl_up(incl, off, 64);
    if ((tid & 63) >= (uint32_t)off) incl += u;
  }
  __syncthreads();                       // wsum may still be read by a previous scan
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  if (tid < 64) {
    const uint32_t ws = tid < (SS_THREADS / 64) ? wsum[tid] : 0;
    uint32_t wi = ws;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t u = __shfl_up(wi, off, 64);
      if (tid >= (uint32_t)off) wi += u;
    }
    if (tid < (SS_THREADS / 64)) wsum[tid] = wi - ws;     // exclusive wave offsets
    if (tid == (SS_THREADS / 64) - 1) wsum[16] = wi;       // grand total
  }
  __syncthreads();
  total = wsum[16];
  return wsum[tid >> 6] + incl - v;
}

// f(index, digit) over the n digits of one window with the whole workgroup: four 16-byte loads in flight
// per lane (one workgroup streams a whole window, so memory-level parallelism has to come from the lane)
template <class F>
__device__ __forceinline__ void window_for_each_digit(const int32_t* __restrict__ dw, uint32_t n, F f) {
  const uint32_t tid = threadIdx.x;
  uint32_t a = 0;
  while (a < n && (reinterpret_cast<uintptr_t>(dw + a) & 15)) ++a;      // <= 3 head words up to 16-byte alignment
  if (tid < a) f(tid, dw[tid]);
  const uint32_t nvec = (n - a) / 4, tail = a + nvec * 4;
  if (tid < n - tail) f(tail + tid, dw[tail + tid]);
  const int4* vb = reinterpret_cast<const int4*>(dw + a);
  for (uint32_t v0 = 0; v0 < nvec; v0 += 4 * SS_THREADS) {
    int4 q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t v = v0 + k * SS_THREADS + tid;
      q[k] = v < nvec ? vb[v] : make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t v = v0 + k * SS_THREADS + tid;
      if (v < nvec) {
        const uint32_t i = a + v * 4;
        f(i, q[k].x);
        f(i + 1, q[k].y);
        f(i + 2, q[k].z);
        f(i + 3, q[k].w);
      }
    }
  }
}

// Virtual window v = w * H + h owns the buckets |digit| in (h * NBh, (h + 1) * NBh] of window w, locally 1 .. NBh.
// toff[v * (NBh + 2) + b] = first partial index of local bucket b (global numbering), [NBh + 1] = end.
// A task descriptor is { start in `sorted`, partial index, chain length, - }; this workgroup's descriptors are
// written in order of decreasing length.
__global__ __launch_bounds__(SS_THREADS) void msm_s_sort_kernel(const int32_t* __restrict__ digits, uint32_t* __restrict__ toff,
                                                                uint32_t* __restrict__ gctr, uint4* __restrict__ desc,
                                                                uint32_t* __restrict__ sorted, uint32_t* __restrict__ nz,
                                                                uint32_t* __restrict__ nzc, uint32_t n, uint32_t NBh, uint32_t H,
                                                                uint32_t L, const uint32_t* __restrict__ blkidx,
                                                                const uint32_t* __restrict__ live_counts, size_t ws_stride) {
  extern __shared__ uint32_t sm[];
  __shared__ uint32_t blk_tab[(1u << 19) / SA_THREADS];      // compacted block -> first original row / SA_THREADS
  uint32_t m;                                      // rows of the compacted digit array of this MSM (A)
  const uint32_t n_pad = (n + SA_THREADS - 1) / SA_THREADS * SA_THREADS;
  {
    const uint32_t e = blockIdx.y;
    const uint32_t nblk = live_counts[e];
    m = nblk * SA_THREADS;
    blkidx = ws_at(blkidx, ws_stride, e);
    for (uint32_t b = threadIdx.x; b < nblk; b += SS_THREADS) blk_tab[b] = blkidx[b];
    digits = ws_at(digits, ws_stride, e);
    toff = ws_at(toff, ws_stride, e);
    gctr = ws_at(gctr, ws_stride, e);
    desc = ws_at(desc, ws_stride, e);
    sorted = ws_at(sorted, ws_stride, e);
    nz = ws_at(nz, ws_stride, e);
    nzc = ws_at(nzc, ws_stride, e);
  }
  const uint32_t NBP = NBh + 1;
  uint32_t* cnt = sm;                         // bucket counts, then the scatter cursors
  uint32_t* kh = cnt + ((NBP + 31) & ~31u);   // key histogram, later the per-key local cursors
  uint32_t* kbase = kh + S_TASK_KEYS;         // first slot of every key
  uint32_t* wsum = kbase + S_TASK_KEYS;       // 17 words of scan scratch
  __shared__ uint32_t hot_n, hot[S_HOT][4], base_pt[2];
  const uint32_t v = blockIdx.x, w = v / H, h = v - w * H, tid = threadIdx.x;
  const uint32_t blo = h * NBh;               // this range: blo < |digit| <= blo + NBh
  const int32_t* dw = digits + (size_t)w * n_pad;
  for (uint32_t b = tid; b < NBP; b += SS_THREADS) cnt[b] = 0;
  kh[tid] = 0;
  if (tid == 0) hot_n = 0;
  __syncthreads();
  window_for_each_digit(dw, m, [&](uint32_t, int32_t d) {
    const uint32_t lb = (uint32_t)(d < 0 ? -d : d) - blo;          // wraps for |d| <= blo
    if (lb - 1u < NBh) (void)lds_inc(cnt, lb);
  });
  __syncthreads();
  const uint32_t per = (NBP + SS_THREADS - 1) / SS_THREADS;
  const uint32_t b0 = tid * per < NBP ? tid * per : NBP, b1 = b0 + per < NBP ? b0 + per : NBP;
  uint32_t sp = 0, st = 0, sz = 0;
  for (uint32_t b = b0; b < b1; ++b) {
    const uint32_t c = cnt[b], full = c / L, rem = c - full * L;
    sp += c;
    st += full + (rem ? 1u : 0u);
    sz += c ? 1u : 0u;
    if (full) atomicAdd(&kh[s_task_key(L)], full);
    if (rem) atomicAdd(&kh[s_task_key(rem)], 1u);
  }
  uint32_t tot_p = 0, tot_t = 0, tot_k = 0, tot_z = 0;
  uint32_t rp = block_excl_scan_1024(sp, wsum, tot_p);
  uint32_t rt = block_excl_scan_1024(st, wsum, tot_t);
  uint32_t rz = block_excl_scan_1024(sz, wsum, tot_z);     // the list of non-empty buckets (reduce 1 picks its form by it)
  {
    const uint32_t kv = kh[tid];
    const uint32_t ke = block_excl_scan_1024(kv, wsum, tot_k);
    kbase[tid] = ke;
    kh[tid] = 0;
  }
  if (tid == 0) {               // this workgroup's runs in `sorted` and in the task / partial numbering
    base_pt[0] = atomicAdd(&gctr[0], tot_p);
    base_pt[1] = atomicAdd(&gctr[1], tot_t);
  }
  __syncthreads();
  const uint32_t base_p = base_pt[0], base_t = base_pt[1];
  uint4* dsc = desc + base_t;
  uint32_t* tw = toff + (size_t)v * (NBh + 2);
  for (uint32_t b = b0; b < b1; ++b) {
    const uint32_t c = cnt[b], full = c / L, rem = c - full * L;
    const uint32_t start = base_p + rp;
    cnt[b] = rp;                              // the bucket's cursor for the scatter pass (local to this run)
    tw[b] = base_t + rt;
    if (c) nz[(size_t)v * NBh + rz++] = b;
    if (full) {
      const uint32_t key = s_task_key(L), at = kbase[key] + atomicAdd(&kh[key], full);
      const uint32_t slot = full > 64 ? atomicAdd(&hot_n, 1u) : S_HOT;
      if (slot < S_HOT) {       // a bucket cut into many tasks: listed by the whole workgroup below
        hot[slot][0] = start; hot[slot][1] = base_t + rt; hot[slot][2] = full; hot[slot][3] = at;
      } else {
        for (uint32_t i = 0; i < full; ++i) dsc[at + i] = make_uint4(start + i * L, base_t + rt + i, L, 0);
      }
    }
    if (rem) {
      const uint32_t key = s_task_key(rem), at = kbase[key] + atomicAdd(&kh[key], 1u);
      dsc[at] = make_uint4(start + full * L, base_t + rt + full, rem, 0);
    }
    rp += c;
    rt += full + (rem ? 1u : 0u);
  }
  if (tid == SS_THREADS - 1) {
    tw[NBP] = base_t + tot_t;
    nzc[v] = tot_z;
  }
  __syncthreads();
  const uint32_t nh = hot_n < S_HOT ? hot_n : S_HOT;
  for (uint32_t e = 0; e < nh; ++e) {
    const uint32_t start = hot[e][0], t0 = hot[e][1], full = hot[e][2], at = hot[e][3];
    for (uint32_t i = tid; i < full; i += SS_THREADS) dsc[at + i] = make_uint4(start + i * L, t0 + i, L, 0);
  }
  uint32_t* sw = sorted + base_p;
  window_for_each_digit(dw, m, [&](uint32_t j, int32_t d) {
    const uint32_t lb = (uint32_t)(d < 0 ? -d : d) - blo;
    if (lb - 1u < NBh) {
      const uint32_t pos = lds_inc(cnt, lb);
      const uint32_t i = blk_tab[j / SA_THREADS] * SA_THREADS + (j % SA_THREADS);
      sw[pos] = i | (d < 0 ? 0x80000000u : 0u);
    }
  });
}

// ---- C: accumulation ---------------------------------------------------------------------------
__global__ __launch_bounds__(ACC_THREADS) void msm_s_accumulate_kernel(const uint32_t* __restrict__ sorted,
                                                                       const uint4* __restrict__ desc,
                                                                       const uint32_t* __restrict__ gctr,
                                                                       const uint32_t* __restrict__ xy, uint32_t* __restrict__ partial,
                                                                       size_t ws_stride) {
  __shared__ uint4 stage[4][ACC_THREADS];
  {
    const uint32_t e = blockIdx.y;
    sorted = ws_at(sorted, ws_stride, e);
    desc = ws_at(desc, ws_stride, e);
    gctr = ws_at(gctr, ws_stride, e);
    partial = ws_at(partial, ws_stride, e);
  }
  const uint32_t lane = threadIdx.x;
  const uint32_t slot = blockIdx.x * ACC_THREADS + lane;
  if (slot >= gctr[1]) return;                 // the grid covers the host-side bound on the task count
  const uint4 d = desc[slot];
  const G1Jac res = accumulate_chain(sorted, xy, d.x, d.x + d.z, stage, lane);
  store_jac(partial + (size_t)d.y * PT_WORDS, res);
}

// ---- D: buckets -> one point per (virtual window, workgroup) ---------------------------------------------
// Buckets cut into many tasks (hot buckets of skewed columns): a group of ceil(count / 4) lanes (at most the whole
// workgroup) each sums a strided share of the partials, then the group is folded by a segmented tree; as many such
// buckets as fit 256 lanes are handled per round.  The sum is left in the bucket's first partial.  hot_b[0 .. nh)
// lists the workgroup's buckets with more than S_FINALIZE_SERIAL partials (collected by the caller).
__device__ __forceinline__ void reduce_hot_buckets(uint32_t* __restrict__ partial, const uint32_t* __restrict__ tw, uint32_t* tree,
                                                   const uint32_t* hot_b, uint32_t nh, uint32_t* hot_lane0, uint32_t* round_info) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t e0 = 0; e0 < nh;) {
    if (tid == 0) {
      uint32_t used = 0, e = e0, maxl = 1;
      while (e < nh) {
        const uint32_t cntp = tw[hot_b[e] + 1] - tw[hot_b[e]];
        uint32_t lanes = (cntp + S_HOT_PER_LANE - 1) / S_HOT_PER_LANE;
        if (lanes > WIN_THREADS) lanes = WIN_THREADS;
        if (used + lanes > WIN_THREADS) break;
        hot_lane0[e] = used;
        used += lanes;
        maxl = lanes > maxl ? lanes : maxl;
        ++e;
      }
      hot_lane0[e] = used;
      round_info[0] = e;
      round_info[1] = maxl;
    }
    __syncthreads();
    const uint32_t e1 = round_info[0], maxl = round_info[1];
    uint32_t my_e = e1, r = 0, lanes = 0;
    for (uint32_t e = e0; e < e1; ++e)
      if (tid >= hot_lane0[e] && tid < hot_lane0[e + 1]) {
        my_e = e;
        r = tid - hot_lane0[e];
        lanes = hot_lane0[e + 1] - hot_lane0[e];
      }
    G1Jac acc = g1_identity();
    uint32_t t0 = 0;
    if (my_e < e1) {
      t0 = tw[hot_b[my_e]];
      const uint32_t t1 = tw[hot_b[my_e] + 1];
      for (uint32_t t = t0 + r; t < t1; t += lanes) acc = g1_add(acc, load_jac(partial + (size_t)t * PT_WORDS));
    }
    store_jac(tree + tid * PT_WORDS, acc);
    __syncthreads();
    for (uint32_t off = 1; off < maxl; off <<= 1) {
      const bool act = my_e < e1 && (r & (2 * off - 1)) == 0 && r + off < lanes;
      G1Jac s = g1_identity();
      if (act) s = g1_add(load_jac(tree + tid * PT_WORDS), load_jac(tree + (tid + off) * PT_WORDS));
      __syncthreads();
      if (act) store_jac(tree + tid * PT_WORDS, s);
      __syncthreads();
    }
    if (my_e < e1 && r == 0) store_jac(partial + (size_t)t0 * PT_WORDS, load_jac(tree + tid * PT_WORDS));
    __syncthreads();
    e0 = e1;
  }
}

// the sum of bucket b's partials (its first partial already holds it if the bucket is in the workgroup's hot list)
__device__ __forceinline__ G1Jac bucket_value(const uint32_t* __restrict__ partial, const uint32_t* __restrict__ tw, uint32_t b,
                                              const uint32_t* hot_b, uint32_t nh) {
  const uint32_t t0 = tw[b], t1 = tw[b + 1];
  bool summed = false;
  if (t1 - t0 > S_FINALIZE_SERIAL)
    for (uint32_t e = 0; e < nh; ++e)
      if (hot_b[e] == b) summed = true;
  if (summed) return load_jac(partial + (size_t)t0 * PT_WORDS);
  G1Jac val = g1_identity();
  for (uint32_t t = t0; t < t1; ++t) val = g1_add(val, load_jac(partial + (size_t)t * PT_WORDS));
  return val;
}

__device__ __forceinline__ G1Jac small_multiple(const G1Jac& p, uint32_t m) {       // m * p by double-and-add, m < 2^31
  G1Jac rr = g1_identity();
  if (m == 0 || p.inf) return rr;
  for (int bit = 31 - __clz(m); bit >= 0; --bit) {
    rr = g1_double(rr);
    if ((m >> bit) & 1) rr = g1_add(rr, p);
  }
  return rr;
}

// A virtual window is "sparse" when at most NBh / S_SPARSE_DIV of its buckets hold anything (advice columns: a few
// hundred small values and six blinding rows in 2^18 rows): then the work is taken per NON-EMPTY bucket (form 2 below)
// instead of per segment of the whole bucket range.
constexpr uint32_t S_SPARSE_DIV = 8;

// One launch covers both forms (a workgroup takes the form its range needs, so dense and sparse ranges of one MSM run
// side by side): form 1 (dense) -- running sums over SEG buckets per lane, short scalar multiple; form 2 (sparse) -- one
// lane per non-empty bucket, its sum times its weight.  Either way an LDS tree folds the workgroup's lanes.
__global__ __launch_bounds__(WIN_THREADS) void msm_s_reduce1_kernel(uint32_t* __restrict__ partial, const uint32_t* __restrict__ toff,
                                                                    const uint32_t* __restrict__ nz, const uint32_t* __restrict__ nzc,
                                                                    uint32_t* __restrict__ seg1, uint32_t NBh, uint32_t H, uint32_t SEG,
                                                                    uint32_t G1n, size_t ws_stride) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  __shared__ uint32_t hot_n, hot_b[S_HOT_LIST], hot_lane0[S_HOT_LIST + 1], round_info[2];
  {
    const uint32_t e = blockIdx.z;
    partial = ws_at(partial, ws_stride, e);
    toff = ws_at(toff, ws_stride, e);
    nz = ws_at(nz, ws_stride, e);
    nzc = ws_at(nzc, ws_stride, e);
    seg1 = ws_at(seg1, ws_stride, e);
  }
  const uint32_t v = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
  const uint32_t h = v % H;
  const uint32_t* tw = toff + (size_t)v * (NBh + 2);
  const uint32_t count = nzc[v];
  G1Jac acc = g1_identity();
  uint32_t active = WIN_THREADS;                   // leading lanes of this workgroup that hold a term of the sum
  if (count > NBh / S_SPARSE_DIV) {
    const uint32_t sgi = g * WIN_THREADS + tid;
    const uint32_t lo = sgi * SEG + 1;             // local bucket numbers lo .. hi
    const bool valid = lo <= NBh;
    const uint32_t hi = valid ? (lo + SEG - 1 < NBh ? lo + SEG - 1 : NBh) : 0;
    {
      const uint32_t nseg = (NBh + SEG - 1) / SEG, first = g * WIN_THREADS;
      active = nseg > first ? (nseg - first < (uint32_t)WIN_THREADS ? nseg - first : (uint32_t)WIN_THREADS) : 0u;
    }
    if (tid == 0) hot_n = 0;
    __syncthreads();
    if (valid)
      for (uint32_t b = lo; b <= hi; ++b)
        if (tw[b + 1] - tw[b] > S_FINALIZE_SERIAL) {
          const uint32_t s = atomicAdd(&hot_n, 1u);
          if (s < S_HOT_LIST) hot_b[s] = b;
        }
    __syncthreads();
    const uint32_t nh = hot_n < S_HOT_LIST ? hot_n : S_HOT_LIST;
    reduce_hot_buckets(partial, tw, tree, hot_b, nh, hot_lane0, round_info);
    G1Jac run = g1_identity();
    if (valid) {
      for (uint32_t b = hi; b >= lo; --b) {
        run = g1_add(run, bucket_value(partial, tw, b, hot_b, nh));
        acc = g1_add(acc, run);
      }
      // sum (b - lo + 1) B_b is in acc; the bucket's weight is h * NBh + b, so (h * NBh + lo - 1) * sum B_b remains
      acc = g1_add(acc, small_multiple(run, h * NBh + lo - 1));
    }
  } else {
    const uint32_t* list = nz + (size_t)v * NBh;
    const uint32_t stride = G1n * WIN_THREADS;
    if (g * WIN_THREADS >= count) {                // nothing for this workgroup (the usual case of a sparse range): no tree
      if (tid == 0) store_jac(seg1 + ((size_t)v * G1n + g) * PT_WORDS, g1_identity());
      return;
    }
    active = count - g * WIN_THREADS < (uint32_t)WIN_THREADS ? count - g * WIN_THREADS : (uint32_t)WIN_THREADS;
    for (uint32_t i0 = g * WIN_THREADS; i0 < count; i0 += stride) {          // one round for every realistic count
      const uint32_t i = i0 + tid;
      const uint32_t b = i < count ? list[i] : 0;
      if (tid == 0) hot_n = 0;
      __syncthreads();
      if (b != 0 && tw[b + 1] - tw[b] > S_FINALIZE_SERIAL) {
        const uint32_t s = atomicAdd(&hot_n, 1u);
        if (s < S_HOT_LIST) hot_b[s] = b;
      }
      __syncthreads();
      const uint32_t nh = hot_n < S_HOT_LIST ? hot_n : S_HOT_LIST;
      reduce_hot_buckets(partial, tw, tree, hot_b, nh, hot_lane0, round_info);
      if (b != 0) acc = g1_add(acc, small_multiple(bucket_value(partial, tw, b, hot_b, nh), h * NBh + b));
      __syncthreads();
    }
  }
  // only the leading `active` lanes hold anything: ceil(log2 active) levels of the tree instead of eight
  const G1Jac res = active ? block_sum_points_upto(tree, acc, active) : g1_identity();
  if (tid == 0) store_jac(seg1 + ((size_t)v * G1n + g) * PT_WORDS, res);
}

// ---- E: one point per window, external format ------------------------------------------------------
__global__ __launch_bounds__(WIN_THREADS) void msm_s_reduce2_kernel(const uint32_t* __restrict__ seg1, uint32_t per_window,
                                                                    uint32_t* __restrict__ winres, const uint32_t* __restrict__ gctr,
                                                                    uint32_t* __restrict__ totals, size_t ws_stride, uint32_t res_stride,
                                                                    uint32_t* __restrict__ live_counts) {
  __shared__ uint32_t tree[WIN_THREADS * PT_WORDS];
  {
    const uint32_t e = blockIdx.y;                 // the results of a group lie back to back: res_stride words per element
    if (blockIdx.x == 0 && threadIdx.x == 0) live_counts[e] = 0;   // A's block counter: clean for the next chain of this slot
    seg1 = ws_at(seg1, ws_stride, e);
    gctr = ws_at(gctr, ws_stride, e);
    winres += (size_t)e * res_stride;
    totals += (size_t)e * res_stride;
  }
  const uint32_t w = blockIdx.x, tid = threadIdx.x;
  G1Jac acc = g1_identity();
  for (uint32_t t = tid; t < per_window; t += WIN_THREADS) acc = g1_add(acc, load_jac(seg1 + ((size_t)w * per_window + t) * PT_WORDS));
  const G1Jac r = block_sum_points_upto(tree, acc, per_window < WIN_THREADS ? per_window : WIN_THREADS);
  if (tid == 0) store_window_ext(winres + (size_t)w * 32, r);
  if (w == 0 && tid == 0) {
    totals[0] = gctr[0];
    totals[1] = gctr[1];
    totals[2] = 0;
  }
}

// ---- which columns of a phase are sparse: surviving 256-row blocks per column -----------------------------------
__global__ __launch_bounds__(SA_THREADS) void msm_s_live_blocks_kernel(const uint32_t* const* __restrict__ columns,
                                                                       const uint8_t* __restrict__ inf, uint32_t n,
                                                                       uint32_t* __restrict__ counts) {
  const uint32_t* __restrict__ scalars = columns[blockIdx.y];
  const uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x;
  uint32_t any = 0;
  if (i < n && inf[i] == 0) {
    const uint4* q = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    const uint4 lo = q[0], hi = q[1];
    any = lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;      // a fully reduced Montgomery word is zero iff the scalar is
  }
  if (__syncthreads_or(any != 0 ? 1 : 0) && threadIdx.x == 0) atomicAdd(&counts[blockIdx.y], 1u);
}

int msm_count_live_blocks(DeviceCtx& ctx, const void* const* d_columns, size_t count, const uint8_t* d_inf, size_t n, hipStream_t stream,
                          uint32_t* live_out) {
  if (count == 0 || n == 0) return HM_OK;
  if (count > 65535) return hm_fail(HM_ERR_BAD_ARG, "msm: more than 65535 commitments in one phase");
  // the buffer belongs to one counting call at a time (two threads may batch on one device): held to the end, the call
  // is synchronous and short
  std::lock_guard<std::mutex> turn(ctx.live_mu);
  uint8_t* buf = (uint8_t*)ctx.live_io.ensure(count * (sizeof(void*) + 4));
  if (!buf) return hm_fail(HM_ERR_HIP, "msm: live-block buffer allocation failed");
  uint32_t* d_counts = (uint32_t*)(buf + count * sizeof(void*));
  HM_HIP_CHECK(hipMemcpyAsync(buf, d_columns, count * sizeof(void*), hipMemcpyHostToDevice, stream));
  HM_HIP_CHECK(hipMemsetAsync(d_counts, 0, count * 4, stream));
  hipLaunchKernelGGL(msm_s_live_blocks_kernel, dim3((uint32_t)((n + SA_THREADS - 1) / SA_THREADS), (uint32_t)count), dim3(SA_THREADS), 0, stream,
                     (const uint32_t* const*)buf, d_inf, (uint32_t)n, d_counts);
  HM_HIP_CHECK(hipGetLastError());
  HM_HIP_CHECK(hipMemcpyAsync(live_out, d_counts, count * 4, hipMemcpyDeviceToHost, stream));
  HM_HIP_CHECK(hipStreamSynchronize(stream));
  return HM_OK;
}

// ---- host ----------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

bool msm_small_applies(size_t n, uint32_t c, bool single_set) {
  static const int enabled = env_int("HALO2_MI355X_SMALL_PLAN", 1);     // 0: measurement only (A/B against the general pipeline)
  return enabled && !single_set && n >= 1 && n < (1u << 19) && c >= 2 && c <= 15;
}

int msm_issue_small(DeviceCtx& ctx, int slot, const uint32_t* const* d_scalars_list, uint32_t group, const uint32_t* d_xy,
                    const uint8_t* d_inf, size_t n, uint32_t c, hipStream_t stream) {
  MsmSlot& sl = ctx.msm_slots[slot];
  if (group < 1 || group > (uint32_t)HM_MSM_GROUP) return hm_fail(HM_ERR_INTERNAL, "msm (small plan): bad group size");
  static const int env_L = env_int("HALO2_MI355X_SMALL_L", 0), env_seg = env_int("HALO2_MI355X_SMALL_SEG", 0),
                   env_H = env_int("HALO2_MI355X_SMALL_H", 0);
  const uint32_t W = (255 + c - 1) / c;
  if (W > 128) return hm_fail(HM_ERR_INTERNAL, "msm: more than 128 windows");
  // balanced widths: n_wide windows of `wide` bits, the rest of wide - 1 (wide <= c)
  const uint32_t base_bits = 255 / W, n_wide_raw = 255 - base_bits * W;
  const uint32_t wide = n_wide_raw ? base_bits + 1 : base_bits, n_wide = n_wide_raw ? n_wide_raw : W;
  const uint32_t NB = 1u << (wide - 1);
  const uint64_t pairs_max = (uint64_t)n * W;
  // bucket-range splits per window: more sort workgroups once one workgroup per window streams too long
  uint32_t H = 1;
  while (H < 8 && (uint64_t)n / H > (1u << 15) && NB / (2 * H) >= 64) H *= 2;
  if (env_H > 0) H = (uint32_t)env_H;
  while (H > 1 && NB / H < 2) H /= 2;
  const uint32_t NBh = NB / H, V = W * H;
  // Task length.  K3's chain costs L mixed additions, D then sums ~mean/L Jacobian partials per bucket (1.6x
  // the instructions each): short tasks only pay while they are needed to fill the chip.
  const double mean = (double)n / (double)NB;
  uint32_t L = (uint32_t)(mean + 4.0 * std::sqrt(mean) + 8.0);
  {
    uint64_t by_fill = pairs_max / 327680;
    if (by_fill < 16) by_fill = 16;          // measured (tools/small_tune.sh): 12 .. 16 beats 8 at every size below 2^18
    if (by_fill < L) L = (uint32_t)by_fill;
  }
  if (env_L > 0) L = (uint32_t)env_L;
  if (L < 2) L = 2;
  if (L > S_MAX_L) L = S_MAX_L;
  // D's segment length: at most one 256-lane workgroup per CU over all virtual windows -- one wave per SIMD;
  // what costs is the depth of a lane's chain (2 SEG additions + a log2(NB)-bit multiple), not the work
  uint32_t SEG = 2;
  while (SEG < NBh && (uint64_t)V * ((NBh / SEG + WIN_THREADS - 1) / WIN_THREADS) > 256) ++SEG;
  // a group multiplies the workgroups: then count WAVES with a segment to sum -- one per SIMD (1024) over the whole group
  // (never fewer than one wave per virtual window and element: 64 segments there is the coarsest useful cut)
  if (group > 1)
    while (SEG < NBh && (NBh + SEG - 1) / SEG > 64 && (uint64_t)V * group * (((NBh + SEG - 1) / SEG + 63) / 64) > 1024) ++SEG;
  if (env_seg > 0) SEG = (uint32_t)env_seg;
  if (SEG > NBh) SEG = NBh;
  const uint32_t nseg = (NBh + SEG - 1) / SEG;
  const uint32_t G1n = (nseg + WIN_THREADS - 1) / WIN_THREADS;
  // sum over all buckets of ceil(cnt / L) <= pairs / L + number of buckets
  const uint64_t T_max = pairs_max / L + (uint64_t)V * NBh + 64;

  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += align(bytes); return o; };
  const size_t o_digits = carve((size_t)((n + SA_THREADS - 1) / SA_THREADS * SA_THREADS) * W * 4);   // rows padded to whole blocks
  const size_t o_toff = carve((size_t)V * (NBh + 2) * 4);
  const size_t o_gctr = carve(16);
  const size_t o_nz = carve((size_t)V * NBh * 4);
  const size_t o_nzc = carve((size_t)V * 4);
  const size_t o_desc = carve(T_max * 16);
  const size_t o_sorted = carve(pairs_max * 4);
  const size_t o_partial = carve(T_max * PT_WORDS * 4);
  const size_t o_seg1 = carve((size_t)V * G1n * PT_WORDS * 4);
  const size_t o_blkidx = carve(((n + SA_THREADS - 1) / SA_THREADS) * 4);
  const size_t ws_stride = off;                                   // one element's arrays; the group's results follow them all
  const uint32_t res_stride = 4 + W * 32;                         // words per element: totals, then the window sums
  const size_t o_res = ws_stride * group;
  const size_t o_live = o_res + (((size_t)group * res_stride * 4 + 255) & ~(size_t)255);     // the group's survivor counters, together
  const size_t cap_before = sl.ws.cap;
  uint8_t* ws = (uint8_t*)sl.ws.ensure(o_live + (size_t)HM_MSM_GROUP * 4);
  if (sl.ws.cap != cap_before) sl.live_ptr = nullptr;             // a fresh allocation (even at the old address) holds no cleared counters
  if (!ws) return hm_fail(HM_ERR_HIP, "msm (small plan): workspace allocation failed");
  int32_t* d_digits = (int32_t*)(ws + o_digits);
  uint32_t* d_toff = (uint32_t*)(ws + o_toff);
  uint32_t* d_gctr = (uint32_t*)(ws + o_gctr);
  uint32_t* d_nz = (uint32_t*)(ws + o_nz);
  uint32_t* d_nzc = (uint32_t*)(ws + o_nzc);
  uint4* d_desc = (uint4*)(ws + o_desc);
  uint32_t* d_sorted = (uint32_t*)(ws + o_sorted);
  uint32_t* d_partial = (uint32_t*)(ws + o_partial);
  uint32_t* d_seg1 = (uint32_t*)(ws + o_seg1);
  uint32_t* d_blkidx = (uint32_t*)(ws + o_blkidx);
  uint32_t* d_live = (uint32_t*)(ws + o_live);
  uint32_t* d_tot = (uint32_t*)(ws + o_res);
  uint32_t* d_win = d_tot + 4;

  const size_t lds_sort = ((size_t)((NBh + 1 + 31) & ~31u) + 2 * S_TASK_KEYS + 32) * 4;
  if (!ctx.msm_small_attr_set) {
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_s_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(((size_t)16416 + 2 * S_TASK_KEYS + 32) * 4)));
    ctx.msm_small_attr_set = true;
  }
  int rc = msm_slot_prepare(sl);
  if (rc != HM_OK) return rc;
  hipEvent_t* ev = sl.ev;
  sl.SW = W;
  sl.c = wide;
  sl.W = W;
  sl.T_max = T_max;
  sl.d_win = d_win;
  sl.d_tot = d_tot;
  sl.balanced = true;
  sl.group = group;
  sl.res_stride = res_stride;
  sl.phase_timed = msm_phase_timing(true);
  const bool pt = sl.phase_timed;
  for (uint32_t w = 0; w < W; ++w) sl.win_bits[w] = (uint8_t)(w < n_wide ? wide : wide - 1);

  // The last chain of this slot left its block counters at zero (reduce2 clears them) unless the layout moved.  The
  // slot claims that again only once THIS chain is enqueued to its last launch: an error return in between leaves
  // live_ptr null, so the next chain clears the counters the digits kernel may already have advanced.
  const bool counters_clean = sl.live_ptr == d_live;
  sl.live_ptr = nullptr;
  HM_HIP_CHECK(hipEventRecord(ev[0], stream));
  if (!counters_clean) HM_HIP_CHECK(hipMemsetAsync(d_live, 0, (size_t)HM_MSM_GROUP * 4, stream));
  SmallGroupScalars gs;
  for (uint32_t e = 0; e < (uint32_t)HM_MSM_GROUP; ++e) gs.s[e] = d_scalars_list[e < group ? e : 0];
  hipLaunchKernelGGL(msm_s_digits_kernel, dim3((uint32_t)((n + SA_THREADS - 1) / SA_THREADS), group), dim3(SA_THREADS), 0, stream,
                     gs, d_inf, d_digits, (uint32_t)n, wide, n_wide, W, d_gctr, d_blkidx, d_live, ws_stride);
  HM_HIP_CHECK(hipGetLastError());
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[1], stream));
  hipLaunchKernelGGL(msm_s_sort_kernel, dim3(V, group), dim3(SS_THREADS), lds_sort, stream, (const int32_t*)d_digits, d_toff, d_gctr, d_desc,
                     d_sorted, d_nz, d_nzc, (uint32_t)n, NBh, H, L, (const uint32_t*)d_blkidx, (const uint32_t*)d_live, ws_stride);
  HM_HIP_CHECK(hipGetLastError());
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[2], stream));
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[5], stream));
  hipLaunchKernelGGL(msm_s_accumulate_kernel, dim3((uint32_t)((T_max + ACC_THREADS - 1) / ACC_THREADS), group), dim3(ACC_THREADS), 0, stream,
                     (const uint32_t*)d_sorted, (const uint4*)d_desc, (const uint32_t*)d_gctr, d_xy, d_partial, ws_stride);
  HM_HIP_CHECK(hipGetLastError());
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[6], stream));
  if (pt) HM_HIP_CHECK(hipEventRecord(ev[3], stream));
  hipLaunchKernelGGL(msm_s_reduce1_kernel, dim3(G1n, V, group), dim3(WIN_THREADS), 0, stream, d_partial, (const uint32_t*)d_toff,
                     (const uint32_t*)d_nz, (const uint32_t*)d_nzc, d_seg1, NBh, H, SEG, G1n, ws_stride);
  // The last kernel writes the chain's results STRAIGHT into the slot's pinned, device-mapped landing zone: no D2H copy.
  // (A hipMemcpyAsync of a group's 17.5 KiB blocked the SUBMITTING thread whenever three such copies were already
  // outstanding -- the fourth chain of a sparse phase was enqueued 6.8 ms late: tools/phase_mix.py, HALO2 trace -- the
  // general pipeline's 144-byte copies do not.)  The results are visible to the host once ev[4] has completed.
  uint32_t* d_land = nullptr;
  HM_HIP_CHECK(hipHostGetDevicePointer((void**)&d_land, sl.h_land, 0));
  hipLaunchKernelGGL(msm_s_reduce2_kernel, dim3(W, group), dim3(WIN_THREADS), 0, stream, (const uint32_t*)d_seg1, H * G1n, d_land + 4,
                     (const uint32_t*)d_gctr, d_land, ws_stride, res_stride, d_live);
  HM_HIP_CHECK(hipGetLastError());
  HM_HIP_CHECK(hipEventRecord(ev[4], stream));
  sl.live_ptr = d_live;
  return HM_OK;
}

}  // namespace hm
