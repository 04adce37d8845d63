// msm_dev.h -- device helpers shared by the MSM translation units (msm.hip: the general pipeline;
// msm_small.hip: the short launch chain for prover-sized inputs).
#pragma once
#include <hip/hip_runtime.h>

#include "g1.h"

namespace hm {

constexpr int PT_WORDS = 28;  // device Jacobian record: 27 limbs + identity flag

__device__ __forceinline__ G1Jac load_jac(const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint32_t w[PT_WORDS];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const uint4 v = q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  G1Jac r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.x.l[i] = w[i];
    r.y.l[i] = w[9 + i];
    r.z.l[i] = w[18 + i];
  }
  r.inf = w[27] != 0;
  return r;
}
__device__ __forceinline__ void store_jac(uint32_t* p, const G1Jac& a) {
  uint32_t w[PT_WORDS];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    w[i] = a.x.l[i];
    w[9 + i] = a.y.l[i];
    w[18 + i] = a.z.l[i];
  }
  w[27] = a.inf ? 1u : 0u;
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ G1Aff load_base(const uint32_t* xy, uint32_t idx) {
  const uint4* q = reinterpret_cast<const uint4*>(xy + (size_t)idx * 16);
  const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  const uint32_t wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const uint32_t wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  G1Aff r;
  r.x = fe_unpack<FqParams>(wx);
  r.y = fe_unpack<FqParams>(wy);
  return r;
}

constexpr int ACC_THREADS = 64;

typedef __attribute__((address_space(3))) void hm_lds_void;
typedef __attribute__((address_space(1))) const void hm_gbl_void;

// Start the gather of base `idx` into this lane's staging slots: four 16-byte LDS-DMA pieces
// (global_load_lds_dwordx4: per-lane source address, destination = M0 base + 16 * lane), so the
// 64-byte point of the NEXT iteration is in flight while the current mixed addition runs and costs
// no VGPRs.
__device__ __forceinline__ void stage_base_async(const uint32_t* xy, uint32_t idx, uint4 (*stage)[ACC_THREADS]) {
  const uint4* src = reinterpret_cast<const uint4*>(xy + (size_t)idx * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    __builtin_amdgcn_global_load_lds((hm_gbl_void*)(src + k), (hm_lds_void*)&stage[k][0], 16, 0, 0);
}

// One accumulation task: the serial chain of extended-Jacobian mixed additions over sorted[start .. end)
// (end > start), the accumulator living in registers; the NEXT base is gathered by LDS-DMA while the
// current addition runs.  The first point is peeled, so the loop body is one straight line with a
// single exit: a lane that meets an equal-x pair (a repeated base, or a base and its negative) leaves
// the loop and finishes its chain with the general Jacobian law.
__device__ __forceinline__ G1Jac accumulate_chain(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ xy,
                                                  uint32_t start, uint32_t end, uint4 (*stage)[ACC_THREADS], uint32_t lane) {
  // First point of the chain: a plain load; the accumulator is never the identity inside the hot loop.
  uint32_t v_cur = sorted[start];
  G1Xyzz acc;
  {
    const G1Aff q0 = load_base(xy, v_cur & 0x7fffffffu);
    acc = g1x_from_affine((v_cur >> 31) ? g1_neg_affine(q0) : q0);
  }
  uint32_t p = start + 1, v_next = 0;
  if (p < end) {
    v_cur = sorted[p];
    stage_base_async(xy, v_cur & 0x7fffffffu, stage);
    if (p + 1 < end) v_next = sorted[p + 1];
  }
  bool general = false;
  for (; p < end; ++p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the staged point (and v_next) have landed
    const uint4 a = stage[0][lane], b4 = stage[1][lane], c4 = stage[2][lane], d4 = stage[3][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... and are in registers before the slots are reused
    const bool neg = (v_cur >> 31) != 0;
    if (p + 1 < end) {
      v_cur = v_next;
      stage_base_async(xy, v_cur & 0x7fffffffu, stage);       // next point: in flight during this addition
      if (p + 2 < end) v_next = sorted[p + 2];
    }
    const uint32_t wx[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
    const uint32_t wy[8] = {c4.x, c4.y, c4.z, c4.w, d4.x, d4.y, d4.z, d4.w};
    G1Aff q;
    q.x = fe_unpack<FqParams>(wx);
    q.y = fe_unpack<FqParams>(wy);
    if (!g1x_madd_fast(acc, q, neg)) {                        // equal x: a repeated base or a base and its negative
      general = true;
      break;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA piece may outlive the loop
  G1Jac res = g1x_to_jac(acc);                                // task partials stay Jacobian downstream
  if (general) {
    // Rare: finish this lane's chain, from the point that hit the exception, with the general
    // Jacobian law (doubling, cancellation to the identity and restart from it included).
    for (; p < end; ++p) {
      const uint32_t v = sorted[p];
      res = g1_madd(res, load_base(xy, v & 0x7fffffffu), (v >> 31) != 0);
    }
  }
  return res;
}

// Workgroup-wide sum of one point per lane through an LDS tree; the result is valid in lane 0.
constexpr int WIN_THREADS = 256;
__device__ __forceinline__ G1Jac block_sum_points(uint32_t* tree, G1Jac acc) {
  const uint32_t t = threadIdx.x;
  store_jac(tree + t * PT_WORDS, acc);
  __syncthreads();
  for (uint32_t off = WIN_THREADS / 2; off > 0; off >>= 1) {
    if (t < off) {
      const G1Jac a = load_jac(tree + t * PT_WORDS), b = load_jac(tree + (t + off) * PT_WORDS);
      store_jac(tree + t * PT_WORDS, g1_add(a, b));
    }
    __syncthreads();
  }
  return load_jac(tree);
}


// cnt[bin]++ in LDS, returning the old value.  Lanes of a wave that hit the SAME counter serialise
// in the LDS atomic unit, and constant or flag columns (every scalar equal, or 0/1) put whole waves
// on one counter.  One cheap wave-uniform test catches exactly that case and replaces the wave's
// atomics by a single one; any other wave takes the plain per-lane atomic.
__device__ __forceinline__ uint32_t lds_inc(uint32_t* cnt, uint32_t bin) {
#ifndef HM_NO_AGG
  const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
  const uint64_t active = __ballot(1);
  if (__ballot(bin == v) == active) {                      // every active lane wants the same counter
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t before = (uint32_t)__popcll(active & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (before == 0) base = atomicAdd(&cnt[v], (uint32_t)__popcll(active));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    return base + before;
  }
#endif
  return atomicAdd(&cnt[bin], 1u);
}

// the same for `count` <= WIN_THREADS leading lanes (the others hold the identity): ceil(log2 count) levels
__device__ __forceinline__ G1Jac block_sum_points_upto(uint32_t* tree, G1Jac acc, uint32_t count) {
  const uint32_t t = threadIdx.x;
  uint32_t top = 1;
  while (top < count) top <<= 1;
  if (t < top) store_jac(tree + t * PT_WORDS, acc);
  __syncthreads();
  for (uint32_t off = top / 2; off > 0; off >>= 1) {
    if (t < off) {
      const G1Jac a = load_jac(tree + t * PT_WORDS), b = load_jac(tree + (t + off) * PT_WORDS);
      store_jac(tree + t * PT_WORDS, g1_add(a, b));
    }
    __syncthreads();
  }
  return load_jac(tree);
}

// a window sum -> the external Jacobian format (12 x u64 + flag word) the host fold reads
__device__ __forceinline__ void store_window_ext(uint32_t* o, const G1Jac& r) {
  uint32_t wx[8], wy[8], wz[8];
  if (r.inf) {
#pragma unroll
    for (int k = 0; k < 8; ++k) wx[k] = wy[k] = wz[k] = 0;
  } else {
    fe_to_ext(wx, r.x);
    fe_to_ext(wy, r.y);
    fe_to_ext(wz, r.z);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o[k] = wx[k];
    o[8 + k] = wy[k];
    o[16 + k] = wz[k];
  }
  o[24] = r.inf ? 1u : 0u;
}

}  // namespace hm
