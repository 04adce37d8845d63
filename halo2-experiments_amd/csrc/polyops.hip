// polyops.hip -- the Fr vector steps create_proof makes between its NTTs and commitments, device-resident
// (SURVEY.md §8f-4's neighbours; all upstream halo2_proofs at the tag pinned by /root/reference/Cargo.toml:10, reached
// from the reference through create_proof, /root/reference/src/circuits/utils.rs:40-48):
//   kate_division      arithmetic.rs kate_division: q(X) = (a(X) - a(z)) / (X - z), the quotient of every multiopen
//                      witness polynomial;  q[i] = a[i+1] + z q[i+1]
//   grand product      plonk/permutation/prover.rs and plonk/lookup/prover.rs build z(X) row by row:
//                      z[0] = start, z[i+1] = z[i] * m[i]
//   batch inversion    ff::BatchInvert on the denominators of those products (zero stays zero)
//   linear combination poly * scalar + poly, the random linear combinations of multiopen and of the h(X) pieces
// The first two are first-order linear recurrences: a lane owns B consecutive elements, the lanes of a workgroup and
// then the workgroups are joined by a log-step scan whose multiplier is uniform per step (powers of z^B) or carried
// with the value (products), and a final launch replays every chunk from its carry-in.
//
// Montgomery forms (ff29.h): external words of a are the internal form of a / 32.  Linear maps work on the raw
// words; a product of two raw values is short by a factor 32, so factors are first multiplied by the internal form
// of 32 (K32), which makes them true internal values, while the running product stays "external read as internal".
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>
#include <vector>

#include "g1.h"
#include "hm_internal.h"
#include "host_fr.h"

namespace hm {

constexpr int PO_THREADS = 256;
constexpr uint32_t PO_MAX_LANES = 65536;      // 256 workgroups of 256 lanes: one second-level workgroup joins them

struct PoFr {              // one field element in 9 x 29-bit limbs, passed by value
  uint32_t l[9];
};

// The independent scans of one proof step (the z columns of the permutation / lookup arguments, the quotients of one
// multiopen round) run as ONE launch chain: blockIdx.y selects the column.  Pointer and word tables by value are read
// with scalar loads (the index is wave-uniform); nothing here is indexed per lane or by bytes.
constexpr int PO_COLS_MAX = 16;
struct PoCols {
  const uint32_t* in[PO_COLS_MAX];
  uint32_t* out[PO_COLS_MAX];
};
struct PoPoints {
  PoFr z[PO_COLS_MAX];       // kate_division: the point of every column, true internal form
};

__device__ __forceinline__ Fr po_arg(const PoFr& a, double vb = 1.0) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i];
  HM_DECLARE(r, vb);
  return r;
}

__device__ __forceinline__ Fr po_load_raw(const uint32_t* a, uint64_t idx) {
  const uint4* src = reinterpret_cast<const uint4*>(a + idx * 8);
  const uint4 lo = src[0], hi = src[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_unpack<FrParams>(w);
}

// v: normalised limbs, value < 3r -> canonical external words
__device__ __forceinline__ void po_store_canonical(uint32_t* out, uint64_t idx, const Fr& v) {
  const Fr c = fe_canonical(v);
  uint32_t w[8];
  fe_pack(w, c);
  uint4* dst = reinterpret_cast<uint4*>(out + idx * 8);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

__device__ __forceinline__ Fr po_load9(const uint32_t* __restrict__ p, double vb) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = p[i];
  HM_DECLARE(r, vb);
  return r;
}
__device__ __forceinline__ void po_store9(uint32_t* __restrict__ p, const Fr& v) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = v.l[i];
}

__device__ __forceinline__ Fr po_pow_small(const Fr& x, uint32_t e) {
  Fr acc = fe_one<FrParams>();
  for (int bit = 31 - __clz(e | 1u); bit >= 0; --bit) {
    acc = fe_sqr(acc);
    if ((e >> bit) & 1u) acc = fe_mul(acc, x);
  }
  return acc;
}

__device__ __forceinline__ void po_lds_put(uint32_t* lds, uint32_t t, const Fr& v) {
#pragma unroll
  for (int i = 0; i < 9; ++i) lds[i * PO_THREADS + t] = v.l[i];
}
__device__ __forceinline__ Fr po_lds_get(const uint32_t* lds, uint32_t t, double vb) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = lds[i * PO_THREADS + t];
  HM_DECLARE(r, vb);
  return r;
}

// ---------------------------------------------------------------------------------------------
// kate_division.  Q[i] = sum_{j > i} a[j] z^(j-i-1) for i < n - 1.
// Lane L owns [L B, (L+1) B).  S_L = sum_{j in chunk} a[j] z^(j - lo); with y = z^B the value entering a chunk from
// above is T_L = Q[hi - 1] = sum_{L' > L} S_L' y^(L'-L-1).
// ---------------------------------------------------------------------------------------------
// inclusive suffix scan over the workgroup with the uniform multiplier y: v_t <- sum_{t' >= t} v_t' y^(t'-t)
__device__ __forceinline__ Fr po_suffix_scan_uniform(uint32_t* lds, Fr v, Fr y) {
  const uint32_t t = threadIdx.x;
  po_lds_put(lds, t, v);
  __syncthreads();
#pragma unroll 1
  for (uint32_t off = 1; off < PO_THREADS; off <<= 1) {
    Fr add = fe_zero<FrParams>();
    HM_DECLARE(add, 0.0);
    const bool has = t + off < PO_THREADS;
    if (has) add = fe_mul(po_lds_get(lds, t + off, 3.0), y);
    __syncthreads();
    if (has) {
      v = fe_reduce_small(fe_norm(fe_add(v, add)));
      po_lds_put(lds, t, v);
    }
    __syncthreads();
    y = fe_sqr(y);
  }
  return v;
}

__global__ __launch_bounds__(PO_THREADS) void fr_kate_chunks_kernel(PoCols cols, uint64_t n, PoPoints pts, uint32_t B,
                                                                    uint32_t* __restrict__ incl, uint32_t* __restrict__ wg_total) {
  __shared__ uint32_t lds[9 * PO_THREADS];
  const uint32_t t = threadIdx.x;
  const uint64_t L = (uint64_t)blockIdx.x * PO_THREADS + t;
  const uint32_t* __restrict__ a = cols.in[blockIdx.y];
  incl += (size_t)blockIdx.y * ((size_t)gridDim.x * PO_THREADS + 1) * 9;
  wg_total += (size_t)blockIdx.y * gridDim.x * 9;
  const Fr z = po_arg(pts.z[blockIdx.y]);
  const uint64_t lo = L * B, hi = lo + B < n ? lo + B : n;
  Fr acc = fe_zero<FrParams>();
  HM_DECLARE(acc, 0.0);
  for (uint64_t j = hi; j > lo; --j) acc = fe_add(fe_mul(acc, z), po_load_raw(a, j - 1));   // limbs < 2^30: a valid factor
  Fr s = fe_reduce_small(fe_norm(acc));
  const Fr y = po_pow_small(z, B);
  s = po_suffix_scan_uniform(lds, s, y);
  po_store9(incl + L * 9, s);
  if (t == 0) po_store9(wg_total + (size_t)blockIdx.x * 9, s);
}

// one workgroup: E_g = sum_{g' > g} W_g' Y^(g'-g-1), Y = z^(256 B)
__global__ __launch_bounds__(PO_THREADS) void fr_kate_join_kernel(const uint32_t* __restrict__ wg_total, uint32_t G, PoPoints pts, uint32_t B,
                                                                  uint32_t* __restrict__ carry) {
  __shared__ uint32_t lds[9 * PO_THREADS];
  const uint32_t t = threadIdx.x;
  wg_total += (size_t)blockIdx.x * G * 9;                  // one joining workgroup per column
  carry += (size_t)blockIdx.x * G * 9;
  Fr Y = po_pow_small(po_arg(pts.z[blockIdx.x]), B);
#pragma unroll 1
  for (int k = 0; k < 8; ++k) Y = fe_sqr(Y);
  Fr v = fe_zero<FrParams>();
  HM_DECLARE(v, 0.0);
  if (t + 1 < G) v = po_load9(wg_total + (size_t)(t + 1) * 9, 3.0);     // exclusive: lane g starts from W_{g+1}
  v = po_suffix_scan_uniform(lds, v, Y);
  if (t < G) po_store9(carry + (size_t)t * 9, v);
}

__global__ __launch_bounds__(PO_THREADS) void fr_kate_replay_kernel(PoCols cols, uint64_t n, PoPoints pts, uint32_t B,
                                                                    const uint32_t* __restrict__ incl, const uint32_t* __restrict__ carry) {
  const uint32_t t = threadIdx.x;
  const uint64_t L = (uint64_t)blockIdx.x * PO_THREADS + t;
  const uint64_t lo = L * B;
  if (lo + 1 >= n) return;                                  // q has n - 1 entries
  const uint64_t hi = lo + B < n ? lo + B : n;
  const uint32_t* __restrict__ a = cols.in[blockIdx.y];
  uint32_t* __restrict__ q = cols.out[blockIdx.y];
  incl += (size_t)blockIdx.y * ((size_t)gridDim.x * PO_THREADS + 1) * 9;
  carry += (size_t)blockIdx.y * gridDim.x * 9;
  const Fr z = po_arg(pts.z[blockIdx.y]);
  // T_L = (inclusive value of the next lane in this workgroup) + y^(255 - t) * E_g
  Fr T = fe_mul(po_load9(carry + (size_t)blockIdx.x * 9, 3.0), po_pow_small(po_pow_small(z, B), PO_THREADS - 1 - t));
  if (t + 1 < PO_THREADS) T = fe_add(T, po_load9(incl + (L + 1) * 9, 3.0));
  Fr cur = fe_reduce_small(fe_norm(T));                     // Q[hi - 1]
  if (hi - 1 < n - 1) po_store_canonical(q, hi - 1, cur);
  for (uint64_t i = hi - 1; i > lo; --i) {                  // Q[i - 1] = a[i] + z Q[i]
    cur = fe_reduce_small(fe_norm(fe_add(fe_mul(cur, z), po_load_raw(a, i))));
    po_store_canonical(q, i - 1, cur);
  }
}

// ---------------------------------------------------------------------------------------------
// grand product.  out[i] = start * prod_{j < i} m[j].
// ---------------------------------------------------------------------------------------------
// inclusive prefix scan of products over the workgroup (true internal values, < 2r)
__device__ __forceinline__ Fr po_prefix_scan_product(uint32_t* lds, Fr v) {
  const uint32_t t = threadIdx.x;
  po_lds_put(lds, t, v);
  __syncthreads();
#pragma unroll 1
  for (uint32_t off = 1; off < PO_THREADS; off <<= 1) {
    const bool has = t >= off;
    Fr nv = v;
    if (has) nv = fe_mul(v, po_lds_get(lds, t - off, 2.0));
    __syncthreads();
    if (has) {
      v = nv;
      po_lds_put(lds, t, v);
    }
    __syncthreads();
  }
  return v;
}

__global__ __launch_bounds__(PO_THREADS) void fr_product_chunks_kernel(PoCols cols, uint64_t n, PoFr k32_int, uint32_t B,
                                                                       uint32_t* __restrict__ incl, uint32_t* __restrict__ wg_total) {
  __shared__ uint32_t lds[9 * PO_THREADS];
  const uint32_t t = threadIdx.x;
  const uint64_t L = (uint64_t)blockIdx.x * PO_THREADS + t;
  const uint32_t* __restrict__ m = cols.in[blockIdx.y];
  incl += (size_t)blockIdx.y * ((size_t)gridDim.x * PO_THREADS + 1) * 9;
  wg_total += (size_t)blockIdx.y * gridDim.x * 9;
  const Fr k32 = po_arg(k32_int);
  const uint64_t lo = L * B, hi = lo + B < n ? lo + B : n;
  Fr p = fe_one<FrParams>();
  for (uint64_t j = lo; j < hi; ++j) p = fe_mul(p, fe_mul(po_load_raw(m, j), k32));
  p = po_prefix_scan_product(lds, p);
  po_store9(incl + L * 9, p);
  if (t == PO_THREADS - 1) po_store9(wg_total + (size_t)blockIdx.x * 9, p);
}

// How the columns of one call start.  `start_raw`: the external words of the caller's start value as 29-bit limbs
// ("external read as internal": the form the outputs are stored in), or, when `d_start` is set, the external words at that
// device address (a later group of a chained call starts from an element the previous group wrote).  chain_row == n:
// every column starts from it.  chain_row < n: column j + 1 starts from out_j[chain_row] -- upstream's `last_z`
// (permutation::keygen / prover: the z polynomial of a column set starts where the previous one stood at the last usable row).
struct PoStart {
  PoFr start_raw;
  const uint32_t* d_start;
  uint64_t chain_row;
};

__device__ __forceinline__ Fr po_start_value(const PoStart& st) {
  if (st.d_start) {
    Fr r = po_load_raw(st.d_start, 0);
    HM_DECLARE(r, 6.0);
    return r;
  }
  return po_arg(st.start_raw, 6.0);
}

// one workgroup per column: E_g = prod_{g' < g} W_g'.  Unchained: carry_g = start * E_g.  Chained: carry_g = E_g (true
// internal), and the lane that owns the chain row's workgroup also leaves P = prod_{i < chain_row} m[i] in colprod --
// fr_product_chain_kernel turns the P's of the columns before into this column's start.
__global__ __launch_bounds__(PO_THREADS) void fr_product_join_kernel(PoCols cols, uint64_t n, PoFr k32_int, uint32_t B, uint32_t G,
                                                                     const uint32_t* __restrict__ incl, const uint32_t* __restrict__ wg_total,
                                                                     PoStart st, uint32_t* __restrict__ carry, uint32_t* __restrict__ colprod) {
  __shared__ uint32_t lds[9 * PO_THREADS];
  const uint32_t t = threadIdx.x;
  incl += (size_t)blockIdx.x * ((size_t)G * PO_THREADS + 1) * 9;
  wg_total += (size_t)blockIdx.x * G * 9;
  carry += (size_t)blockIdx.x * G * 9;
  Fr v = fe_one<FrParams>();
  if (t >= 1 && t - 1 < G) v = po_load9(wg_total + (size_t)(t - 1) * 9, 2.0);   // exclusive: lane g ends at W_{g-1}
  v = po_prefix_scan_product(lds, v);
  const bool chained = st.chain_row < n;
  if (t < G) po_store9(carry + (size_t)t * 9, chained ? v : fe_mul(v, po_start_value(st)));
  if (chained) {
    const uint64_t Lu = st.chain_row / B;                   // the lane whose chunk holds the chain row
    if (t == (uint32_t)(Lu / PO_THREADS)) {
      Fr P = v;
      if (Lu % PO_THREADS) P = fe_mul(P, po_load9(incl + (Lu - 1) * 9, 2.0));
      const uint32_t* __restrict__ m = cols.in[blockIdx.x];
      const Fr k32 = po_arg(k32_int);
      for (uint64_t i = Lu * B; i < st.chain_row; ++i) P = fe_mul(P, fe_mul(po_load_raw(m, i), k32));
      po_store9(colprod + (size_t)blockIdx.x * 9, P);
    }
  }
}

// chained columns only, one workgroup per column: start_j = start * prod_{j' < j} P_j', carry_g *= start_j
__global__ __launch_bounds__(PO_THREADS) void fr_product_chain_kernel(uint32_t G, PoStart st, const uint32_t* __restrict__ colprod,
                                                                      uint32_t* __restrict__ carry) {
  const uint32_t t = threadIdx.x;
  if (t >= G) return;
  carry += (size_t)blockIdx.x * G * 9;
  Fr s = po_start_value(st);
#pragma unroll 1
  for (uint32_t j = 0; j < blockIdx.x; ++j) s = fe_mul(po_load9(colprod + (size_t)j * 9, 2.0), s);
  po_store9(carry + (size_t)t * 9, fe_mul(po_load9(carry + (size_t)t * 9, 2.0), s));
}

__global__ __launch_bounds__(PO_THREADS) void fr_product_replay_kernel(PoCols cols, uint64_t n, PoFr k32_int, uint32_t B,
                                                                       const uint32_t* __restrict__ incl, const uint32_t* __restrict__ carry) {
  const uint32_t t = threadIdx.x;
  const uint64_t L = (uint64_t)blockIdx.x * PO_THREADS + t;
  const uint64_t lo = L * B;
  if (lo >= n) return;
  const uint64_t hi = lo + B < n ? lo + B : n;
  const uint32_t* m = cols.in[blockIdx.y];
  uint32_t* out = cols.out[blockIdx.y];
  incl += (size_t)blockIdx.y * ((size_t)gridDim.x * PO_THREADS + 1) * 9;
  carry += (size_t)blockIdx.y * gridDim.x * 9;
  const Fr k32 = po_arg(k32_int);
  Fr cur = po_load9(carry + (size_t)blockIdx.x * 9, 2.0);
  if (t > 0) cur = fe_mul(cur, po_load9(incl + (L - 1) * 9, 2.0));
  for (uint64_t i = lo; i < hi; ++i) {
    const Fr f = fe_mul(po_load_raw(m, i), k32);             // read before the store: out may be m itself
    po_store_canonical(out, i, cur);
    cur = fe_mul(cur, f);
  }
}

// ---------------------------------------------------------------------------------------------
// batch inversion, in place; zero stays zero.  A lane owns INV_B consecutive elements (its prefix products live in
// registers / scratch); the 64 lanes of a wave share one Fermat inversion of the product of their chunks.  Cost per
// element: 5 products + ~400 / INV_B for the inversion and the wave scans, so large arrays take 32 elements per lane
// and small ones 8 (more lanes, same ~0.2 ms latency of one inversion chain).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ Fr po_shfl(const Fr& v, int src_lane) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__shfl((int)v.l[i], src_lane, 64);
#ifdef HM_BOUNDS
  r.vb = v.vb; r.lb = v.lb; r.tb = v.tb;
#endif
  return r;
}

// x^(r - 2), x a true internal value.  (Measured alternative, not kept: Kaliski's almost-Montgomery-inverse on 8 x 32-bit words --
// uniform over the wave here, since every lane inverts the same total -- plus one product with a 2^-k table: the z columns of the
// k = 11 / 17 / 18 proofs 0.25 / 0.28 / 0.67 -> 0.21 / 0.24 / 0.61 ms; a hundred lines and a 9 KB table for 1 % of the smallest replay.)
__device__ __forceinline__ Fr po_invert(const Fr& x) {
  const uint64_t e[4] = {0x43e1f593efffffffull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  Fr acc = x;                                    // bit 253 of r - 2 is its top bit
#pragma unroll 1
  for (int bit = 252; bit >= 0; --bit) {
    acc = fe_sqr(acc);
    if ((e[bit >> 6] >> (bit & 63)) & 1ull) acc = fe_mul(acc, x);
  }
  return acc;
}

template <int INV_B>
__global__ __launch_bounds__(PO_THREADS) void fr_batch_invert_kernel(uint32_t* __restrict__ v, uint64_t n, PoFr k32_int) {
  const uint64_t L = (uint64_t)blockIdx.x * PO_THREADS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const Fr k32 = po_arg(k32_int);
  const uint64_t lo = L * INV_B;
  Fr val[INV_B], pre[INV_B];                     // val: true internal values (1 in place of a zero or a missing element)
  uint64_t zero_mask = 0;                        // INV_B <= 64
  Fr run = fe_one<FrParams>();
#pragma unroll
  for (int k = 0; k < INV_B; ++k) {
    val[k] = fe_one<FrParams>();
    if (lo + k < n) {
      const Fr raw = po_load_raw(v, lo + k);
      uint32_t any = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) any |= raw.l[i];
      if (any == 0) zero_mask |= 1ull << k;
      else val[k] = fe_mul(raw, k32);
    }
    pre[k] = run;                                // product of the lane's elements before k
    run = fe_mul(run, val[k]);
  }
  // wave: inclusive prefix and suffix products of the lane totals
  Fr pfx = run, sfx = run;
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    const Fr a = po_shfl(pfx, lane - off < 0 ? lane : lane - off);
    const Fr b = po_shfl(sfx, lane + off > 63 ? lane : lane + off);
    if (lane >= off) pfx = fe_mul(pfx, a);
    if (lane + off <= 63) sfx = fe_mul(sfx, b);
  }
  const Fr total = po_shfl(pfx, 63);
  const Fr inv_total = po_invert(total);
  // 1 / run_lane = inv_total * (product of the lanes before) * (product of the lanes after)
  Fr before = po_shfl(pfx, lane == 0 ? 0 : lane - 1), after = po_shfl(sfx, lane == 63 ? 63 : lane + 1);
  Fr inv_run = inv_total;
  if (lane > 0) inv_run = fe_mul(inv_run, before);
  if (lane < 63) inv_run = fe_mul(inv_run, after);
  const Fr int2ext = fe_const<FrParams>(FrParams::INT2EXT);
#pragma unroll
  for (int k = INV_B - 1; k >= 0; --k) {
    if (lo + k < n) {
      const Fr inv_k = fe_mul(inv_run, pre[k]);              // 1 / val[k]
      if (zero_mask & (1ull << k)) {
        uint4* dst = reinterpret_cast<uint4*>(v + (lo + k) * 8);
        dst[0] = make_uint4(0, 0, 0, 0);
        dst[1] = make_uint4(0, 0, 0, 0);
      } else {
        po_store_canonical(v, lo + k, fe_mul(inv_k, int2ext));
      }
    }
    inv_run = fe_mul(inv_run, val[k]);
  }
}

// ---------------------------------------------------------------------------------------------
// linear combination: out[i] = sum_j c_j * poly_j[i], up to LC_MAX terms per launch
// ---------------------------------------------------------------------------------------------
constexpr int LC_MAX = 24;
struct LcArgs {
  const uint32_t* poly[LC_MAX];
  PoFr c[LC_MAX];                    // true internal form of the coefficients
};

__global__ __launch_bounds__(PO_THREADS) void fr_lincomb_kernel(LcArgs args, uint32_t count, uint64_t n, uint32_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * PO_THREADS + threadIdx.x;
  if (i >= n) return;
  Fr acc = fe_zero<FrParams>();
  HM_DECLARE(acc, 0.0);
  for (uint32_t j = 0; j < count; ++j) {
    const Fr term = fe_mul(po_load_raw(args.poly[j], i), po_arg(args.c[j]));   // < 2r, normalised
    acc = fe_add(acc, term);
    if ((j & 3u) == 3u) acc = fe_reduce_small(fe_norm(acc));                    // every four terms (limbs < 5 * 2^29): back below 3r
  }
  po_store_canonical(out, i, fe_reduce_small(fe_norm(acc)));
}

// ---------------------------------------------------------------------------------------------
// a[i] *= pattern[i mod P]: EvaluationDomain::divide_by_vanishing_poly (the inverse vanishing polynomial takes only
// 2^(extended_k - k) values on the extended coset), P a power of two <= PP_MAX, the pattern by value
// ---------------------------------------------------------------------------------------------
constexpr int PP_MAX = 64;
struct PeriodicPattern {
  PoFr v[PP_MAX];                    // true internal form
};
__global__ __launch_bounds__(PO_THREADS) void fr_mul_periodic_kernel(uint32_t* a, uint64_t n, PeriodicPattern pat, uint32_t mask) {
  __shared__ uint32_t s_pat[PP_MAX * 9];
  for (uint32_t k = threadIdx.x; k < (mask + 1) * 9; k += PO_THREADS) s_pat[k] = pat.v[k / 9].l[k % 9];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * PO_THREADS + threadIdx.x;
  if (i >= n) return;
  Fr c;
  const uint32_t* p = s_pat + ((uint32_t)i & mask) * 9;
#pragma unroll
  for (int k = 0; k < 9; ++k) c.l[k] = p[k];
  HM_DECLARE(c, 1.0);
  po_store_canonical(a, i, fe_mul(po_load_raw(a, i), c));
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static void po_internal(const uint64_t x_ext[4], PoFr& out) { host::fr_to_internal9(host::fr_load(x_ext), out.l); }
static void po_raw(const uint64_t x_ext[4], PoFr& out) {       // the external words as 29-bit limbs, unconverted
  const uint32_t* w = reinterpret_cast<const uint32_t*>(x_ext);
  uint32_t ww[8];
  std::memcpy(ww, w, 32);
  const Fr r = fe_unpack<FrParams>(ww);
  for (int i = 0; i < 9; ++i) out.l[i] = r.l[i];
}

struct PoPlan {
  uint32_t B, G;
  uint64_t lanes;
};
static PoPlan po_plan(uint64_t n) {
  PoPlan p;
  p.B = (uint32_t)((n + PO_MAX_LANES - 1) / PO_MAX_LANES);
  if (p.B < 4) p.B = n >= 4 ? 4 : 1;
  p.lanes = (n + p.B - 1) / p.B;
  p.G = (uint32_t)((p.lanes + PO_THREADS - 1) / PO_THREADS);
  return p;
}

// scratch of `cols` scans: per column the inclusive values of G * 256 lanes (+ one guard lane), then per column G workgroup
// totals, then G carries, then one column product each (chained grand products)
struct PoScratch {
  uint32_t *incl, *wg_total, *carry, *colprod;
};
static bool po_scratch(AuxSlot* slot, const PoPlan& p, size_t cols, PoScratch* out) {
  const size_t lanes = (size_t)p.G * PO_THREADS + 1;
  const size_t need = cols * (lanes + 2 * (size_t)p.G + 1) * 36;
  const size_t keep = (size_t)64 * 15 * 28 * 4;             // never shrink below the fixed-base table (see poly.hip)
  uint8_t* buf = (uint8_t*)slot->table.ensure(need > keep ? need : keep);
  if (!buf) return false;
  out->incl = (uint32_t*)buf;
  out->wg_total = out->incl + cols * lanes * 9;
  out->carry = out->wg_total + cols * (size_t)p.G * 9;
  out->colprod = out->carry + cols * (size_t)p.G * 9;
  return true;
}

// `count` independent divisions of n-coefficient polynomials, each by its own point, PO_COLS_MAX per launch chain
int fr_kate_division_batch_run(DeviceCtx& ctx, const void* const* d_a, uint64_t n, const uint64_t* z_ext, void* const* d_q, size_t count,
                               hipStream_t stream) {
  if (n < 2 || count == 0) return HM_OK;                     // a constant has the empty quotient
  const PoPlan p = po_plan(n);
  if (p.G > PO_THREADS) return hm_fail(HM_ERR_INTERNAL, "kate_division: plan exceeds one joining workgroup");
  AuxSlot* slot = aux_acquire(ctx, stream);
  if (!slot) return HM_ERR_HIP;
  PoScratch sc;
  const size_t per = count < (size_t)PO_COLS_MAX ? count : (size_t)PO_COLS_MAX;
  if (!po_scratch(slot, p, per, &sc)) return hm_fail(HM_ERR_HIP, "kate_division: scratch allocation failed");
  for (size_t first = 0; first < count; first += per) {
    const uint32_t c = (uint32_t)(count - first < per ? count - first : per);
    PoCols cols;
    PoPoints pts;
    std::memset(&cols, 0, sizeof cols);
    std::memset(&pts, 0, sizeof pts);
    for (uint32_t j = 0; j < c; ++j) {
      cols.in[j] = (const uint32_t*)d_a[first + j];
      cols.out[j] = (uint32_t*)d_q[first + j];
      po_internal(z_ext + (first + j) * 4, pts.z[j]);
    }
    hipLaunchKernelGGL(fr_kate_chunks_kernel, dim3(p.G, c), dim3(PO_THREADS), 0, stream, cols, n, pts, p.B, sc.incl, sc.wg_total);
    hipLaunchKernelGGL(fr_kate_join_kernel, dim3(c), dim3(PO_THREADS), 0, stream, (const uint32_t*)sc.wg_total, p.G, pts, p.B, sc.carry);
    hipLaunchKernelGGL(fr_kate_replay_kernel, dim3(p.G, c), dim3(PO_THREADS), 0, stream, cols, n, pts, p.B, (const uint32_t*)sc.incl,
                       (const uint32_t*)sc.carry);
  }
  HM_HIP_CHECK(hipGetLastError());
  return aux_release(ctx, slot, stream);
}

int fr_kate_division_run(DeviceCtx& ctx, const uint32_t* d_a, uint64_t n, const uint64_t z_ext[4], uint32_t* d_q, hipStream_t stream) {
  const void* a = d_a;
  void* q = d_q;
  return fr_kate_division_batch_run(ctx, &a, n, z_ext, &q, 1, stream);
}

// `count` running products over n rows; chain_row >= n: every column starts from `start`; chain_row < n: column j + 1 starts
// from out_j[chain_row] (PoStart).  PO_COLS_MAX columns per launch chain; a later group of a chained call reads its start
// from the element the group before wrote (stream order).
int fr_grand_product_batch_run(DeviceCtx& ctx, const void* const* d_m, uint64_t n, const uint64_t start_ext[4], uint64_t chain_row,
                               void* const* d_out, size_t count, hipStream_t stream) {
  if (n == 0 || count == 0) return HM_OK;
  const PoPlan p = po_plan(n);
  if (p.G > PO_THREADS) return hm_fail(HM_ERR_INTERNAL, "grand_product: plan exceeds one joining workgroup");
  AuxSlot* slot = aux_acquire(ctx, stream);
  if (!slot) return HM_ERR_HIP;
  PoScratch sc;
  const size_t per = count < (size_t)PO_COLS_MAX ? count : (size_t)PO_COLS_MAX;
  if (!po_scratch(slot, p, per, &sc)) return hm_fail(HM_ERR_HIP, "grand_product: scratch allocation failed");
  PoFr k32;
  host::fr_to_internal9(host::FR_32, k32.l);
  const bool chained = chain_row < n;
  PoStart st;
  std::memset(&st, 0, sizeof st);
  po_raw(start_ext, st.start_raw);
  st.chain_row = chained ? chain_row : n;
  for (size_t first = 0; first < count; first += per) {
    const uint32_t c = (uint32_t)(count - first < per ? count - first : per);
    PoCols cols;
    std::memset(&cols, 0, sizeof cols);
    for (uint32_t j = 0; j < c; ++j) {
      cols.in[j] = (const uint32_t*)d_m[first + j];
      cols.out[j] = (uint32_t*)d_out[first + j];
    }
    st.d_start = chained && first ? (const uint32_t*)d_out[first - 1] + chain_row * 8 : nullptr;
    hipLaunchKernelGGL(fr_product_chunks_kernel, dim3(p.G, c), dim3(PO_THREADS), 0, stream, cols, n, k32, p.B, sc.incl, sc.wg_total);
    hipLaunchKernelGGL(fr_product_join_kernel, dim3(c), dim3(PO_THREADS), 0, stream, cols, n, k32, p.B, p.G, (const uint32_t*)sc.incl,
                       (const uint32_t*)sc.wg_total, st, sc.carry, sc.colprod);
    if (chained)
      hipLaunchKernelGGL(fr_product_chain_kernel, dim3(c), dim3(PO_THREADS), 0, stream, p.G, st, (const uint32_t*)sc.colprod, sc.carry);
    hipLaunchKernelGGL(fr_product_replay_kernel, dim3(p.G, c), dim3(PO_THREADS), 0, stream, cols, n, k32, p.B, (const uint32_t*)sc.incl,
                       (const uint32_t*)sc.carry);
  }
  HM_HIP_CHECK(hipGetLastError());
  return aux_release(ctx, slot, stream);
}

int fr_grand_product_run(DeviceCtx& ctx, const uint32_t* d_m, uint64_t n, const uint64_t start_ext[4], uint32_t* d_out, hipStream_t stream) {
  const void* m = d_m;
  void* o = d_out;
  return fr_grand_product_batch_run(ctx, &m, n, start_ext, n, &o, 1, stream);
}

int fr_batch_invert_run(uint32_t* d_v, uint64_t n, hipStream_t stream) {
  if (n == 0) return HM_OK;
  PoFr k32;
  host::fr_to_internal9(host::FR_32, k32.l);
  // Elements per lane.  A wave's time is its chain -- 5 products per element, ~12 for the wave scans, ~380 for the shared Fermat
  // inversion -- and a SIMD runs one wave at full speed, two at half: what counts is how many ROUNDS of 1 024 waves (256 CUs x 4
  // SIMDs) the launch needs.  The 2.9 M denominators of the k = 18 proof took 1 408 waves at 32 per lane -- two rounds on a
  // third of the SIMDs, 0.565 ms; at 48 per lane they are 939 waves, one round, 0.34 ms (A single workgroup-wide inversion
  // instead -- one wave inverting for sixteen -- was measured too: less work, but the other fifteen waves wait out the same
  // chain and the launch has too few workgroups to fill the gaps: 0.8 -> 1.17 ms for the z columns of that proof.)
  const uint64_t one_round = (uint64_t)1024 * 64;                         // lanes of one round
  auto launch = [&](auto tag) {
    constexpr int B = decltype(tag)::value;
    const uint64_t lanes = (n + B - 1) / B;
    hipLaunchKernelGGL(fr_batch_invert_kernel<B>, dim3((uint32_t)((lanes + PO_THREADS - 1) / PO_THREADS)), dim3(PO_THREADS), 0, stream, d_v, n, k32);
  };
  if (n < ((uint64_t)1 << 21)) launch(std::integral_constant<int, 8>{});            // small: more lanes, the same ~0.2 ms chain
  else if (n <= one_round * 32) launch(std::integral_constant<int, 32>{});
  else if (n <= one_round * 48) launch(std::integral_constant<int, 48>{});
  else if (n <= one_round * 64) launch(std::integral_constant<int, 64>{});
  else launch(std::integral_constant<int, 32>{});                                   // many rounds either way: the least scratch per lane
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int fr_linear_combination_run(const void* const* d_polys, const uint64_t* coeffs_ext, size_t count, uint64_t n, uint32_t* d_out,
                              hipStream_t stream) {
  if (n == 0) return HM_OK;
  const uint64_t one_ext[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};   // R mod r
  const uint32_t blocks = (uint32_t)((n + PO_THREADS - 1) / PO_THREADS);
  if (count == 0) {
    HM_HIP_CHECK(hipMemsetAsync(d_out, 0, n * 32, stream));
    return HM_OK;
  }
  // terms that read d_out itself must all be consumed by the FIRST launch (later launches see the running sum there)
  std::vector<size_t> order;
  order.reserve(count);
  for (size_t j = 0; j < count; ++j)
    if (d_polys[j] == (const void*)d_out) order.push_back(j);
  if (order.size() > (size_t)LC_MAX)
    return hm_fail(HM_ERR_BAD_ARG, "linear_combination: the output appears among the inputs more than 24 times");
  for (size_t j = 0; j < count; ++j)
    if (d_polys[j] != (const void*)d_out) order.push_back(j);
  size_t done = 0;
  bool first = true;
  while (done < count) {
    LcArgs args;
    std::memset(&args, 0, sizeof args);
    uint32_t k = 0;
    if (!first) {                                            // the running sum re-enters with coefficient 1
      args.poly[0] = d_out;
      po_internal(one_ext, args.c[0]);
      k = 1;
    }
    while (k < (uint32_t)LC_MAX && done < count) {
      args.poly[k] = (const uint32_t*)d_polys[order[done]];
      po_internal(coeffs_ext + order[done] * 4, args.c[k]);
      ++k;
      ++done;
    }
    hipLaunchKernelGGL(fr_lincomb_kernel, dim3(blocks), dim3(PO_THREADS), 0, stream, args, k, n, d_out);
    first = false;
  }
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int fr_mul_periodic_run(uint32_t* d_a, uint64_t n, const uint64_t* pattern_ext, uint32_t period, hipStream_t stream) {
  if (n == 0) return HM_OK;
  if (period == 0 || period > (uint32_t)PP_MAX || (period & (period - 1)))
    return hm_fail(HM_ERR_BAD_ARG, "mul_periodic: the period must be a power of two <= 64");
  PeriodicPattern pat;
  std::memset(&pat, 0, sizeof pat);
  for (uint32_t k = 0; k < period; ++k) po_internal(pattern_ext + (size_t)k * 4, pat.v[k]);
  hipLaunchKernelGGL(fr_mul_periodic_kernel, dim3((uint32_t)((n + PO_THREADS - 1) / PO_THREADS)), dim3(PO_THREADS), 0, stream, d_a, n, pat, period - 1);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

}  // namespace hm
