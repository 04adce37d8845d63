// ntt.hip -- radix-2 NTT over BN256 Fr for gfx950 (replaces halo2_proofs::arithmetic::best_fft,
// natural order in, natural order out, unscaled, a'[j] = sum_i a[i] omega^(ij)).
//
// Structure (MI355X-first, not the reference's recursion): the 2^log_n transform is factored
// into <= 3 digit passes  n = R0*R1*R2  (Cooley-Tukey "four-step" applied twice).  One workgroup
// owns a tile of 2^LOG_TILE elements = R (the pass's digit, a full R-point sub-transform done
// entirely in LDS with radix-2 DIT stages) x C adjacent columns (so every HBM access is a
// C*32-byte contiguous segment).  The first pass reads `a` and writes the scratch buffer, the
// middle pass works in place in scratch, the last pass reads scratch rows and writes `a` in
// natural order (the digit reversal is folded into its store), canonicalising on the way out.
// HBM traffic = passes * 64 B/element; there is no separate bit-reversal pass and no n/2-entry
// twiddle table (the reference's is 256 MiB at 2^24): inter-pass twiddles come from two
// sqrt(n)-sized tables (one extra product), stage twiddles from a <= 512-entry table.
//
// Field layer: ff29.h.  The NTT is linear in the data, so the reference's radix-2^256
// Montgomery words are simply reinterpreted as internal-form values (of x/32) -- no conversion
// on the way in or out; only omega and the twiddles are converted (once per (omega, log_n)).
#include <hip/hip_runtime.h>

#include "g1.h"
#include "hm_internal.h"
#include "host_fr.h"

namespace hm {

#ifndef HM_NTT_THREADS
#define HM_NTT_THREADS 512
#endif
constexpr int NTT_THREADS = HM_NTT_THREADS;

// ---------------------------------------------------------------------------------------------
// twiddle tables
// ---------------------------------------------------------------------------------------------
// out[j] = base^(j << shift), internal form, normalised, value < 2r.   9 x u32 per entry.
struct FrWords {       // one external field element, passed to kernels by value
  uint32_t w[8];
};
struct FrWords3 {
  uint32_t w[24];
};

__global__ void ntt_pow_table_kernel(FrWords omega_ext, uint32_t* __restrict__ out, uint32_t count, uint32_t shift) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = omega_ext.w[i];
  const Fr base = fe_from_ext<FrParams>(w);
  const uint64_t e = (uint64_t)j << shift;
  Fr acc = fe_one<FrParams>();
  for (int bit = 40; bit >= 0; --bit) {
    acc = fe_sqr(acc);
    if ((e >> bit) & 1) acc = fe_mul(acc, base);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) out[(size_t)j * 9 + i] = acc.l[i];
}

__device__ __forceinline__ Fr load_tw(const uint32_t* __restrict__ tab, uint32_t idx) {
  Fr r;
  const uint32_t* p = tab + (size_t)idx * 9;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = p[i];
  HM_DECLARE(r, 2.0);
  return r;
}

// ---------------------------------------------------------------------------------------------
// LDS tile helpers: structure-of-arrays, plane l holds limb l of every element of the tile
// ---------------------------------------------------------------------------------------------
// element index -> word index inside a plane.  XOR-ing bits 3..4 with bits 6..7 keeps the first
// register round (8 consecutive digit positions per lane, i.e. a 64-word lane stride) off the
// same banks without growing the tile (a +8-per-64 pad would push two tiles past 160 KiB).
__device__ __forceinline__ uint32_t lds_swz(uint32_t e) { return e ^ (((e >> 6) & 3u) << 3); }

template <int LOG_TILE>
__device__ __forceinline__ Fr lds_load(const uint32_t* lds, uint32_t e) {
  Fr r;
  const uint32_t a = lds_swz(e);
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = lds[(i << LOG_TILE) + a];
  return r;
}
template <int LOG_TILE>
__device__ __forceinline__ void lds_store(uint32_t* lds, uint32_t e, const Fr& a) {
  const uint32_t ad = lds_swz(e);
#pragma unroll
  for (int i = 0; i < 9; ++i) lds[(i << LOG_TILE) + ad] = a.l[i];
}

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t bits) {
  return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

struct NttCosetTables {          // pointer table by value (indexed by blockIdx.z: scalar loads)
  const uint32_t* table[16];
  uint32_t count;
};

struct NttPassParams {
  uint32_t log_n;        // transform size
  uint32_t s;            // log2 of this pass's digit R
  uint32_t log_stride;   // log2 of the element stride between consecutive digit values (non-last passes)
  uint32_t log_c;        // log2 of the columns (non-last) / rows (last) per tile
  uint32_t last;         // 1: stride-1 pass with the digit-reversing, canonicalising store
  uint32_t log_r0;       // last pass of a 3-pass plan: log2 R0 (else 0)
  uint32_t log_rows;     // last pass: log2 of the number of rows = log_n - s
  uint32_t log_lb;       // split point of the two-level inter-pass twiddle table
  uint32_t fin_mode;     // last pass: 0 = canonicalise only, 1 = multiply by fin[0], 2 = output i *= fin[i % 3]
  uint32_t has_coset;    // first pass: multiply a[i] by coset[i % 3] on load
  uint32_t direct_tw;    // non-last pass: tw_lo holds omega_m^e for every e < m (no forming product)
  uint32_t log_z;        // first pass of an extending transform: the input holds only the first n >> log_z
                         // coefficients (the rest are zero by definition); 0 = ordinary transform
  uint32_t has_table;    // first pass: multiply a[i] by in_scale[i] on load (the transform on a coset shift * <omega>: in_scale
                         // holds 32 * shift^i as external words, so that the product of two external words is the external
                         // word of a[i] * shift^i -- the radix is 2^261)
  // fused constants, 9-limb internal form, BY VALUE (a call never shares a constants buffer with another
  // call on another stream): coset[3] for the first pass, fin[3] for the last
  uint32_t coset[27];
  uint32_t fin[27];
};

// everything a register round needs
struct NttTileCtx {
  uint32_t tid, s, log_c, cmask, tile_elems, i0;
  uint64_t base, dest_lo0;
};

// One register round of the R-point DIT: every lane takes groups of 2^RB positions
// p = base + j * 2^q0 (same column), runs stages q0 .. q0+RB-1 on them in registers and writes them
// back normalised: one LDS round trip per <= 3 stages instead of one per stage.  (The pass epilogue
// stays a separate one-element-per-iteration loop: folding it into the last round multiplies the
// number of inlined ~230-instruction products and pushes the kernel out of the instruction cache.)
template <int LOG_TILE, int RB>
__device__ __forceinline__ void ntt_round(uint32_t* lds, const NttTileCtx& cx, uint32_t q0,
                                          const uint32_t* __restrict__ stage_tw) {
  constexpr int GE = 1 << RB;
  const uint32_t s = cx.s, log_c = cx.log_c;
  const uint32_t ngroups = cx.tile_elems >> RB;
  for (uint32_t gid = cx.tid; gid < ngroups; gid += NTT_THREADS) {
    const uint32_t c = gid & cx.cmask, u = gid >> log_c;
    const uint32_t base_lo = u & ((1u << q0) - 1u);
    const uint32_t pbase = ((u >> q0) << (q0 + RB)) | base_lo;
    Fr x[GE];
#pragma unroll
    for (int j = 0; j < GE; ++j) x[j] = lds_load<LOG_TILE>(lds, ((pbase + ((uint32_t)j << q0)) << log_c) + c);
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const uint32_t q = q0 + t;
#pragma unroll
      for (int m = 0; m < (1 << t); ++m) {
        if (q == 0) {   // stage 0: twiddle 1; b is a raw input (< 2^256) or a product
#pragma unroll
          for (int h = 0; h < (1 << (RB - t - 1)); ++h) {
            const int j = m + (h << (t + 1));
            const Fr a = x[j], b = x[j + (1 << t)];
            x[j] = fe_add(a, b);
            x[j + (1 << t)] = fe_sub<6, 29>(a, b);
          }
        } else if (m == 0 && q0 == 0) {   // first round, later stage, twiddle omega^0 = 1: no product
#pragma unroll
          for (int h = 0; h < (1 << (RB - t - 1)); ++h) {
            const int j = m + (h << (t + 1));
            const Fr a = x[j];
            const Fr tt = fe_norm(x[j + (1 << t)]);       // a lazy stage-0 output, value < 12r
            x[j] = fe_add(a, tt);
            x[j + (1 << t)] = fe_sub<13, 29>(a, tt);
          }
        } else {
          const Fr w = load_tw(stage_tw, (base_lo + ((uint32_t)m << q0)) << (s - 1 - q));
#pragma unroll
          for (int h = 0; h < (1 << (RB - t - 1)); ++h) {
            const int j = m + (h << (t + 1));
            const Fr a = x[j];
            const Fr tt = fe_mul(x[j + (1 << t)], w);
            x[j] = fe_add(a, tt);
            x[j + (1 << t)] = fe_sub<3, 29>(a, tt);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < GE; ++j) lds_store<LOG_TILE>(lds, ((pbase + ((uint32_t)j << q0)) << log_c) + c, fe_norm(x[j]));
  }
}

// One digit pass.  in/out: n x 8 u32 (external words).  stage_tw: omega_R^j, j < R/2.
// tw_lo/tw_hi: omega_n^j (j < 2^log_lb) and omega_n^(j << log_lb).
// FUSED = false: the plain best_fft pass (no constants multiplied in): kept as its own instantiation so that the
// fused forms' extra state costs it nothing (registers, the constants' LDS copy and barrier).
// TABLE (implies FUSED): the first pass multiplies every input element by its entry of in_scale (a third instantiation, so
// that the other two keep their register allocation: with the table words the four-deep preload spills).
template <int LOG_TILE, bool FUSED, bool TABLE = false>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(const uint32_t* in, uint32_t* out,
                                                               NttPassParams pp, const uint32_t* __restrict__ stage_tw,
                                                               const uint32_t* __restrict__ tw_lo,
                                                               const uint32_t* __restrict__ tw_hi,
                                                               const uint32_t* __restrict__ in_scale, NttCosetTables zc) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t s_const[FUSED ? 54 : 1];   // [0, 27): coset pattern, [27, 54): final constants (indexed per element)
  const uint32_t tid = threadIdx.x;
  if (FUSED && (pp.has_coset | pp.fin_mode)) {
    if (tid == 0) {
#pragma unroll
      for (int i = 0; i < 27; ++i) {
        s_const[i] = pp.coset[i];
        s_const[27 + i] = pp.fin[i];
      }
    }
    __syncthreads();
  }
  const uint32_t s = pp.s, log_c = pp.log_c;
  const uint32_t tile_elems = 1u << (s + log_c);  // <= 2^LOG_TILE
  const uint32_t cmask = (1u << log_c) - 1u;
  const uint64_t tile = blockIdx.x;
  // batched transforms: blockIdx.y selects one of `gridDim.y` back-to-back arrays of n elements
  in += ((size_t)blockIdx.y << (pp.log_n - pp.log_z)) * 8;   // an extending first pass reads compact input arrays
  uint32_t out_index = blockIdx.y;
  if (TABLE && zc.count) {                                    // several cosets of the same inputs: blockIdx.z = coset
    out_index = blockIdx.y * zc.count + blockIdx.z;
    in_scale = zc.table[blockIdx.z];
  }
  out += ((size_t)out_index << pp.log_n) * 8;

  // ---- tile -> global index mapping ---------------------------------------------------------
  // non-last: element (d, c) lives at  o*m + d*stride + i0 + c        (m = R*stride)
  // last    : element (d, c) lives at  row(c)*R + d, row(c) = digit-reversed (dest_lo0 + c)
  uint64_t base = 0;      // non-last: o*m + i0
  uint32_t i0 = 0;        // non-last: first column (for the twiddle exponent)
  uint64_t dest_lo0 = 0;  // last: first destination low index
  if (!pp.last) {
    const uint32_t log_tiles_per_block = pp.log_stride - log_c;
    const uint64_t o = tile >> log_tiles_per_block;
    i0 = (uint32_t)(tile & ((1ull << log_tiles_per_block) - 1ull)) << log_c;
    base = (o << (s + pp.log_stride)) + i0;
  } else {
    dest_lo0 = tile << log_c;
  }
  auto row_of = [&](uint32_t c) -> uint64_t {  // last pass only
    const uint64_t dl = dest_lo0 + c;
    if (pp.log_r0 == 0) return dl;                       // 1- or 2-pass plan: row = k0 = dest_lo
    const uint64_t k0 = dl & ((1ull << pp.log_r0) - 1ull);  // dest_lo = k0 + R0*k1
    const uint64_t k1 = dl >> pp.log_r0;
    return (k0 << (pp.log_rows - pp.log_r0)) + k1;       // row = k0*R1 + k1
  };

  // ---- load: unpack to 29-bit limbs, scatter to the bit-reversed digit position -------------
  // Extending first pass (log_z > 0, EvaluationDomain::coeff_to_extended): only digit values
  // d < R >> log_z carry coefficients, the rest of the zero-padded array is never read.  Their
  // bit-reversed positions are multiples of 2^log_z, and DIT stages 0 .. log_z-1 pair each of them with
  // zeros only -- (a, 0) -> (a, a), no product -- so the value is stored to the 2^log_z positions of its
  // group at once and the rounds start at stage log_z.
  // The usual tile (2^LOG_TILE elements, nothing skipped) is NTT_THREADS x 4: all four of a lane's elements are requested
  // before the first is unpacked.  (As a plain loop the compiler waits for each 32-byte load before it issues the next: four
  // serial HBM round trips per tile, hidden only by the other workgroup of the CU.)
  auto where = [&](uint32_t e, uint32_t& d, uint32_t& c) -> uint64_t {
    if (!pp.last) {
      c = e & cmask;
      d = e >> log_c;
      return base + ((uint64_t)d << pp.log_stride) + c;
    }
    d = e & ((1u << s) - 1u);
    c = e >> s;
    return (row_of(c) << s) + d;
  };
  auto place = [&](uint64_t g, uint32_t d, uint32_t c, const uint4& lo, const uint4& hi, const uint4& tl, const uint4& th) {
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    Fr x = fe_unpack<FrParams>(w);
    if (TABLE && pp.has_table) {
      const uint32_t tw8[8] = {tl.x, tl.y, tl.z, tl.w, th.x, th.y, th.z, th.w};
      Fr ts = fe_unpack<FrParams>(tw8);
      HM_DECLARE(ts, 6.0);
      x = fe_mul(x, ts);
    }
    if (FUSED && pp.has_coset) {
      Fr cz;
      const uint32_t* cp = s_const + (uint32_t)(g % 3) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) cz.l[i] = cp[i];
      HM_DECLARE(cz, 2.0);
      x = fe_mul(x, cz);
    }
    if (pp.log_z == 0) {
      lds_store<LOG_TILE>(lds, (bitrev(d, s) << log_c) + c, x);
    } else {
      if (!((FUSED && pp.has_coset) || (TABLE && pp.has_table))) x = fe_reduce_small(x);   // raw 256-bit words: bring below 3r like a product
      const uint32_t p0 = bitrev(d, s);                        // low log_z bits are zero
      for (uint32_t t = 0; t < (1u << pp.log_z); ++t) lds_store<LOG_TILE>(lds, ((p0 + t) << log_c) + c, x);
    }
  };
  const uint32_t n_load = tile_elems >> pp.log_z;
  if (TABLE && pp.has_table && n_load == 4u * NTT_THREADS) {
    // two rounds of two elements: with the table words a four-deep preload does not fit the 128 registers of this occupancy
    for (int half = 0; half < 2; ++half) {
      uint32_t d[2], c[2];
      uint64_t g[2];
      uint4 lo[2], hi[2], tl[2], th[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        g[k] = where(tid + (uint32_t)(2 * half + k) * NTT_THREADS, d[k], c[k]);
        const uint4* src = reinterpret_cast<const uint4*>(in + g[k] * 8);
        const uint4* ts = reinterpret_cast<const uint4*>(in_scale + g[k] * 8);
        lo[k] = src[0];
        hi[k] = src[1];
        tl[k] = ts[0];
        th[k] = ts[1];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) place(g[k], d[k], c[k], lo[k], hi[k], tl[k], th[k]);
    }
  } else if (n_load == 4u * NTT_THREADS) {
    uint32_t d[4], c[4];
    uint64_t g[4];
    uint4 lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g[k] = where(tid + (uint32_t)k * NTT_THREADS, d[k], c[k]);
      const uint4* src = reinterpret_cast<const uint4*>(in + g[k] * 8);
      lo[k] = src[0];
      hi[k] = src[1];
    }
    const uint4 none = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) place(g[k], d[k], c[k], lo[k], hi[k], none, none);
  } else {
    for (uint32_t e = tid; e < n_load; e += NTT_THREADS) {
      uint32_t d, c;
      const uint64_t g = where(e, d, c);
      const uint4* src = reinterpret_cast<const uint4*>(in + g * 8);
      const uint4 lo = src[0], hi = src[1];
      uint4 tl = make_uint4(0, 0, 0, 0), th = tl;
      if (TABLE && pp.has_table) {
        const uint4* ts = reinterpret_cast<const uint4*>(in_scale + g * 8);
        tl = ts[0];
        th = ts[1];
      }
      place(g, d, c, lo, hi, tl, th);
    }
  }
  __syncthreads();

  // ---- R-point DIT over the digit in register rounds of <= 3 stages ---------------------------
  NttTileCtx cx;
  cx.tid = tid; cx.s = s; cx.log_c = log_c; cx.cmask = cmask; cx.tile_elems = tile_elems;
  cx.i0 = i0; cx.base = base; cx.dest_lo0 = dest_lo0;
  // split s into rounds of 3 stages, ending 3 / 2+2 / 3+2 so that no round is a single stage unless s == 1
  uint32_t q0 = pp.log_z;
  while (q0 < s) {
    uint32_t r = s - q0;
    if (r > 2) r = 2;
    if (r == 2) ntt_round<LOG_TILE, 2>(lds, cx, q0, stage_tw);
    else ntt_round<LOG_TILE, 1>(lds, cx, q0, stage_tw);
    q0 += r;
#ifndef HM_NTT_TIMING_NO_ROUND_BARRIER      // TIMING EXPERIMENT ONLY (tools/ab_build.sh): results are wrong without the barrier
    __syncthreads();
#endif
  }

  // ---- epilogue: inter-pass twiddle + packed store, or canonicalise (+ fused scale) on the last pass
  if (!pp.last) {
    const uint32_t shift_m = pp.log_n - (s + pp.log_stride);  // omega_m = omega_n^(2^shift_m)
    const uint32_t lbmask = (1u << pp.log_lb) - 1u;
    // the twiddle of the NEXT element is requested before this one's product starts (a product is ~1 000 cycles: an L2 hit)
    Fr tw_next = fe_zero<FrParams>();
    if (pp.direct_tw && tid < tile_elems) tw_next = load_tw(tw_lo, (i0 + (tid & cmask)) * (tid >> log_c));
    for (uint32_t e = tid; e < tile_elems; e += NTT_THREADS) {
      const uint32_t c = e & cmask, k = e >> log_c;
      const Fr y0 = lds_load<LOG_TILE>(lds, e);
      Fr tw;
      if (pp.direct_tw) {
        tw = tw_next;                                           // omega_m^(i*k), i*k < m <= 2^16
        const uint32_t en = e + NTT_THREADS;
        if (en < tile_elems) tw_next = load_tw(tw_lo, (i0 + (en & cmask)) * (en >> log_c));
      } else {
        const uint64_t ex = ((uint64_t)(i0 + c) * k) << shift_m;  // < n
        const uint32_t elo = (uint32_t)ex & lbmask, ehi = (uint32_t)(ex >> pp.log_lb);
        tw = load_tw(tw_hi, ehi);
        if (elo != 0) tw = fe_mul(tw, load_tw(tw_lo, elo));
      }
      const Fr y = fe_mul(y0, tw);  // product output: normalised, < 2r < 2^256
      uint32_t w[8];
      fe_pack(w, y);
      uint4* dst = reinterpret_cast<uint4*>(out + (base + ((uint64_t)k << pp.log_stride) + c) * 8);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  } else {
    for (uint32_t e = tid; e < tile_elems; e += NTT_THREADS) {
      const uint32_t c = e & cmask, k = e >> log_c;
      const Fr x = lds_load<LOG_TILE>(lds, e);
      const uint64_t g = (dest_lo0 + c) + ((uint64_t)k << pp.log_rows);
      // with a fused constant the product reduces; without one the cheap quotient-estimate reduction does
      Fr y;
      if (FUSED && pp.fin_mode) {
        Fr fin;
        const uint32_t* fp = s_const + 27 + (pp.fin_mode == 2 ? (uint32_t)(g % 3) * 9 : 0u);
#pragma unroll
        for (int i = 0; i < 9; ++i) fin.l[i] = fp[i];
        HM_DECLARE(fin, 1.0);
        y = fe_canonical(fe_mul(x, fin));
      } else {
        y = fe_canonical(fe_reduce_small(x));
      }
      uint32_t w[8];
      fe_pack(w, y);
      uint4* dst = reinterpret_cast<uint4*>(out + g * 8);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// element-wise a[i] *= c (external canonical in/out); used by the domain helpers
__global__ void fr_scale_kernel(uint32_t* __restrict__ a, FrWords c_ext, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t cw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cw[k] = c_ext.w[k];
  // c_int = c * 2^261; a_ext * c_int * 2^-261 = (a*c) in external form
  const Fr c_int = fe_from_ext<FrParams>(cw);
  uint4* p = reinterpret_cast<uint4*>(a + i * 8);
  const uint4 lo = p[0], hi = p[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Fr y = fe_canonical(fe_mul(fe_unpack<FrParams>(w), c_int));
  uint32_t o[8];
  fe_pack(o, y);
  p[0] = make_uint4(o[0], o[1], o[2], o[3]);
  p[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// a_b[i] *= table[i] / 32 for `gridDim.y` back-to-back arrays of n elements (table: external words of 32 * s_i, so that the
// product of two external words is the external word of a[i] * s_i)
__global__ void fr_mul_table_kernel(uint32_t* __restrict__ a, const uint32_t* __restrict__ table, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4* p = reinterpret_cast<uint4*>(a + ((uint64_t)blockIdx.y * n + i) * 8);
  const uint4* t = reinterpret_cast<const uint4*>(table + i * 8);
  const uint4 lo = p[0], hi = p[1], tl = t[0], th = t[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  const uint32_t tw[8] = {tl.x, tl.y, tl.z, tl.w, th.x, th.y, th.z, th.w};
  Fr ts = fe_unpack<FrParams>(tw);
  HM_DECLARE(ts, 6.0);
  Fr y = fe_canonical(fe_mul(fe_unpack<FrParams>(w), ts));
  uint32_t o[8];
  fe_pack(o, y);
  p[0] = make_uint4(o[0], o[1], o[2], o[3]);
  p[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// the same with one table per array (blockIdx.y = array): several cosets' partials in one launch
__global__ void fr_mul_tables_kernel(uint32_t* __restrict__ a, NttCosetTables tabs, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4* p = reinterpret_cast<uint4*>(a + ((uint64_t)blockIdx.y * n + i) * 8);
  const uint4* t = reinterpret_cast<const uint4*>(tabs.table[blockIdx.y] + i * 8);
  const uint4 lo = p[0], hi = p[1], tl = t[0], th = t[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  const uint32_t tw[8] = {tl.x, tl.y, tl.z, tl.w, th.x, th.y, th.z, th.w};
  Fr ts = fe_unpack<FrParams>(tw);
  HM_DECLARE(ts, 6.0);
  Fr y = fe_canonical(fe_mul(fe_unpack<FrParams>(w), ts));
  uint32_t o[8];
  fe_pack(o, y);
  p[0] = make_uint4(o[0], o[1], o[2], o[3]);
  p[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// a[i] *= c3[i % 3] (EvaluationDomain::distribute_powers_zeta); c3: 3 external constants
__global__ void fr_mul_pattern3_kernel(uint32_t* a, FrWords3 c3_ext, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t cw[8];
  const uint32_t sel = (uint32_t)(i % 3);
#pragma unroll
  for (int k = 0; k < 8; ++k) cw[k] = sel == 0 ? c3_ext.w[k] : sel == 1 ? c3_ext.w[8 + k] : c3_ext.w[16 + k];
  const Fr c_int = fe_from_ext<FrParams>(cw);
  uint4* p = reinterpret_cast<uint4*>(a + i * 8);
  const uint4 lo = p[0], hi = p[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  Fr y = fe_canonical(fe_mul(fe_unpack<FrParams>(w), c_int));
  uint32_t o[8];
  fe_pack(o, y);
  p[0] = make_uint4(o[0], o[1], o[2], o[3]);
  p[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// ---------------------------------------------------------------------------------------------
// host side: plan + launch
// ---------------------------------------------------------------------------------------------
#ifndef HM_NTT_LOG_TILE
#define HM_NTT_LOG_TILE 11
#endif
constexpr int LOG_TILE = HM_NTT_LOG_TILE;
// Sub-problems up to 2^NTT_DIRECT_LOG get a DIRECT inter-pass twiddle table (omega_m^e for every e < m, 36 B per entry: 75 MB at
// 2^21) instead of two sqrt-sized tables and a forming product per element.  Round 5 (tools/ntt_plans.py, profiles/r05_ntt_plans.txt):
// 16 -> 21 together with the smaller digit first takes 6.7 % of the VALU instructions out of a 2^21 transform (3 328 -> 3 105 lane
// instructions per element) and 4-6 % off every prover-sized shape (48 x 2^21: 11.5 -> 10.9 ms); the table is read as 36-byte
// gathers and the transform stays VALU-bound.  Beyond 2^21 the plan has three passes and only its middle pass can be direct.
#ifndef HM_NTT_DIRECT_LOG
#define HM_NTT_DIRECT_LOG 21
#endif
constexpr uint32_t NTT_DIRECT_LOG = HM_NTT_DIRECT_LOG;
// (a direct first-pass table is used by two-pass plans only: in a three-pass plan the first pass's m is n but the middle pass's is not)

static int plan_digits(uint32_t log_n, uint32_t digits[3]) {
  if (log_n <= (uint32_t)LOG_TILE) {
    digits[0] = log_n;
    return 1;
  }
  // two passes while both digits fit a tile (measured: 2-8 % faster than three up to 2^21, slower at 2^22
  // where the 11-bit digit leaves single-column, 32-byte accesses)
#ifndef HM_NTT_2PASS_MAX
#define HM_NTT_2PASS_MAX 21
#endif
  const int passes = log_n <= HM_NTT_2PASS_MAX ? 2 : 3;
  uint32_t rem = log_n;
  for (int p = 0; p < passes; ++p) {
    digits[p] = (rem + (passes - p) - 1) / (passes - p);
    rem -= digits[p];
  }
#ifndef HM_NTT_LARGE_FIRST      // two passes: the smaller digit first (2^21 = 2^10 x 2^11): the first pass then has two columns per tile
  if (passes == 2 && digits[0] > digits[1]) { const uint32_t t = digits[0]; digits[0] = digits[1]; digits[1] = t; }
#endif
  return passes;
}

int ntt_plan_first_digit(uint32_t log_n, int* passes) {
  uint32_t digits[3] = {0, 0, 0};
  *passes = plan_digits(log_n, digits);
  return (int)digits[0];
}

constexpr size_t kNttTablesMax = 64;
constexpr size_t kNttTableBytesMax = (size_t)1 << 30;        // 1 GiB: a prover keeps four (omega, omega^-1 at k and extended_k): < 200 MB

void ntt_tables_release(NttTables& t) {
  if (t.d_lo) (void)hipFree(t.d_lo);
  if (t.d_hi) (void)hipFree(t.d_hi);
  if (t.d_mid) (void)hipFree(t.d_mid);
  for (auto& s : t.d_stage)
    if (s) (void)hipFree(s);
  if (t.ready) (void)hipEventDestroy(t.ready);
  t = NttTables{};
}

// Tables for (omega, log_n): built once on the first caller's stream.  Later callers on OTHER streams
// wait on the device for the `ready` event until it has been seen complete (stream-safe publication).
NttTables* ntt_get_tables(DeviceCtx& ctx, const uint64_t omega_ext[4], uint32_t log_n, hipStream_t stream) {
  for (auto& t : ctx.ntt_tables) {
    if (t->log_n == log_n && std::memcmp(t->omega, omega_ext, 32) == 0) {
      if (!t->published) {
        if (hipEventQuery(t->ready) == hipSuccess) {
          t->published = true;
        } else if (stream != t->build_stream) {
          HM_HIP_CHECK_PTR(hipStreamWaitEvent(stream, t->ready, 0));
        }
      }
      t->last_use = ++ctx.ntt_table_clock;
      return t.get();
    }
  }
  // bounded (LRU): with direct tables up to 75 MB a caller that walks through many roots of unity must not keep them all.  A kernel
  // in flight on another stream may still read the set that goes: the device is synchronised first (rare: a prover uses four sets).
  {
    const size_t guess = log_n > (uint32_t)LOG_TILE && log_n <= NTT_DIRECT_LOG ? ((size_t)36 << log_n) : ((size_t)36 << ((log_n + 1) / 2 + 1));
    bool synced = false;
    while (!ctx.ntt_tables.empty() && (ctx.ntt_tables.size() >= kNttTablesMax || ctx.ntt_table_bytes + guess > kNttTableBytesMax)) {
      if (!synced) {
        (void)hipDeviceSynchronize();
        synced = true;
      }
      size_t oldest = 0;
      for (size_t i = 1; i < ctx.ntt_tables.size(); ++i)
        if (ctx.ntt_tables[i]->last_use < ctx.ntt_tables[oldest]->last_use) oldest = i;
      ctx.ntt_table_bytes -= ctx.ntt_tables[oldest]->bytes;
      ntt_tables_release(*ctx.ntt_tables[oldest]);
      ctx.ntt_tables.erase(ctx.ntt_tables.begin() + oldest);
    }
  }
  auto t = std::make_unique<NttTables>();
  t->log_n = log_n;
  std::memcpy(t->omega, omega_ext, 32);
  uint32_t digits[3] = {0, 0, 0};
  const int passes = plan_digits(log_n, digits);
  t->log_lb = (log_n + 1) / 2;
  FrWords om;
  std::memcpy(om.w, omega_ext, 32);
  bool ok = true;
  auto make = [&](uint32_t count, uint32_t shift, uint32_t** dst) {
    if (!ok) return;
    if (hipMalloc(dst, (size_t)count * 36) != hipSuccess) { *dst = nullptr; ok = false; return; }
    t->bytes += (size_t)count * 36;
    hipLaunchKernelGGL(ntt_pow_table_kernel, dim3((count + 127) / 128), dim3(128), 0, stream, om, *dst, count, shift);
    if (hipGetLastError() != hipSuccess) ok = false;
  };
  if (passes > 1) {
    if (log_n <= NTT_DIRECT_LOG && passes == 2) {
      make(1u << log_n, 0, &t->d_lo);                                              // omega_n^e for every e < n
    } else {
      make(1u << t->log_lb, 0, &t->d_lo);
      make(1u << (log_n - t->log_lb), t->log_lb, &t->d_hi);
      const uint32_t log_m1 = log_n - digits[0];                                   // size of the middle pass's sub-problem
      if (passes == 3 && log_m1 <= NTT_DIRECT_LOG) make(1u << log_m1, digits[0], &t->d_mid);   // omega_m1^e = omega_n^(e << s0)
    }
  }
  for (int p = 0; p < passes; ++p) {
    const uint32_t s = digits[p];
    if (t->d_stage[s] == nullptr) {
      const uint32_t count = s == 0 ? 1 : (1u << (s - 1));
      make(count == 0 ? 1 : count, log_n - s, &t->d_stage[s]);
    }
  }
  if (ok && hipEventCreateWithFlags(&t->ready, hipEventDisableTiming) != hipSuccess) ok = false;
  if (ok && hipEventRecord(t->ready, stream) != hipSuccess) ok = false;
  if (!ok) {
    (void)hipStreamSynchronize(stream);      // kernels already enqueued may still write the tables being freed
    ntt_tables_release(*t);
    hm_fail(HM_ERR_HIP, "ntt: twiddle table allocation failed");
    return nullptr;
  }
  t->build_stream = stream;
  t->last_use = ++ctx.ntt_table_clock;
  ctx.ntt_table_bytes += t->bytes;
  ctx.ntt_tables.push_back(std::move(t));
  return ctx.ntt_tables.back().get();
}

AuxSlot* aux_acquire(DeviceCtx& ctx, hipStream_t stream) {
  AuxSlot* pick = nullptr;
  for (auto& s : ctx.aux)
    if (s.used && s.stream == stream) pick = &s;        // stream order already protects its buffers
  if (!pick)
    for (auto& s : ctx.aux)
      if (!s.used) { pick = &s; break; }
  if (!pick) {
    for (auto& s : ctx.aux)                              // a slot whose last user has finished
      if (hipEventQuery(s.done) == hipSuccess && (!pick || s.last_use < pick->last_use)) pick = &s;
  }
  if (!pick) {                                           // all busy on other streams: queue behind the oldest
    pick = &ctx.aux[0];
    for (auto& s : ctx.aux)
      if (s.last_use < pick->last_use) pick = &s;
    if (hipStreamWaitEvent(stream, pick->done, 0) != hipSuccess) {
      hm_fail(HM_ERR_HIP, "aux slot: hipStreamWaitEvent failed");
      return nullptr;
    }
  }
  if (!pick->done && hipEventCreateWithFlags(&pick->done, hipEventDisableTiming) != hipSuccess) {
    hm_fail(HM_ERR_HIP, "aux slot: event creation failed");
    return nullptr;
  }
  pick->used = true;
  pick->stream = stream;
  pick->last_use = ++ctx.aux_clock;
  return pick;
}

int aux_release(DeviceCtx& ctx, AuxSlot* slot, hipStream_t stream) {
  (void)ctx;
  HM_HIP_CHECK(hipEventRecord(slot->done, stream));
  return HM_OK;
}

static void fill_internal9(const uint64_t ext[4], uint32_t out[9]) { host::fr_to_internal9(host::fr_load(ext), out); }

// d_a: `batch` back-to-back arrays of n x 32 B on the device, each transformed in place.  `fused`: optional
// constants (host pointers, external words) multiplied in by the first / last pass.
// d_in != nullptr: an extending transform (EvaluationDomain::coeff_to_extended): `batch` compact arrays of
// n >> log_z coefficients at d_in, zero-padded to n by definition, transformed into d_a (out of place;
// the zero part is never materialised and the first log_z stages of the first pass cost nothing).
int ntt_run(DeviceCtx& ctx, uint32_t* d_a, const uint64_t omega_ext[4], uint32_t log_n, uint32_t batch, const NttFused& fused,
            hipStream_t stream, const uint32_t* d_in, uint32_t log_z) {
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "ntt: log_n > 28 (Fr has 2-adicity 28)");
  if (batch == 0) return HM_OK;
  // several cosets of the same inputs (fused.n_in_scales): the first pass reads `batch` inputs and writes batch * cosets arrays,
  // every later pass is an ordinary batched pass over those
  const uint32_t cosets = fused.n_in_scales, in_batch = batch;
  if (cosets > HM_NTT_COSETS_MAX) return hm_fail(HM_ERR_BAD_ARG, "ntt: more than 16 cosets per call");
  if (cosets) {
    if ((uint64_t)batch * cosets > 65535) return hm_fail(HM_ERR_BAD_ARG, "ntt: batch * cosets > 65535");
    if (!d_in || log_z != 0) return hm_fail(HM_ERR_INTERNAL, "ntt: the multi-coset form is out of place and not extending");
    batch *= cosets;
  }
  if (batch > 65535) return hm_fail(HM_ERR_BAD_ARG, "ntt: batch > 65535");
  NttTables* tab = ntt_get_tables(ctx, omega_ext, log_n, stream);
  if (!tab) return HM_ERR_HIP;
  uint32_t digits[3] = {0, 0, 0};
  const int passes = plan_digits(log_n, digits);
  const uint64_t n = 1ull << log_n;
  uint32_t* scratch = nullptr;
  AuxSlot* slot = nullptr;
  if (passes > 1) {
    slot = aux_acquire(ctx, stream);
    if (!slot) return HM_ERR_HIP;
    scratch = (uint32_t*)slot->scratch.ensure(n * 32 * batch);
    if (!scratch) return hm_fail(HM_ERR_HIP, "ntt: scratch allocation failed");
  }
  const size_t lds_bytes = (size_t)9 * sizeof(uint32_t) << LOG_TILE;
  if (!ctx.ntt_attr_set) {   // per device: a process may drive several GPUs
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_kernel<LOG_TILE, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_kernel<LOG_TILE, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_kernel<LOG_TILE, true, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    ctx.ntt_attr_set = true;
  }
  if (d_in && log_z != 0 && (passes < 2 || log_z > digits[0]))
    return hm_fail(HM_ERR_INTERNAL, "ntt: extending form needs a multi-pass plan and log_z <= first digit");
  // fused constants -> internal 9-limb form on the host (a handful of 4 x u64 products)
  uint32_t coset9[27] = {}, fin9[27] = {};
  uint32_t fin_mode = 0;
  if (fused.coset)
    for (int j = 0; j < 3; ++j) fill_internal9(fused.coset + 4 * j, coset9 + 9 * j);
  if (fused.post3) {
    fin_mode = 2;
    for (int j = 0; j < 3; ++j) {
      host::Fr4 c = host::fr_load(fused.post3 + 4 * j);
      if (fused.scale) c = host::fr_mul(c, host::fr_load(fused.scale));
      host::fr_to_internal9(c, fin9 + 9 * j);
    }
  } else if (fused.scale) {
    fin_mode = 1;
    fill_internal9(fused.scale, fin9);
  }
  uint32_t log_stride = log_n;
  for (int p = 0; p < passes; ++p) {
    NttPassParams pp{};
    pp.log_n = log_n;
    pp.s = digits[p];
    log_stride -= digits[p];
    pp.log_stride = log_stride;
    pp.last = (p == passes - 1) ? 1u : 0u;
    pp.log_lb = tab->log_lb;
    pp.log_rows = log_n - pp.s;
    pp.log_r0 = (pp.last && passes == 3) ? digits[0] : 0u;
    pp.fin_mode = pp.last ? fin_mode : 0u;
    pp.has_coset = (p == 0 && fused.coset) ? 1u : 0u;
    if (pp.has_coset) std::memcpy(pp.coset, coset9, sizeof coset9);
    if (pp.fin_mode) std::memcpy(pp.fin, fin9, sizeof fin9);
    pp.log_z = (p == 0 && d_in) ? log_z : 0u;
    pp.has_table = (p == 0 && (fused.d_in_scale || cosets)) ? 1u : 0u;
    uint32_t log_c = (uint32_t)LOG_TILE > pp.s ? (uint32_t)LOG_TILE - pp.s : 0u;
    const uint32_t avail = pp.last ? pp.log_rows : pp.log_stride;  // columns / rows that exist
    if (log_c > avail) log_c = avail;
    pp.log_c = log_c;
    const uint64_t tiles = n >> (pp.s + log_c);
    const uint32_t* src = (p == 0) ? (d_in ? d_in : d_a) : scratch;
    uint32_t* dst = (p == passes - 1) ? d_a : scratch;
    const uint32_t* lo_tab = tab->d_lo;
    if (!pp.last) {
      if (log_n <= NTT_DIRECT_LOG && passes == 2) {
        pp.direct_tw = 1;            // d_lo = omega_n^e; omega_m^e = omega_n^(e << (log_n - log_m)) only for p == 0 (m = n)
      } else if (p == 1 && tab->d_mid) {
        pp.direct_tw = 1;
        lo_tab = tab->d_mid;
      }
    }
    NttCosetTables zc{};
    if (pp.has_table) {
      if (cosets) {
        zc.count = cosets;
        for (uint32_t c = 0; c < cosets; ++c) zc.table[c] = fused.d_in_scales[c];
      }
      hipLaunchKernelGGL((ntt_pass_kernel<LOG_TILE, true, true>), dim3((uint32_t)tiles, in_batch, cosets ? cosets : 1), dim3(NTT_THREADS), lds_bytes,
                         stream, src, dst, pp, (const uint32_t*)tab->d_stage[pp.s], lo_tab, (const uint32_t*)tab->d_hi, fused.d_in_scale, zc);
    } else if (pp.has_coset | pp.fin_mode) {
      hipLaunchKernelGGL((ntt_pass_kernel<LOG_TILE, true>), dim3((uint32_t)tiles, batch), dim3(NTT_THREADS), lds_bytes, stream, src, dst,
                         pp, (const uint32_t*)tab->d_stage[pp.s], lo_tab, (const uint32_t*)tab->d_hi, (const uint32_t*)nullptr, zc);
    } else {
      hipLaunchKernelGGL((ntt_pass_kernel<LOG_TILE, false>), dim3((uint32_t)tiles, batch), dim3(NTT_THREADS), lds_bytes, stream, src, dst,
                         pp, (const uint32_t*)tab->d_stage[pp.s], lo_tab, (const uint32_t*)tab->d_hi, (const uint32_t*)nullptr, zc);
    }
    HM_HIP_CHECK(hipGetLastError());
  }
  if (slot) return aux_release(ctx, slot, stream);
  return HM_OK;
}

// ---------------------------------------------------------------------------------------------
// transforms on a coset shift * <omega> (the extended domain of evaluate_h taken one coset of the n-th roots at a time:
// 2^(extended_k - k) independent n-point problems instead of one 2^extended_k-point problem -- what lets the cosets go to
// different GPUs)
// ---------------------------------------------------------------------------------------------
constexpr size_t kCosetTablesMax = 48;
constexpr size_t kCosetTableBytesMax = (size_t)2 << 30;      // 2 GiB of HBM held between calls at most (48 tables of 2^24 would be 24 GiB)

void coset_tables_release(DeviceCtx& ctx) {
  for (auto& t : ctx.coset_tables) {
    if (t->d) (void)hipFree(t->d);
    if (t->ready) (void)hipEventDestroy(t->ready);
  }
  ctx.coset_tables.clear();
  ctx.coset_table_bytes = 0;
}

// An allocation failed somewhere else in the library (hm_device_malloc: a prover's own polynomial): every cached table goes -- up to
// 1 GiB of twiddle sets and 2 GiB of coset power tables, all rebuilt on demand.  Kernels in flight may still read them: the device is
// synchronised first.  ctx.mu held.  -> bytes given back.
size_t ntt_caches_give_back(DeviceCtx& ctx) {
  const size_t held = ctx.ntt_table_bytes + ctx.coset_table_bytes;
  if (ctx.ntt_tables.empty() && ctx.coset_tables.empty()) return 0;
  (void)hipDeviceSynchronize();
  for (auto& t : ctx.ntt_tables) ntt_tables_release(*t);
  ctx.ntt_tables.clear();
  ctx.ntt_table_bytes = 0;
  coset_tables_release(ctx);
  return held;
}

// Evict the least recently used table that the CURRENT call has not asked for (last_use < keep_from: a call gathers up to 16
// table pointers before it launches, so its own tables must stay).  A kernel in flight may still read the table that goes:
// the device is synchronised first.  false = nothing evictable.
static bool coset_table_evict_one(DeviceCtx& ctx, uint64_t keep_from, bool* synced) {
  size_t oldest = ctx.coset_tables.size();
  for (size_t i = 0; i < ctx.coset_tables.size(); ++i)
    if (ctx.coset_tables[i]->last_use < keep_from && (oldest == ctx.coset_tables.size() || ctx.coset_tables[i]->last_use < ctx.coset_tables[oldest]->last_use))
      oldest = i;
  if (oldest == ctx.coset_tables.size()) return false;
  if (!*synced) {
    (void)hipDeviceSynchronize();
    *synced = true;
  }
  CosetTable& t = *ctx.coset_tables[oldest];
  if (t.d) (void)hipFree(t.d);
  if (t.ready) (void)hipEventDestroy(t.ready);
  ctx.coset_table_bytes -= (size_t)32 << t.log_n;
  ctx.coset_tables.erase(ctx.coset_tables.begin() + oldest);
  return true;
}

// keep_from: the coset clock when the calling entry point started (tables it has already been handed are not evicted under it)
static const uint32_t* coset_table_get(DeviceCtx& ctx, const uint64_t shift_ext[4], uint32_t log_n, bool internal, hipStream_t stream,
                                       uint64_t keep_from) {
  for (auto& t : ctx.coset_tables) {
    if (t->log_n == log_n && t->internal == (internal ? 1u : 0u) && std::memcmp(t->shift, shift_ext, 32) == 0) {
      if (!t->published) {
        if (hipEventQuery(t->ready) == hipSuccess) {
          t->published = true;
        } else if (stream != t->build_stream) {
          HM_HIP_CHECK_PTR(hipStreamWaitEvent(stream, t->ready, 0));
        }
      }
      t->last_use = ++ctx.coset_clock;
      return t->d;
    }
  }
  const uint64_t n = 1ull << log_n;
  const size_t bytes = (size_t)n * 32;
  bool synced = false;
  // bounded by count AND by bytes (LRU); the cap is soft for ONE call's own working set (nothing evictable: go on)
  while ((ctx.coset_tables.size() >= kCosetTablesMax || ctx.coset_table_bytes + bytes > kCosetTableBytesMax) &&
         coset_table_evict_one(ctx, keep_from, &synced)) {
  }
  auto t = std::make_unique<CosetTable>();
  std::memcpy(t->shift, shift_ext, 32);
  t->log_n = log_n;
  t->internal = internal ? 1u : 0u;
  bool ok = hipMalloc(&t->d, bytes) == hipSuccess;
  while (!ok) {                                            // out of memory: the cache gives back what it can, one table at a time
    (void)hipGetLastError();
    t->d = nullptr;
    if (!coset_table_evict_one(ctx, keep_from, &synced)) break;
    ok = hipMalloc(&t->d, bytes) == hipSuccess;
  }
  if (!ok) t->d = nullptr;
  const host::Fr4 c = internal ? host::fr_mul(host::FR_32, host::FR_32) : host::FR_32;       // 1024 resp. 32, Montgomery words
  if (ok) ok = fr_powers_run(t->d, n, shift_ext, stream) == HM_OK && fr_scale_run(t->d, c.l, n, stream) == HM_OK;
  if (ok && hipEventCreateWithFlags(&t->ready, hipEventDisableTiming) != hipSuccess) ok = false;
  if (ok && hipEventRecord(t->ready, stream) != hipSuccess) ok = false;
  if (!ok) {
    (void)hipStreamSynchronize(stream);
    if (t->d) (void)hipFree(t->d);
    if (t->ready) (void)hipEventDestroy(t->ready);
    hm_fail(HM_ERR_HIP, "ntt: coset table allocation failed");
    return nullptr;
  }
  t->build_stream = stream;
  t->last_use = ++ctx.coset_clock;
  ctx.coset_table_bytes += bytes;
  ctx.coset_tables.push_back(std::move(t));
  return ctx.coset_tables.back()->d;
}

int ntt_coset_run(DeviceCtx& ctx, const uint32_t* d_in, uint32_t* d_out, uint32_t batch, const uint64_t omega_ext[4], uint32_t log_n,
                  const uint64_t shift_ext[4], bool internal, hipStream_t stream) {
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "ntt: log_n > 28 (Fr has 2-adicity 28)");
  if (batch == 0) return HM_OK;
  const uint32_t* table = coset_table_get(ctx, shift_ext, log_n, internal, stream, ctx.coset_clock + 1);
  if (!table) return HM_ERR_HIP;
  NttFused f;
  f.d_in_scale = table;
  return ntt_run(ctx, d_out, omega_ext, log_n, batch, f, stream, d_in == d_out ? nullptr : d_in, 0);
}

int ntt_cosets_run(DeviceCtx& ctx, const uint32_t* d_in, uint32_t* d_out, uint32_t batch, const uint64_t omega_ext[4], uint32_t log_n,
                   const uint64_t* shifts_ext, uint32_t count, bool internal, hipStream_t stream) {
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "ntt: log_n > 28 (Fr has 2-adicity 28)");
  if (batch == 0 || count == 0) return HM_OK;
  if (count > HM_NTT_COSETS_MAX) return hm_fail(HM_ERR_BAD_ARG, "ntt: more than 16 cosets per call");
  NttFused f;
  f.n_in_scales = count;
  const uint64_t mine = ctx.coset_clock + 1;               // the tables handed out from here on belong to this call
  for (uint32_t c = 0; c < count; ++c) {
    f.d_in_scales[c] = coset_table_get(ctx, shifts_ext + 4 * c, log_n, internal, stream, mine);
    if (!f.d_in_scales[c]) return HM_ERR_HIP;
  }
  return ntt_run(ctx, d_out, omega_ext, log_n, batch, f, stream, d_in, 0);
}

int ntt_cosets_inverse_run(DeviceCtx& ctx, uint32_t* d_a, uint32_t count, const uint64_t omega_inv_ext[4], uint32_t log_n,
                           const uint64_t divisor_ext[4], const uint64_t* shift_invs_ext, hipStream_t stream) {
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "ntt: log_n > 28 (Fr has 2-adicity 28)");
  if (count == 0) return HM_OK;
  if (count > HM_NTT_COSETS_MAX) return hm_fail(HM_ERR_BAD_ARG, "ntt: more than 16 cosets per call");
  NttCosetTables tabs{};
  tabs.count = count;
  const uint64_t mine = ctx.coset_clock + 1;
  for (uint32_t c = 0; c < count; ++c) {
    tabs.table[c] = coset_table_get(ctx, shift_invs_ext + 4 * c, log_n, false, stream, mine);
    if (!tabs.table[c]) return HM_ERR_HIP;
  }
  NttFused f;
  f.scale = divisor_ext;
  const int rc = ntt_run(ctx, d_a, omega_inv_ext, log_n, count, f, stream);
  if (rc != HM_OK) return rc;
  const uint64_t n = 1ull << log_n;
  hipLaunchKernelGGL(fr_mul_tables_kernel, dim3((uint32_t)((n + 255) / 256), count), dim3(256), 0, stream, d_a, tabs, n);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int ntt_coset_inverse_run(DeviceCtx& ctx, uint32_t* d_a, uint32_t batch, const uint64_t omega_inv_ext[4], uint32_t log_n,
                          const uint64_t divisor_ext[4], const uint64_t shift_inv_ext[4], hipStream_t stream) {
  if (log_n > 28) return hm_fail(HM_ERR_BAD_ARG, "ntt: log_n > 28 (Fr has 2-adicity 28)");
  if (batch == 0) return HM_OK;
  const uint32_t* table = coset_table_get(ctx, shift_inv_ext, log_n, false, stream, ctx.coset_clock + 1);
  if (!table) return HM_ERR_HIP;
  NttFused f;
  f.scale = divisor_ext;
  const int rc = ntt_run(ctx, d_a, omega_inv_ext, log_n, batch, f, stream);
  if (rc != HM_OK) return rc;
  const uint64_t n = 1ull << log_n;
  hipLaunchKernelGGL(fr_mul_table_kernel, dim3((uint32_t)((n + 255) / 256), batch), dim3(256), 0, stream, d_a, table, n);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int fr_scale_run(uint32_t* d_a, const uint64_t c_ext[4], uint64_t n, hipStream_t stream) {
  FrWords c;
  std::memcpy(c.w, c_ext, 32);
  hipLaunchKernelGGL(fr_scale_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_a, c, n);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int fr_mul_pattern3_run(uint32_t* d_a, const uint64_t c3_ext[12], uint64_t n, hipStream_t stream) {
  FrWords3 c;
  std::memcpy(c.w, c3_ext, 96);
  hipLaunchKernelGGL(fr_mul_pattern3_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_a, c, n);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

}  // namespace hm
