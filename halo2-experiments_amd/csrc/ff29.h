// ff29.h -- 256-bit prime-field arithmetic for gfx950 on 9 unsaturated 29-bit limbs.
//
// Why this shape (measured, tools/ubench/int_rate.hip on MI355X): every 32-bit integer VALU op,
// including v_mad_u64_u32 (32x32+64 -> 64), issues at ~4.2 cycles per wave64 instruction, i.e. the
// wide multiply-add costs the same as an add.  The cheapest 254-bit Montgomery product is therefore
// the one with the fewest instructions: with 29-bit limbs the 58-bit partial products of a whole
// column (9 from a*b, 9 from m*p, plus carries) fit a 64-bit accumulator, so the product is a pure
// chain of v_mad_u64_u32 with no carry handling at all (~230 instructions, vs ~300 for saturated
// 8x32-bit limbs where every wide mad needs a carry-catching companion).  Additions are 9
// independent v_add_u32; the 7 spare bits of the 261-bit container absorb lazy (unreduced) sums.
//
// Representation ("internal form"): value v (any residue representative, v < 2^261) stands for
// the field element v * 2^-261 mod p.  The reference's memory format ("external form",
// halo2curves bn256) is 4 x u64 Montgomery with radix 2^256; conversion is one
// mont_mul by a constant (EXT2INT / INT2EXT).  Linear maps (the NTT) need no conversion at all.
//
// Bound discipline: compile the host build with -DHM_BOUNDS and every element carries worst-case
// (data-independent) bounds -- `lb` on limbs 0..7, `tb` on the top limb, `vb` on value/modulus --
// and every primitive asserts its precondition.  tests/test_ff29_host.py runs every formula in
// this library once under that build, which proves the bounds for all inputs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HM_HD __host__ __device__ __forceinline__
#else
#define HM_HD inline
#endif

#ifdef HM_BOUNDS
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define HM_CHECK(cond, what)                                                        \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      std::fprintf(stderr, "HM_BOUNDS violation: %s (%s:%d)\n", what, __FILE__, __LINE__); \
      std::abort();                                                                 \
    }                                                                               \
  } while (0)
#endif

namespace hm {

#include "bn256_constants.inc"

constexpr uint32_t MASK29 = (1u << 29) - 1u;

struct Arr9 {
  uint32_t v[9];
};

// k * MOD with limb i raised by 2^BITS borrowed from limb i+1 (same integer): adding it before a
// limb-wise subtraction keeps every limb non-negative as long as the subtrahend's limbs are
// <= 2^BITS - 2^(BITS-29) (top limb: <= the constant's top limb).
template <class F, int K, int BITS>
constexpr Arr9 make_sub_const() {
  Arr9 out{};
  uint64_t carry = 0;
  uint32_t c[9] = {};
  for (int i = 0; i < 9; ++i) {
    uint64_t t = (uint64_t)F::MOD[i] * (uint64_t)K + carry;
    if (i < 8) {
      c[i] = (uint32_t)(t & MASK29);
      carry = t >> 29;
    } else {
      c[i] = (uint32_t)t;
    }
  }
  const uint32_t b = 1u << (BITS - 29);
  out.v[0] = c[0] + (1u << BITS);
  for (int i = 1; i < 8; ++i) out.v[i] = c[i] + (1u << BITS) - b;
  out.v[8] = c[8] - b;
  return out;
}
template <class F, int K, int BITS>
struct SubConst {
  static constexpr Arr9 value = make_sub_const<F, K, BITS>();
};

template <class F>
struct Fe {
  uint32_t l[9];
#ifdef HM_BOUNDS
  double vb = 0;    // value < vb * MOD
  uint64_t lb = 0;  // limbs 0..7 <= lb
  uint64_t tb = 0;  // limb 8 <= tb
#endif
};

#ifdef HM_BOUNDS
template <class F>
inline double mod_as_double() {
  double m = 0;
  for (int i = 8; i >= 0; --i) m = m * 536870912.0 + (double)F::MOD[i];
  return m;
}
template <class F>
inline uint64_t top_bound_from_value(double vb) {  // floor(vb*MOD / 2^232) + 1
  return (uint64_t)std::floor(vb * mod_as_double<F>() / std::ldexp(1.0, 232)) + 1;
}
template <class F>
inline void set_bounds(Fe<F>& a, double vb, uint64_t lb, uint64_t tb) {
  a.vb = vb; a.lb = lb; a.tb = tb;
  HM_CHECK(lb < (1ull << 32) && tb < (1ull << 32), "limb bound exceeds 32 bits");
  HM_CHECK(vb * mod_as_double<F>() < std::ldexp(1.0, 261), "value bound exceeds 2^261");
}
// declare an element read from memory: normalised limbs, value < vb*MOD
template <class F>
inline void declare(Fe<F>& a, double vb) {
  set_bounds(a, vb, MASK29, top_bound_from_value<F>(vb));
  for (int i = 0; i < 8; ++i) HM_CHECK(a.l[i] <= MASK29, "declared element has an unnormalised limb");
  HM_CHECK(a.l[8] <= a.tb, "declared element exceeds its value bound");
}
#define HM_DECLARE(a, vb) ::hm::declare(a, vb)
#else
#define HM_DECLARE(a, vb) ((void)0)
#endif

template <class F>
HM_HD Fe<F> fe_zero() {
  Fe<F> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = 0;
#ifdef HM_BOUNDS
  set_bounds(r, 0.0, 0, 0);
#endif
  return r;
}

template <class F>
HM_HD Fe<F> fe_const(const uint32_t (&c)[9]) {
  Fe<F> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = c[i];
#ifdef HM_BOUNDS
  set_bounds(r, 1.0, MASK29, F::MOD[8]);
#endif
  return r;
}
template <class F>
HM_HD Fe<F> fe_one() { return fe_const<F>(F::ONE); }

// ---------------------------------------------------------------------------------------------
// Montgomery product, radix 2^261.  Operand scanning with the reduction interleaved; t[] is a
// sliding window of 64-bit column accumulators.  Output: normalised limbs, value < a*b/2^261 + MOD.
// ---------------------------------------------------------------------------------------------
template <class F>
HM_HD Fe<F> fe_mul(const Fe<F>& a, const Fe<F>& b) {
#ifdef HM_BOUNDS
  {
    long double A = (long double)(a.lb > a.tb ? a.lb : a.tb), B = (long double)(b.lb > b.tb ? b.lb : b.tb);
    HM_CHECK(9.0L * A * B + 9.0L * 288230376151711744.0L + 1099511627776.0L < 18446744073709551616.0L,
             "fe_mul column sum may overflow 64 bits");
  }
#endif
  uint64_t t[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] += (uint64_t)a.l[j] * b.l[i];
    const uint32_t m = ((uint32_t)t[0] * F::INV29) & MASK29;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] += (uint64_t)m * F::MOD[j];
    t[1] += t[0] >> 29;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = t[j + 1];
    t[9] = 0;
  }
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r.l[j] = (uint32_t)t[j] & MASK29;
    t[j + 1] += t[j] >> 29;
  }
  r.l[8] = (uint32_t)t[8];
#ifdef HM_BOUNDS
  {
    double vb = a.vb * b.vb * mod_as_double<F>() / std::ldexp(1.0, 261) + 1.0;
    set_bounds(r, vb, MASK29, top_bound_from_value<F>(vb));
    HM_CHECK(r.l[8] <= r.tb, "fe_mul result exceeds its bound");
  }
#endif
  return r;
}

// Fused a*b + c*d with ONE Montgomery reduction: (a*b + c*d) * 2^-261.  Both products accumulate
// into the same 64-bit columns (18 + 9 partial products per column), so it costs 243 wide
// multiplies instead of 324 for two separate products plus an addition.
template <class F>
HM_HD Fe<F> fe_mul2(const Fe<F>& a, const Fe<F>& b, const Fe<F>& c, const Fe<F>& d) {
#ifdef HM_BOUNDS
  {
    long double A = (long double)(a.lb > a.tb ? a.lb : a.tb), B = (long double)(b.lb > b.tb ? b.lb : b.tb);
    long double C = (long double)(c.lb > c.tb ? c.lb : c.tb), D = (long double)(d.lb > d.tb ? d.lb : d.tb);
    HM_CHECK(9.0L * (A * B + C * D) + 9.0L * 288230376151711744.0L + 1099511627776.0L < 18446744073709551616.0L,
             "fe_mul2 column sum may overflow 64 bits");
  }
#endif
  uint64_t t[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] += (uint64_t)a.l[j] * b.l[i];
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] += (uint64_t)c.l[j] * d.l[i];
    const uint32_t m = ((uint32_t)t[0] * F::INV29) & MASK29;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] += (uint64_t)m * F::MOD[j];
    t[1] += t[0] >> 29;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = t[j + 1];
    t[9] = 0;
  }
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r.l[j] = (uint32_t)t[j] & MASK29;
    t[j + 1] += t[j] >> 29;
  }
  r.l[8] = (uint32_t)t[8];
#ifdef HM_BOUNDS
  {
    double vb = (a.vb * b.vb + c.vb * d.vb) * mod_as_double<F>() / std::ldexp(1.0, 261) + 1.0;
    set_bounds(r, vb, MASK29, top_bound_from_value<F>(vb));
    HM_CHECK(r.l[8] <= r.tb, "fe_mul2 result exceeds its bound");
  }
#endif
  return r;
}

// Montgomery square: 45 wide multiplies instead of 81 (cross terms use pre-doubled limbs).
template <class F>
HM_HD Fe<F> fe_sqr(const Fe<F>& a) {
#ifdef HM_BOUNDS
  {
    long double A = (long double)(a.lb > a.tb ? a.lb : a.tb);
    HM_CHECK(2.0L * A < 4294967296.0L, "fe_sqr doubled limb overflows");
    HM_CHECK(9.0L * A * A + 9.0L * 288230376151711744.0L + 1099511627776.0L < 18446744073709551616.0L,
             "fe_sqr column sum may overflow 64 bits");
  }
#endif
  uint32_t d[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) d[j] = a.l[j] << 1;
  uint64_t t[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    // row i contributes a_i*a_i to column 2i and a_i*(2a_j) to column i+j (j > i); in window
    // coordinates (column - i) that is slots i .. 8.
    if (2 * i - i <= 8) t[i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
    for (int j = i + 1; j < 9; ++j)
      if (j <= 8) t[j] += (uint64_t)a.l[i] * d[j];
    const uint32_t m = ((uint32_t)t[0] * F::INV29) & MASK29;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] += (uint64_t)m * F::MOD[j];
    t[1] += t[0] >> 29;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = t[j + 1];
    t[9] = 0;
  }
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r.l[j] = (uint32_t)t[j] & MASK29;
    t[j + 1] += t[j] >> 29;
  }
  r.l[8] = (uint32_t)t[8];
#ifdef HM_BOUNDS
  {
    double vb = a.vb * a.vb * mod_as_double<F>() / std::ldexp(1.0, 261) + 1.0;
    set_bounds(r, vb, MASK29, top_bound_from_value<F>(vb));
    HM_CHECK(r.l[8] <= r.tb, "fe_sqr result exceeds its bound");
  }
#endif
  return r;
}

// ---------------------------------------------------------------------------------------------
// lazy linear operations
// ---------------------------------------------------------------------------------------------
template <class F>
HM_HD Fe<F> fe_add(const Fe<F>& a, const Fe<F>& b) {
  Fe<F> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
#ifdef HM_BOUNDS
  set_bounds(r, a.vb + b.vb, a.lb + b.lb, a.tb + b.tb);
#endif
  return r;
}

template <class F>
HM_HD Fe<F> fe_dbl(const Fe<F>& a) { return fe_add(a, a); }

template <class F>
HM_HD Fe<F> fe_mul4(const Fe<F>& a) {
  Fe<F> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] << 2;
#ifdef HM_BOUNDS
  set_bounds(r, 4 * a.vb, 4 * a.lb, 4 * a.tb);
#endif
  return r;
}

// a - b + K*MOD, limb-wise, no borrows.  Requires b's limbs <= 2^BITS - 2^(BITS-29) and b's top
// limb <= the constant's top limb (implied by b < (K-1)*MOD for normalised b).
template <int K, int BITS, class F>
HM_HD Fe<F> fe_sub(const Fe<F>& a, const Fe<F>& b) {
  constexpr Arr9 S = SubConst<F, K, BITS>::value;
#ifdef HM_BOUNDS
  HM_CHECK(b.lb <= (1ull << BITS) - (1ull << (BITS - 29)), "fe_sub: subtrahend limb bound too large for BITS");
  HM_CHECK(b.tb <= S.v[8], "fe_sub: subtrahend top limb may exceed the constant's");
#endif
  Fe<F> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + S.v[i] - b.l[i];
#ifdef HM_BOUNDS
  uint64_t smax = 0;
  for (int i = 0; i < 8; ++i) smax = S.v[i] > smax ? S.v[i] : smax;
  set_bounds(r, a.vb + K, a.lb + smax, a.tb + S.v[8]);
#endif
  return r;
}

// carry-propagate to 29-bit limbs (value unchanged; the top limb keeps the excess)
template <class F>
HM_HD Fe<F> fe_norm(const Fe<F>& a) {
#ifdef HM_BOUNDS
  HM_CHECK(a.lb + 16 < (1ull << 32) && a.tb + 16 < (1ull << 32), "fe_norm: carry add may overflow");
#endif
  Fe<F> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t t = a.l[i] + c;
    r.l[i] = t & MASK29;
    c = t >> 29;
  }
  r.l[8] = a.l[8] + c;
#ifdef HM_BOUNDS
  set_bounds(r, a.vb, MASK29, top_bound_from_value<F>(a.vb));
  HM_CHECK(r.l[8] <= r.tb, "fe_norm result exceeds its bound");
#endif
  return r;
}

// ---------------------------------------------------------------------------------------------
// comparisons / canonical form
// ---------------------------------------------------------------------------------------------
// a is a fe_mul/fe_sqr output (normalised limbs, value < 3*MOD): is it 0 mod MOD?
template <class F>
HM_HD bool fe_is_zero_mod(const Fe<F>& a) {
#ifdef HM_BOUNDS
  HM_CHECK(a.lb <= MASK29 && a.vb <= 3.0, "fe_is_zero_mod needs a normalised value < 3*MOD");
#endif
  uint32_t z0 = 0, z1 = 0, z2 = 0;
  // 2*MOD in 29-bit limbs (constant-folded)
  uint32_t two[9];
  {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t t = 2u * F::MOD[i] + carry;
      two[i] = t & MASK29;
      carry = t >> 29;
    }
    two[8] = 2u * F::MOD[8] + carry;
  }
  // cheap filter on the low limb first: almost every caller sees a non-multiple of MOD
  if ((a.l[0] != 0u) & (a.l[0] != F::MOD[0]) & (a.l[0] != two[0])) return false;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    z0 |= a.l[i];
    z1 |= a.l[i] ^ F::MOD[i];
    z2 |= a.l[i] ^ two[i];
  }
  return (z0 == 0) | (z1 == 0) | (z2 == 0);
}

// a: normalised limbs, value < 3*MOD  ->  the canonical representative in [0, MOD)
template <class F>
HM_HD Fe<F> fe_canonical(const Fe<F>& a) {
#ifdef HM_BOUNDS
  HM_CHECK(a.lb <= MASK29 && a.vb <= 3.0, "fe_canonical needs a normalised value < 3*MOD");
#endif
  Fe<F> cur = a;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    uint32_t d[9];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t t = cur.l[i] - F::MOD[i] - borrow;
      d[i] = t & MASK29;
      borrow = t >> 31;
    }
    const uint32_t top = cur.l[8] - F::MOD[8] - borrow;
    d[8] = top;
    const bool neg = (top >> 31) != 0;  // top limbs are < 2^29, so bit 31 is the sign
#pragma unroll
    for (int i = 0; i < 9; ++i) cur.l[i] = neg ? cur.l[i] : d[i];
  }
#ifdef HM_BOUNDS
  set_bounds(cur, 1.0, MASK29, F::MOD[8]);
#endif
  return cur;
}

// a: normalised limbs, any value < 2^261  ->  normalised limbs, value < 3*MOD, same residue.
// Quotient estimate from the top limb (q <= floor(a / MOD), short by at most 2), then a - q*MOD with
// signed 64-bit carries: ~50 instructions, against ~215 for a Montgomery product by ONE.
template <class F>
HM_HD Fe<F> fe_reduce_small(const Fe<F>& a) {
#ifdef HM_BOUNDS
  HM_CHECK(a.lb <= MASK29 && a.tb < (1ull << 29), "fe_reduce_small needs normalised limbs and a value < 2^261");
#endif
  const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * F::QK) >> 53);   // floor(top / (TOPMOD + 1)), possibly one less
  Fe<F> r;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * F::MOD[i]) + carry;
    r.l[i] = (uint32_t)t & MASK29;
    carry = t >> 29;
  }
  r.l[8] = (uint32_t)((int64_t)a.l[8] - (int64_t)((uint64_t)q * F::MOD[8]) + carry);
#ifdef HM_BOUNDS
  set_bounds(r, 3.0, MASK29, top_bound_from_value<F>(3.0));
  HM_CHECK(r.l[8] <= r.tb, "fe_reduce_small result exceeds 3*MOD");
#endif
  return r;
}

// ---------------------------------------------------------------------------------------------
// packing: 8 x u32 little-endian (the reference's 4 x u64 limbs) <-> 9 x 29-bit limbs
// ---------------------------------------------------------------------------------------------
template <class F>
HM_HD Fe<F> fe_unpack(const uint32_t (&w)[8]) {
  Fe<F> r;
  r.l[0] = w[0] & MASK29;
  r.l[1] = ((w[0] >> 29) | (w[1] << 3)) & MASK29;
  r.l[2] = ((w[1] >> 26) | (w[2] << 6)) & MASK29;
  r.l[3] = ((w[2] >> 23) | (w[3] << 9)) & MASK29;
  r.l[4] = ((w[3] >> 20) | (w[4] << 12)) & MASK29;
  r.l[5] = ((w[4] >> 17) | (w[5] << 15)) & MASK29;
  r.l[6] = ((w[5] >> 14) | (w[6] << 18)) & MASK29;
  r.l[7] = ((w[6] >> 11) | (w[7] << 21)) & MASK29;
  r.l[8] = w[7] >> 8;
#ifdef HM_BOUNDS
  set_bounds(r, std::ldexp(1.0, 256) / mod_as_double<F>(), MASK29, (1u << 24) - 1);
#endif
  return r;
}

// a must be normalised with value < 2^256
template <class F>
HM_HD void fe_pack(uint32_t (&w)[8], const Fe<F>& a) {
#ifdef HM_BOUNDS
  HM_CHECK(a.lb <= MASK29 && a.tb < (1ull << 24), "fe_pack needs a normalised value < 2^256");
#endif
  w[0] = a.l[0] | (a.l[1] << 29);
  w[1] = (a.l[1] >> 3) | (a.l[2] << 26);
  w[2] = (a.l[2] >> 6) | (a.l[3] << 23);
  w[3] = (a.l[3] >> 9) | (a.l[4] << 20);
  w[4] = (a.l[4] >> 12) | (a.l[5] << 17);
  w[5] = (a.l[5] >> 15) | (a.l[6] << 14);
  w[6] = (a.l[6] >> 18) | (a.l[7] << 11);
  w[7] = (a.l[7] >> 21) | (a.l[8] << 8);
}

// external (radix 2^256 Montgomery, canonical) -> internal form, and back (canonical output)
template <class F>
HM_HD Fe<F> fe_from_ext(const uint32_t (&w)[8]) {
  Fe<F> x = fe_unpack<F>(w);
  return fe_mul(x, fe_const<F>(F::EXT2INT));
}
template <class F>
HM_HD void fe_to_ext(uint32_t (&w)[8], const Fe<F>& a_normalised) {
  Fe<F> x = fe_canonical(fe_mul(a_normalised, fe_const<F>(F::INT2EXT)));
  fe_pack(w, x);
}

}  // namespace hm
