// host_fr.h -- BN256 Fr on the HOST in the reference's memory format (4 x u64 Montgomery, radix
// 2^256): the handful of products a call needs before its kernels start -- combining the constants
// that are fused into an NTT pass (ifft divisor x coset pattern) and converting them to the 9 x 29-bit
// internal form of ff29.h, so that they travel to the kernels BY VALUE (no shared constants buffer).
#pragma once
#include <stdint.h>
#include <string.h>

namespace hm {
namespace host {

typedef unsigned __int128 u128;
struct Fr4 {
  uint64_t l[4];
};

static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;   // -r^-1 mod 2^64
static const Fr4 FR_ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
// 32 in Montgomery form: x_ext * 32 is the integer that ff29.h's internal form (radix 2^261) holds
static const Fr4 FR_32 = {{0x2fd4e1568fffff57ULL, 0x75bba827a494b01aULL, 0x5301fa84819caa80ULL, 0x0dc83629563d4475ULL}};

static inline Fr4 fr_mul(const Fr4& a, const Fr4& b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  for (int i = 0; i < 4; ++i) {
    const uint64_t bi = b.l[i];
    u128 c = (u128)a.l[0] * bi + t0; t0 = (uint64_t)c; c >>= 64;
    c += (u128)a.l[1] * bi + t1; t1 = (uint64_t)c; c >>= 64;
    c += (u128)a.l[2] * bi + t2; t2 = (uint64_t)c; c >>= 64;
    c += (u128)a.l[3] * bi + t3; t3 = (uint64_t)c; c >>= 64;
    c += t4; t4 = (uint64_t)c;
    const uint64_t t5 = (uint64_t)(c >> 64);
    const uint64_t m = t0 * FR_INV;
    c = (u128)m * FR_MOD[0] + t0; c >>= 64;
    c += (u128)m * FR_MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;
    c += (u128)m * FR_MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;
    c += (u128)m * FR_MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;
    c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
  }
  // one conditional subtraction (inputs < r => t < 2r)
  u128 d = (u128)t0 - FR_MOD[0];
  const uint64_t r0 = (uint64_t)d;
  d = (u128)t1 - FR_MOD[1] - ((uint64_t)(d >> 64) & 1);
  const uint64_t r1 = (uint64_t)d;
  d = (u128)t2 - FR_MOD[2] - ((uint64_t)(d >> 64) & 1);
  const uint64_t r2 = (uint64_t)d;
  d = (u128)t3 - FR_MOD[3] - ((uint64_t)(d >> 64) & 1);
  const uint64_t r3 = (uint64_t)d;
  const bool ge = t4 || !((uint64_t)(d >> 64) & 1);
  Fr4 r;
  r.l[0] = ge ? r0 : t0; r.l[1] = ge ? r1 : t1; r.l[2] = ge ? r2 : t2; r.l[3] = ge ? r3 : t3;
  return r;
}

static inline Fr4 fr_sub(const Fr4& a, const Fr4& b) {            // canonical inputs -> canonical output
  u128 d = (u128)a.l[0] - b.l[0];
  uint64_t t0 = (uint64_t)d;
  d = (u128)a.l[1] - b.l[1] - ((uint64_t)(d >> 64) & 1);
  uint64_t t1 = (uint64_t)d;
  d = (u128)a.l[2] - b.l[2] - ((uint64_t)(d >> 64) & 1);
  uint64_t t2 = (uint64_t)d;
  d = (u128)a.l[3] - b.l[3] - ((uint64_t)(d >> 64) & 1);
  uint64_t t3 = (uint64_t)d;
  if ((uint64_t)(d >> 64) & 1) {   // borrow: add r back
    u128 c = (u128)t0 + FR_MOD[0];
    t0 = (uint64_t)c;
    c = (c >> 64) + t1 + FR_MOD[1];
    t1 = (uint64_t)c;
    c = (c >> 64) + t2 + FR_MOD[2];
    t2 = (uint64_t)c;
    c = (c >> 64) + t3 + FR_MOD[3];
    t3 = (uint64_t)c;
  }
  return Fr4{{t0, t1, t2, t3}};
}
static inline bool fr_is_zero(const Fr4& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline bool fr_eq(const Fr4& a, const Fr4& b) {
  return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}
static inline Fr4 fr_inv(const Fr4& a) {   // a^(r-2); zero stays zero
  const uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
  Fr4 acc = FR_ONE;
  for (int i = 255; i >= 0; --i) {
    acc = fr_mul(acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, a);
  }
  return acc;
}

static inline Fr4 fr_load(const uint64_t w[4]) {
  Fr4 r;
  memcpy(r.l, w, 32);
  return r;
}

// external Montgomery words (canonical) -> the 9 x 29-bit limbs of the internal form (canonical, < r)
static inline void fr_to_internal9(const Fr4& x_ext, uint32_t out[9]) {
  const Fr4 v = fr_mul(x_ext, FR_32);
  uint32_t w[8];
  memcpy(w, v.l, 32);
  out[0] = w[0] & 0x1fffffffu;
  out[1] = ((w[0] >> 29) | (w[1] << 3)) & 0x1fffffffu;
  out[2] = ((w[1] >> 26) | (w[2] << 6)) & 0x1fffffffu;
  out[3] = ((w[2] >> 23) | (w[3] << 9)) & 0x1fffffffu;
  out[4] = ((w[3] >> 20) | (w[4] << 12)) & 0x1fffffffu;
  out[5] = ((w[4] >> 17) | (w[5] << 15)) & 0x1fffffffu;
  out[6] = ((w[5] >> 14) | (w[6] << 18)) & 0x1fffffffu;
  out[7] = ((w[6] >> 11) | (w[7] << 21)) & 0x1fffffffu;
  out[8] = w[7] >> 8;
}

}  // namespace host
}  // namespace hm
