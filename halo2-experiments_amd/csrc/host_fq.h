// host_fq.h -- BN256 Fq / G1 on the HOST in the reference's own memory format (4 x u64 Montgomery,
// radix 2^256): used only for the last, latency-bound steps of an MSM that are not worth a launch --
// the Horner fold of the W window sums (c doublings each) and the final affine normalisation --
// and for hm_g1_sum.  ~30 ns per product on one core, so the whole fold is tens of microseconds.
#pragma once
#include <stdint.h>
#include <string.h>

namespace hm {
namespace host {

typedef unsigned __int128 u128;
struct Fq4 {
  uint64_t l[4];
};

static const uint64_t FQ_MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t FQ_INV = 0x87d20782e4866389ULL;   // -p^-1 mod 2^64
static const Fq4 FQ_ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};

static inline bool fq_is_zero(const Fq4& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline bool fq_eq(const Fq4& a, const Fq4& b) {
  return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}
// r = a - p if a >= p (a < 2p)
static inline Fq4 fq_cond_sub(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3, uint64_t carry) {
  u128 b = (u128)t0 - FQ_MOD[0];
  const uint64_t r0 = (uint64_t)b;
  b = (u128)t1 - FQ_MOD[1] - ((uint64_t)(b >> 64) & 1);
  const uint64_t r1 = (uint64_t)b;
  b = (u128)t2 - FQ_MOD[2] - ((uint64_t)(b >> 64) & 1);
  const uint64_t r2 = (uint64_t)b;
  b = (u128)t3 - FQ_MOD[3] - ((uint64_t)(b >> 64) & 1);
  const uint64_t r3 = (uint64_t)b;
  const bool ge = carry || !((uint64_t)(b >> 64) & 1);
  Fq4 r;
  r.l[0] = ge ? r0 : t0; r.l[1] = ge ? r1 : t1; r.l[2] = ge ? r2 : t2; r.l[3] = ge ? r3 : t3;
  return r;
}
static inline Fq4 fq_add(const Fq4& a, const Fq4& b) {
  u128 c = (u128)a.l[0] + b.l[0];
  const uint64_t t0 = (uint64_t)c;
  c = (c >> 64) + a.l[1] + b.l[1];
  const uint64_t t1 = (uint64_t)c;
  c = (c >> 64) + a.l[2] + b.l[2];
  const uint64_t t2 = (uint64_t)c;
  c = (c >> 64) + a.l[3] + b.l[3];
  return fq_cond_sub(t0, t1, t2, (uint64_t)c, (uint64_t)(c >> 64));
}
static inline Fq4 fq_sub(const Fq4& a, const Fq4& b) {
  u128 d = (u128)a.l[0] - b.l[0];
  uint64_t t0 = (uint64_t)d;
  d = (u128)a.l[1] - b.l[1] - ((uint64_t)(d >> 64) & 1);
  uint64_t t1 = (uint64_t)d;
  d = (u128)a.l[2] - b.l[2] - ((uint64_t)(d >> 64) & 1);
  uint64_t t2 = (uint64_t)d;
  d = (u128)a.l[3] - b.l[3] - ((uint64_t)(d >> 64) & 1);
  uint64_t t3 = (uint64_t)d;
  if ((uint64_t)(d >> 64) & 1) {   // borrow: add p back
    u128 c = (u128)t0 + FQ_MOD[0];
    t0 = (uint64_t)c;
    c = (c >> 64) + t1 + FQ_MOD[1];
    t1 = (uint64_t)c;
    c = (c >> 64) + t2 + FQ_MOD[2];
    t2 = (uint64_t)c;
    c = (c >> 64) + t3 + FQ_MOD[3];
    t3 = (uint64_t)c;
  }
  return Fq4{{t0, t1, t2, t3}};
}
static inline Fq4 fq_dbl(const Fq4& a) { return fq_add(a, a); }
static inline Fq4 fq_mul(const Fq4& a, const Fq4& b) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  for (int i = 0; i < 4; ++i) {
    const uint64_t bi = b.l[i];
    u128 c = (u128)a.l[0] * bi + t0; t0 = (uint64_t)c; c >>= 64;
    c += (u128)a.l[1] * bi + t1; t1 = (uint64_t)c; c >>= 64;
    c += (u128)a.l[2] * bi + t2; t2 = (uint64_t)c; c >>= 64;
    c += (u128)a.l[3] * bi + t3; t3 = (uint64_t)c; c >>= 64;
    c += t4; t4 = (uint64_t)c;
    const uint64_t t5 = (uint64_t)(c >> 64);
    const uint64_t m = t0 * FQ_INV;
    c = (u128)m * FQ_MOD[0] + t0; c >>= 64;
    c += (u128)m * FQ_MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;
    c += (u128)m * FQ_MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;
    c += (u128)m * FQ_MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;
    c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
  }
  return fq_cond_sub(t0, t1, t2, t3, t4);
}
static inline Fq4 fq_sqr(const Fq4& a) { return fq_mul(a, a); }
static inline Fq4 fq_inv(const Fq4& a) {   // a^(p-2)
  uint64_t e[4] = {FQ_MOD[0] - 2, FQ_MOD[1], FQ_MOD[2], FQ_MOD[3]};
  Fq4 acc = FQ_ONE;
  for (int i = 255; i >= 0; --i) {
    acc = fq_sqr(acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = fq_mul(acc, a);
  }
  return acc;
}

struct G1J {          // Jacobian, identity <=> z == 0
  Fq4 x, y, z;
};
static inline G1J g1_identity() { G1J r; memset(&r, 0, sizeof r); return r; }
static inline bool g1_is_identity(const G1J& p) { return fq_is_zero(p.z); }

static inline G1J g1_double(const G1J& p) {      // a = 0
  if (g1_is_identity(p)) return p;
  const Fq4 A = fq_sqr(p.x), B = fq_sqr(p.y), C = fq_sqr(B);
  const Fq4 D = fq_dbl(fq_dbl(fq_mul(p.x, B)));             // 4 X Y^2
  const Fq4 E = fq_add(fq_dbl(A), A);                        // 3 X^2
  G1J r;
  r.x = fq_sub(fq_sqr(E), fq_dbl(D));
  const Fq4 c8 = fq_dbl(fq_dbl(fq_dbl(C)));
  r.y = fq_sub(fq_mul(E, fq_sub(D, r.x)), c8);
  r.z = fq_dbl(fq_mul(p.y, p.z));
  return r;
}
static inline G1J g1_add(const G1J& p, const G1J& q) {
  if (g1_is_identity(p)) return q;
  if (g1_is_identity(q)) return p;
  const Fq4 z1z1 = fq_sqr(p.z), z2z2 = fq_sqr(q.z);
  const Fq4 u1 = fq_mul(p.x, z2z2), u2 = fq_mul(q.x, z1z1);
  const Fq4 s1 = fq_mul(p.y, fq_mul(q.z, z2z2)), s2 = fq_mul(q.y, fq_mul(p.z, z1z1));
  if (fq_eq(u1, u2)) {
    if (fq_eq(s1, s2)) return g1_double(p);
    return g1_identity();
  }
  const Fq4 h = fq_sub(u2, u1);
  const Fq4 i = fq_sqr(fq_dbl(h));
  const Fq4 j = fq_mul(h, i);
  const Fq4 r = fq_dbl(fq_sub(s2, s1));
  const Fq4 v = fq_mul(u1, i);
  G1J o;
  o.x = fq_sub(fq_sub(fq_sqr(r), j), fq_dbl(v));
  o.y = fq_sub(fq_mul(r, fq_sub(v, o.x)), fq_dbl(fq_mul(s1, j)));
  o.z = fq_mul(fq_dbl(fq_mul(p.z, q.z)), h);
  return o;
}
// -> (x, y, 1), or all-zero for the identity
static inline void g1_normalise(const G1J& p, uint64_t out[12], int* is_identity) {
  if (g1_is_identity(p)) {
    memset(out, 0, 96);
    *is_identity = 1;
    return;
  }
  const Fq4 zi = fq_inv(p.z), zi2 = fq_sqr(zi), zi3 = fq_mul(zi2, zi);
  const Fq4 x = fq_mul(p.x, zi2), y = fq_mul(p.y, zi3);
  memcpy(out, x.l, 32);
  memcpy(out + 4, y.l, 32);
  memcpy(out + 8, FQ_ONE.l, 32);
  *is_identity = 0;
}

}  // namespace host
}  // namespace hm
