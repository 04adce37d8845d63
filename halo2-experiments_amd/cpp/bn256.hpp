// bn256.hpp -- host-side BN256 types in the reference's memory layout (halo2curves::bn256, SURVEY.md
// §8a): Fr / Fq = 4 x u64 little-endian Montgomery limbs (radix 2^256, fully reduced),
// G1Affine = {x, y} with (0, 0) the identity, G1 = Jacobian {x, y, z} with z = 0 the identity.
// The arithmetic here is what the HOST side of the path needs (domain constants, SRS scalars,
// checks); the heavy lifting goes through the C ABI of libhalo2_mi355x.so.
#pragma once
#include <cstdint>
#include <cstring>

namespace halo2 {
namespace bn256 {

typedef unsigned __int128 u128;

struct FrParams {
  static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0xc2e1f593efffffffULL;
  static constexpr uint64_t R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
};
struct FqParams {
  static constexpr uint64_t MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0x87d20782e4866389ULL;
  static constexpr uint64_t R[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
};

template <class P>
struct Field {
  uint64_t l[4];

  static Field zero() { return Field{{0, 0, 0, 0}}; }
  static Field one() { return Field{{P::R[0], P::R[1], P::R[2], P::R[3]}}; }
  static Field from_raw(const uint64_t v[4]) {           // canonical integer (< modulus) -> Montgomery
    Field a{{v[0], v[1], v[2], v[3]}}, r2{{P::R2[0], P::R2[1], P::R2[2], P::R2[3]}};
    return a * r2;
  }
  static Field from_u64(uint64_t v) {
    const uint64_t raw[4] = {v, 0, 0, 0};
    return from_raw(raw);
  }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Field& o) const { return std::memcmp(l, o.l, 32) == 0; }
  bool operator!=(const Field& o) const { return !(*this == o); }

  static Field cond_sub(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3, uint64_t carry) {
    u128 b = (u128)t0 - P::MOD[0];
    const uint64_t r0 = (uint64_t)b;
    b = (u128)t1 - P::MOD[1] - ((uint64_t)(b >> 64) & 1);
    const uint64_t r1 = (uint64_t)b;
    b = (u128)t2 - P::MOD[2] - ((uint64_t)(b >> 64) & 1);
    const uint64_t r2 = (uint64_t)b;
    b = (u128)t3 - P::MOD[3] - ((uint64_t)(b >> 64) & 1);
    const uint64_t r3 = (uint64_t)b;
    const bool ge = carry || !((uint64_t)(b >> 64) & 1);
    return ge ? Field{{r0, r1, r2, r3}} : Field{{t0, t1, t2, t3}};
  }
  Field operator+(const Field& o) const {
    u128 c = (u128)l[0] + o.l[0];
    const uint64_t t0 = (uint64_t)c;
    c = (c >> 64) + l[1] + o.l[1];
    const uint64_t t1 = (uint64_t)c;
    c = (c >> 64) + l[2] + o.l[2];
    const uint64_t t2 = (uint64_t)c;
    c = (c >> 64) + l[3] + o.l[3];
    return cond_sub(t0, t1, t2, (uint64_t)c, (uint64_t)(c >> 64));
  }
  Field operator-(const Field& o) const {
    u128 d = (u128)l[0] - o.l[0];
    uint64_t t0 = (uint64_t)d;
    d = (u128)l[1] - o.l[1] - ((uint64_t)(d >> 64) & 1);
    uint64_t t1 = (uint64_t)d;
    d = (u128)l[2] - o.l[2] - ((uint64_t)(d >> 64) & 1);
    uint64_t t2 = (uint64_t)d;
    d = (u128)l[3] - o.l[3] - ((uint64_t)(d >> 64) & 1);
    uint64_t t3 = (uint64_t)d;
    if ((uint64_t)(d >> 64) & 1) {
      u128 c = (u128)t0 + P::MOD[0];
      t0 = (uint64_t)c;
      c = (c >> 64) + t1 + P::MOD[1];
      t1 = (uint64_t)c;
      c = (c >> 64) + t2 + P::MOD[2];
      t2 = (uint64_t)c;
      c = (c >> 64) + t3 + P::MOD[3];
      t3 = (uint64_t)c;
    }
    return Field{{t0, t1, t2, t3}};
  }
  Field operator-() const { return zero() - *this; }
  Field operator*(const Field& o) const {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    for (int i = 0; i < 4; ++i) {
      const uint64_t bi = o.l[i];
      u128 c = (u128)l[0] * bi + t0; t0 = (uint64_t)c; c >>= 64;
      c += (u128)l[1] * bi + t1; t1 = (uint64_t)c; c >>= 64;
      c += (u128)l[2] * bi + t2; t2 = (uint64_t)c; c >>= 64;
      c += (u128)l[3] * bi + t3; t3 = (uint64_t)c; c >>= 64;
      c += t4; t4 = (uint64_t)c;
      const uint64_t t5 = (uint64_t)(c >> 64);
      const uint64_t m = t0 * P::INV;
      c = (u128)m * P::MOD[0] + t0; c >>= 64;
      c += (u128)m * P::MOD[1] + t1; t0 = (uint64_t)c; c >>= 64;
      c += (u128)m * P::MOD[2] + t2; t1 = (uint64_t)c; c >>= 64;
      c += (u128)m * P::MOD[3] + t3; t2 = (uint64_t)c; c >>= 64;
      c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
    }
    return cond_sub(t0, t1, t2, t3, t4);
  }
  Field square() const { return *this * *this; }
  Field pow(const uint64_t e[4]) const {
    Field acc = one();
    for (int i = 255; i >= 0; --i) {
      acc = acc.square();
      if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
    }
    return acc;
  }
  Field pow_u64(uint64_t e) const {
    const uint64_t ee[4] = {e, 0, 0, 0};
    return pow(ee);
  }
  Field invert() const {   // a^(m-2); callers never pass zero
    const uint64_t e[4] = {P::MOD[0] - 2, P::MOD[1], P::MOD[2], P::MOD[3]};
    return pow(e);
  }
};

typedef Field<FrParams> Fr;
typedef Field<FqParams> Fq;

struct G1Affine {
  Fq x, y;
  static G1Affine identity() { return G1Affine{Fq::zero(), Fq::zero()}; }
  static G1Affine generator() { return G1Affine{Fq::from_u64(1), Fq::from_u64(2)}; }
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
  bool operator==(const G1Affine& o) const { return x == o.x && y == o.y; }
};
struct G1 {
  Fq x, y, z;
  static G1 identity() { return G1{Fq::zero(), Fq::zero(), Fq::zero()}; }
  bool is_identity() const { return z.is_zero(); }
};
static_assert(sizeof(Fr) == 32 && sizeof(G1Affine) == 64 && sizeof(G1) == 96, "layout must match halo2curves");

// Fq2 = Fq[u] / (u^2 + 1) and G2 (y^2 = x^3 + 3 / (9 + u)), host only: ParamsKZG::setup needs ONE scalar
// multiple of the G2 generator (s_g2); layout of G2Affine as halo2curves holds it: x.c0, x.c1, y.c0, y.c1.
struct Fq2 {
  Fq c0, c1;
  static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
  Fq2 operator+(const Fq2& o) const { return Fq2{c0 + o.c0, c1 + o.c1}; }
  Fq2 operator-(const Fq2& o) const { return Fq2{c0 - o.c0, c1 - o.c1}; }
  Fq2 operator*(const Fq2& o) const { return Fq2{c0 * o.c0 - c1 * o.c1, c0 * o.c1 + c1 * o.c0}; }
  Fq2 square() const { return *this * *this; }
  Fq2 invert() const {
    const Fq d = (c0.square() + c1.square()).invert();
    return Fq2{c0 * d, -(c1 * d)};
  }
};
struct G2Affine {
  Fq2 x, y;                                   // the identity is (0, 0)
  static G2Affine identity() { return G2Affine{Fq2::zero(), Fq2::zero()}; }
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
  bool operator==(const G2Affine& o) const { return x == o.x && y == o.y; }
  static G2Affine generator() {               // the alt_bn128 G2 generator (EIP-197)
    const uint64_t x0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
    const uint64_t x1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
    const uint64_t y0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
    const uint64_t y1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
    return G2Affine{Fq2{Fq::from_raw(x0), Fq::from_raw(x1)}, Fq2{Fq::from_raw(y0), Fq::from_raw(y1)}};
  }
  G2Affine add(const G2Affine& q) const {     // affine law with one inversion: setup does ~380 of these
    if (is_identity()) return q;
    if (q.is_identity()) return *this;
    Fq2 lam;
    if (x == q.x) {
      if (!(y == q.y) || y.is_zero()) return identity();
      const Fq2 xx = x.square();
      lam = (xx + xx + xx) * (y + y).invert();
    } else {
      lam = (q.y - y) * (q.x - x).invert();
    }
    const Fq2 x3 = lam.square() - x - q.x;
    return G2Affine{x3, lam * (x - x3) - y};
  }
  G2Affine mul(const Field<FrParams>& k_mont) const {
    const Field<FrParams> one_raw{{1, 0, 0, 0}};
    const Field<FrParams> k = k_mont * one_raw;             // Montgomery -> canonical
    G2Affine acc = identity();
    for (int i = 255; i >= 0; --i) {
      acc = acc.add(acc);
      if ((k.l[i >> 6] >> (i & 63)) & 1) acc = acc.add(*this);
    }
    return acc;
  }
};
static_assert(sizeof(G2Affine) == 128, "layout must match halo2curves");

// Fr constants of halo2curves::bn256 (FieldExt / PrimeField): S, ROOT_OF_UNITY = 7^((r-1)/2^28), ZETA
constexpr uint32_t FR_S = 28;
inline Fr fr_root_of_unity() {
  uint64_t e[4];   // (r - 1) >> 28
  const uint64_t m[4] = {FrParams::MOD[0] - 1, FrParams::MOD[1], FrParams::MOD[2], FrParams::MOD[3]};
  for (int i = 0; i < 4; ++i) e[i] = (m[i] >> 28) | (i < 3 ? (m[i + 1] << 36) : 0);
  return Fr::from_u64(7).pow(e);
}
inline Fr fr_zeta() {
  const uint64_t z[4] = {0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL, 0x048b6e193fd84104ULL, 0x30644e72e131a029ULL};
  return Fr::from_raw(z);
}

}  // namespace bn256
}  // namespace halo2
