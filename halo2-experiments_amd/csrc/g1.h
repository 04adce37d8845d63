// g1.h -- BN256 G1 (y^2 = x^3 + 3 over Fq) in Jacobian coordinates on the ff29 field layer.
//
// Coordinates are kept in the internal Montgomery form of ff29.h.  "Class" bounds (checked for
// every formula by the HM_BOUNDS host build, see tests/test_ff29_host.py):
//     Jacobian point:  X, Y normalised limbs, X < HM_G1_XB * p, Y < HM_G1_YB * p;  Z a product output (< 2p)
//     affine operand:  x, y product outputs (< 2p)
// Identity is carried as an explicit flag (never encoded in Z) so the hot loop has no compare
// against zero; the exceptional cases of the addition law (P = Q, P = -Q) are detected from
// products the formula computes anyway.
//
// Formulas: madd-2007-bl / add-2007-bl / dbl-2009-l from the Explicit-Formulas Database, the
// textbook laws for a = 0 short-Weierstrass curves, re-associated so that no stored coordinate
// needs a modular reduction (only carry normalisation).
#pragma once
#include "ff29.h"

namespace hm {

using Fq = Fe<FqParams>;
using Fr = Fe<FrParams>;

struct G1Aff {
  Fq x, y;
};
struct G1Jac {
  Fq x, y, z;
  bool inf;
};

// class bounds on stored X and Y (multiples of p); the subtraction constants below are chosen against them
#define HM_G1_XB 12.0
#define HM_G1_YB 5.0
#define HM_G1_ZB 2.0

#ifdef HM_BOUNDS
inline void g1_check_class(const G1Jac& p, const char* what) {
  HM_CHECK(p.x.lb <= MASK29 && p.y.lb <= MASK29 && p.z.lb <= MASK29, what);
  HM_CHECK(p.x.vb <= HM_G1_XB && p.y.vb <= HM_G1_YB && p.z.vb <= HM_G1_ZB, what);
}
#define HM_G1_CHECK(p, what) g1_check_class(p, what)
#else
#define HM_G1_CHECK(p, what) ((void)0)
#endif

HM_HD G1Jac g1_identity() {
  G1Jac r;
  r.x = fe_zero<FqParams>();
  r.y = fe_zero<FqParams>();
  r.z = fe_zero<FqParams>();
  r.inf = true;
  return r;
}

HM_HD G1Jac g1_from_affine(const G1Aff& p) {
  G1Jac r;
  r.x = p.x;
  r.y = p.y;
  r.z = fe_one<FqParams>();
  r.inf = false;
  return r;
}

// y -> -y for an affine operand (value < 2p): 3p - y, renormalised
HM_HD G1Aff g1_neg_affine(const G1Aff& p) {
  G1Aff r;
  r.x = p.x;
  r.y = fe_norm(fe_sub<3, 29>(fe_zero<FqParams>(), p.y));
  return r;
}
// Doubling for a = 0 (dbl-2009-l; D = 4*X*Y^2 taken as a product, and X3, Y3 produced by fused
// products so that they are reduction outputs): p must not be the identity.
HM_HD G1Jac g1_double_nz(const G1Jac& p) {
  HM_G1_CHECK(p, "g1_double_nz input outside class");
  const Fq one = fe_one<FqParams>();
  const Fq A = fe_sqr(p.x);                                  // X^2
  const Fq B = fe_sqr(p.y);                                  // Y^2
  const Fq C = fe_sqr(B);                                    // Y^4
  const Fq S = fe_mul(p.x, B);                               // X*Y^2
  const Fq D = fe_norm(fe_mul4(S));                          // D = 4*X*Y^2
  const Fq E = fe_norm(fe_add(fe_dbl(A), A));                // E = 3*X^2
  const Fq n2d = fe_sub<10, 30>(fe_zero<FqParams>(), fe_dbl(D));   // -2D  (limbs < 2^31)
  G1Jac r;
  r.x = fe_mul2(E, E, n2d, one);                             // X3 = E^2 - 2D
  const Fq dx = fe_norm(fe_sub<3, 29>(D, r.x));              // D - X3
  const Fq c8 = fe_dbl(fe_norm(fe_mul4(C)));                 // 8*Y^4 (limbs < 2^30)
  const Fq n8c = fe_sub<9, 30>(fe_zero<FqParams>(), c8);     // -8C
  r.y = fe_mul2(E, dx, n8c, one);                            // Y3 = E(D - X3) - 8C
  r.z = fe_mul(fe_dbl(p.y), p.z);                            // Z3 = 2*Y*Z
  r.inf = false;
  return r;
}

HM_HD G1Jac g1_double(const G1Jac& p) {
  if (p.inf) return p;
  return g1_double_nz(p);
}

// madd-2007-bl re-associated: acc (Jacobian, class bounds) + (neg ? -q : q), q affine with product-
// output coordinates.  7M + 3S + one fused double product; the sign is applied to S2 = y2*Z1^3
// (the only place y2 enters), so negating the operand costs 18 instructions instead of a
// subtraction and a renormalisation.  Neither operand may be the identity; the result may be.
HM_HD G1Jac g1_madd_nz(const G1Jac& p, const G1Aff& q, bool neg = false) {
  HM_G1_CHECK(p, "g1_madd_nz input outside class");
  const Fq z1z1 = fe_sqr(p.z);
  const Fq u2 = fe_mul(q.x, z1z1);
  const Fq s2p = fe_mul(q.y, fe_mul(p.z, z1z1));
  Fq s2;                                                      // +-S2, limbs <= 2^30
  {
    const Fq s2n = fe_sub<3, 29>(fe_zero<FqParams>(), s2p);
#pragma unroll
    for (int i = 0; i < 9; ++i) s2.l[i] = neg ? s2n.l[i] : s2p.l[i];
#ifdef HM_BOUNDS
    s2.vb = s2n.vb; s2.lb = s2n.lb; s2.tb = s2n.tb;
#endif
  }
  const Fq h = fe_norm(fe_sub<13, 29>(u2, p.x));             // U2 - X1
  const Fq hh = fe_sqr(h);
  if (fe_is_zero_mod(hh)) {                                  // X1 = U2: doubling or inverse
    const Fq r0 = fe_norm(fe_sub<6, 29>(s2, p.y));
    const Fq rr0 = fe_sqr(r0);
    if (fe_is_zero_mod(rr0)) return g1_double_nz(p);
    return g1_identity();
  }
  const Fq i4 = fe_mul4(hh);                                 // I = 4*HH   (limbs < 2^31)
  const Fq j = fe_mul(h, i4);
  const Fq r0 = fe_norm(fe_sub<6, 29>(s2, p.y));             // +-S2 - Y1
  const Fq r = fe_dbl(r0);                                   // r = 2(S2 - Y1)  (limbs < 2^30)
  const Fq v = fe_mul(p.x, i4);
  const Fq rr = fe_sqr(r);
  const Fq t2 = fe_add(j, fe_dbl(v));                        // J + 2V  (limbs < 3*2^29)
  G1Jac o;
  o.x = fe_norm(fe_sub<6, 31>(rr, t2));                      // r^2 - J - 2V
  const Fq vx = fe_norm(fe_sub<13, 29>(v, o.x));             // V - X3
  const Fq ny2 = fe_dbl(fe_sub<6, 29>(fe_zero<FqParams>(), p.y));   // -2*Y1 (limbs < 2^31)
  o.y = fe_mul2(r, vx, ny2, j);                              // r (V - X3) - 2 Y1 J
  o.z = fe_mul(p.z, fe_dbl(h));                              // 2 Z1 H
  o.inf = false;
  return o;
}

HM_HD G1Jac g1_madd(const G1Jac& p, const G1Aff& q, bool neg = false) {
  if (p.inf) return g1_from_affine(neg ? g1_neg_affine(q) : q);
  return g1_madd_nz(p, q, neg);
}

// add-2007-bl re-associated (11M + 4S + one fused double product): both Jacobian, neither the identity.
HM_HD G1Jac g1_add_nz(const G1Jac& p, const G1Jac& q) {
  HM_G1_CHECK(p, "g1_add_nz input p outside class");
  HM_G1_CHECK(q, "g1_add_nz input q outside class");
  const Fq z1z1 = fe_sqr(p.z);
  const Fq z2z2 = fe_sqr(q.z);
  const Fq u1 = fe_mul(p.x, z2z2);
  const Fq u2 = fe_mul(q.x, z1z1);
  const Fq s1 = fe_mul(p.y, fe_mul(q.z, z2z2));
  const Fq s2 = fe_mul(q.y, fe_mul(p.z, z1z1));
  const Fq h = fe_norm(fe_sub<3, 29>(u2, u1));
  const Fq hh = fe_sqr(h);
  if (fe_is_zero_mod(hh)) {
    const Fq r0 = fe_norm(fe_sub<3, 29>(s2, s1));
    const Fq rr0 = fe_sqr(r0);
    if (fe_is_zero_mod(rr0)) return g1_double_nz(p);
    return g1_identity();
  }
  const Fq i4 = fe_mul4(hh);
  const Fq j = fe_mul(h, i4);
  const Fq r0 = fe_norm(fe_sub<3, 29>(s2, s1));
  const Fq r = fe_dbl(r0);
  const Fq v = fe_mul(u1, i4);
  const Fq rr = fe_sqr(r);
  const Fq t2 = fe_add(j, fe_dbl(v));
  G1Jac o;
  o.x = fe_norm(fe_sub<4, 31>(rr, t2));
  const Fq vx = fe_norm(fe_sub<7, 29>(v, o.x));
  const Fq ns1 = fe_dbl(fe_sub<3, 29>(fe_zero<FqParams>(), s1));   // -2*S1 (limbs < 2^31)
  o.y = fe_mul2(r, vx, ns1, j);                              // r (V - X3) - 2 S1 J
  o.z = fe_mul(fe_mul(p.z, q.z), fe_dbl(h));                 // 2 Z1 Z2 H
  o.inf = false;
  return o;
}

HM_HD G1Jac g1_add(const G1Jac& p, const G1Jac& q) {
  if (p.inf) return q;
  if (q.inf) return p;
  return g1_add_nz(p, q);
}

// ---------------------------------------------------------------------------------------------
// Extended Jacobian ("XYZZ") accumulator for the bucket chains: x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.
// The mixed addition (madd-2008-s) needs no Z1^2 / Z1^3 recomputation, so it is 6M + 2S + one
// fused double product -- one square (~170 instructions) less than the Jacobian form -- at the
// price of a fourth stored coordinate.  Class bounds: X < HM_XYZZ_XB*p, Y < HM_XYZZ_YB*p
// (normalised limbs), ZZ and ZZZ product outputs (< 2p).
// ---------------------------------------------------------------------------------------------
struct G1Xyzz {
  Fq x, y, zz, zzz;
  bool inf;
};
#define HM_XYZZ_XB 8.0
#define HM_XYZZ_YB 3.0

#ifdef HM_BOUNDS
inline void g1_check_class(const G1Xyzz& p, const char* what) {
  HM_CHECK(p.x.lb <= MASK29 && p.y.lb <= MASK29 && p.zz.lb <= MASK29 && p.zzz.lb <= MASK29, what);
  HM_CHECK(p.x.vb <= HM_XYZZ_XB && p.y.vb <= HM_XYZZ_YB && p.zz.vb <= 2.0 && p.zzz.vb <= 2.0, what);
}
#endif

HM_HD G1Xyzz g1x_identity() {
  G1Xyzz r;
  r.x = fe_zero<FqParams>();
  r.y = fe_zero<FqParams>();
  r.zz = fe_zero<FqParams>();
  r.zzz = fe_zero<FqParams>();
  r.inf = true;
  return r;
}

HM_HD G1Xyzz g1x_from_affine(const G1Aff& p) {
  G1Xyzz r;
  r.x = p.x;
  r.y = p.y;
  r.zz = fe_one<FqParams>();
  r.zzz = fe_one<FqParams>();
  r.inf = false;
  return r;
}

// (X, Y, ZZ, ZZZ) -> Jacobian (X*ZZ^2, Y*ZZZ^2, ZZZ): every output a product, so inside the Jacobian class
HM_HD G1Jac g1x_to_jac(const G1Xyzz& p) {
  G1Jac r;
  if (p.inf) return g1_identity();
  HM_G1_CHECK(p, "g1x_to_jac input outside class");
  r.x = fe_mul(p.x, fe_sqr(p.zz));
  r.y = fe_mul(p.y, fe_sqr(p.zzz));
  r.z = p.zzz;
  r.inf = false;
  return r;
}

HM_HD G1Xyzz g1x_from_jac(const G1Jac& p) {
  G1Xyzz r;
  if (p.inf) return g1x_identity();
  HM_G1_CHECK(p, "g1x_from_jac input outside class");
  r.zz = fe_sqr(p.z);
  r.zzz = fe_mul(p.z, r.zz);
  // X, Y: a product by ONE keeps the residue and brings a Jacobian-class value (X < 12p, Y < 5p)
  // under the tighter XYZZ bounds (rare path: only after an in-bucket doubling)
  r.x = fe_mul(p.x, fe_one<FqParams>());
  r.y = fe_mul(p.y, fe_one<FqParams>());
  r.inf = false;
  return r;
}

// acc += (neg ? -q : q) by the ordinary law, for acc != identity.  Returns false and leaves acc
// untouched in the exceptional case X1 = U2 (a repeated base in one bucket, or a base and its
// negative): the caller finishes such a chain with the general Jacobian law.  Keeping that case OUT
// of this function keeps the hot loop free of its code and of the register merge of its result.
HM_HD bool g1x_madd_fast(G1Xyzz& acc, const G1Aff& q, bool neg) {
  HM_G1_CHECK(acc, "g1x_madd_fast input outside class");
  const Fq u2 = fe_mul(q.x, acc.zz);
  const Fq pp0 = fe_norm(fe_sub<9, 29>(u2, acc.x));          // P = U2 - X1
  const Fq pp = fe_sqr(pp0);                                  // PP
  if (fe_is_zero_mod(pp)) return false;
  const Fq s2p = fe_mul(q.y, acc.zzz);
  Fq s2;                                                      // +-S2, limbs <= 2^30
  {
    const Fq s2n = fe_sub<3, 29>(fe_zero<FqParams>(), s2p);
#pragma unroll
    for (int i = 0; i < 9; ++i) s2.l[i] = neg ? s2n.l[i] : s2p.l[i];
#ifdef HM_BOUNDS
    s2.vb = s2n.vb; s2.lb = s2n.lb; s2.tb = s2n.tb;
#endif
  }
  const Fq r = fe_norm(fe_sub<4, 29>(s2, acc.y));             // R = +-S2 - Y1
  const Fq ppp = fe_mul(pp0, pp);
  const Fq qq = fe_mul(acc.x, pp);                            // Q = X1 * PP
  const Fq rr = fe_sqr(r);
  const Fq t2 = fe_add(ppp, fe_dbl(qq));                      // PPP + 2Q  (limbs < 3*2^29)
  const Fq x3 = fe_norm(fe_sub<6, 31>(rr, t2));               // R^2 - PPP - 2Q
  const Fq vx = fe_sub<9, 29>(qq, x3);                        // Q - X3, left lazy (limbs < 2^30.6): only a multiplicand of the fused product
  const Fq ny = fe_sub<4, 29>(fe_zero<FqParams>(), acc.y);    // -Y1 (limbs < 2^30)
  acc.y = fe_mul2(r, vx, ny, ppp);                            // R (Q - X3) - Y1 PPP
  acc.x = x3;
  acc.zz = fe_mul(acc.zz, pp);
  acc.zzz = fe_mul(acc.zzz, ppp);
  return true;
}

// general form (host checks, cold paths): identity operands and the exceptional cases included
HM_HD G1Xyzz g1x_madd(const G1Xyzz& p, const G1Aff& q, bool neg = false) {
  if (p.inf) return g1x_from_affine(neg ? g1_neg_affine(q) : q);
  G1Xyzz r = p;
  if (g1x_madd_fast(r, q, neg)) return r;
  return g1x_from_jac(g1_madd_nz(g1x_to_jac(p), q, neg));
}

}  // namespace hm
