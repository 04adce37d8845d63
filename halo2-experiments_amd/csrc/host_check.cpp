// host_check.cpp -- CPU build of the device arithmetic headers (ff29.h, g1.h) with worst-case
// bound tracking (-DHM_BOUNDS).  Not part of the product path: it exists so that the `-m "not gpu"`
// tests can (a) compare the exact code the kernels run with the oracle and (b) prove, by running
// every formula once with its inputs declared at the class bounds, that no 64-bit column sum,
// limb or lazy value can overflow for ANY input (the tracked bounds are data-independent).
//
// Build: g++ -O2 -std=c++17 -DHM_BOUNDS -shared -fPIC -o libhm_hostcheck.so host_check.cpp
#include <cstring>

#include "g1.h"

using namespace hm;

namespace {

template <class F>
Fe<F> load_ext(const uint64_t* p) {
  uint32_t w[8];
  std::memcpy(w, p, 32);
  return fe_from_ext<F>(w);
}
template <class F>
void store_ext(uint64_t* p, const Fe<F>& a) {
  uint32_t w[8];
  fe_to_ext(w, a);
  std::memcpy(p, w, 32);
}

template <class F>
void force_bounds(Fe<F>& a, double vb) {  // declare the class maximum regardless of the actual value
  a.vb = vb;
  a.lb = MASK29;
  a.tb = top_bound_from_value<F>(vb);
}

G1Jac load_jac(const uint64_t* p, int inf) {
  G1Jac r;
  r.x = load_ext<FqParams>(p);
  r.y = load_ext<FqParams>(p + 4);
  r.z = load_ext<FqParams>(p + 8);
  r.inf = inf != 0;
  return r;
}
void store_jac(uint64_t* p, int* inf, const G1Jac& r) {
  if (r.inf) {
    std::memset(p, 0, 96);
    *inf = 1;
    return;
  }
  store_ext(p, r.x);
  store_ext(p + 4, r.y);
  store_ext(p + 8, r.z);
  *inf = 0;
}
void at_class_max(G1Jac& p) {
  force_bounds(p.x, HM_G1_XB);
  force_bounds(p.y, HM_G1_YB);
  force_bounds(p.z, HM_G1_ZB);
}

template <class F>
void field_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    Fe<F> x = load_ext<F>(a + 4 * i), y = load_ext<F>(b + 4 * i), r;
    switch (op) {
      case 0: r = fe_mul(x, y); break;
      case 1: r = fe_sqr(x); break;
      case 2: r = fe_norm(fe_add(x, y)); break;
      case 3: r = fe_norm(fe_sub<3, 29>(x, y)); break;
      case 4: {  // lazy chain: ((x + y) * (x - y + 3p)) + x*x, exercises unnormalised operands
        Fe<F> s = fe_add(x, y), d = fe_sub<3, 29>(x, y);
        r = fe_norm(fe_add(fe_mul(s, fe_norm(d)), fe_sqr(x)));
        break;
      }
      case 5: {  // x + 40*y accumulated lazily (value up to ~82 MOD), then the cheap reduction
        Fe<F> acc = x;
        for (int k = 0; k < 40; ++k) acc = fe_norm(fe_add(acc, y));
        r = fe_reduce_small(acc);
        break;
      }
      default: r = x;
    }
    store_ext(o + 4 * i, r);
  }
}

}  // namespace

extern "C" {

// field: 0 = Fq, 1 = Fr.  op: 0 mul, 1 sqr, 2 add, 3 sub, 4 lazy chain.  external Montgomery in/out
void hc_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
  if (field == 0) field_op<FqParams>(op, a, b, o, n);
  else field_op<FrParams>(op, a, b, o, n);
}

// unpack/pack round trip and canonicalisation of raw (possibly non-canonical) 256-bit words
void hc_fr_reduce_raw(const uint64_t* a, uint64_t* o, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t w[8];
    std::memcpy(w, a + 4 * i, 32);
    Fe<FrParams> x = fe_unpack<FrParams>(w);
    // x * 2^261 * 2^-261 = x mod r, canonical
    Fe<FrParams> y = fe_canonical(fe_mul(x, fe_const<FrParams>(FrParams::ONE)));
    fe_pack(w, y);
    std::memcpy(o + 4 * i, w, 32);
  }
}

void hc_g1_madd(const uint64_t* pj, int pinf, const uint64_t* q_aff, int neg, uint64_t* out, int* out_inf) {
  G1Jac p = load_jac(pj, pinf);
  G1Aff q;
  q.x = load_ext<FqParams>(q_aff);
  q.y = load_ext<FqParams>(q_aff + 4);
  store_jac(out, out_inf, g1_madd(p, q, neg != 0));
}
void hc_g1_add(const uint64_t* pj, int pinf, const uint64_t* qj, int qinf, uint64_t* out, int* out_inf) {
  store_jac(out, out_inf, g1_add(load_jac(pj, pinf), load_jac(qj, qinf)));
}
void hc_g1_double(const uint64_t* pj, int pinf, uint64_t* out, int* out_inf) {
  store_jac(out, out_inf, g1_double(load_jac(pj, pinf)));
}
// k chained madds of the same affine point followed by doublings: stresses class closure on data
void hc_g1_chain(const uint64_t* q_aff, int k, int dbl, uint64_t* out, int* out_inf) {
  G1Aff q;
  q.x = load_ext<FqParams>(q_aff);
  q.y = load_ext<FqParams>(q_aff + 4);
  G1Jac acc = g1_identity();
  for (int i = 0; i < k; ++i) acc = g1_madd(acc, q);
  for (int i = 0; i < dbl; ++i) acc = g1_double(acc);
  store_jac(out, out_inf, acc);
}

// the bucket accumulator's XYZZ chain: k mixed additions of q (alternating sign pattern given by the
// bits of `signs`), optionally starting from a doubling-triggering repeat, converted to Jacobian
void hc_g1x_chain(const uint64_t* q_affs, int k, uint64_t signs, uint64_t* out, int* out_inf) {
  G1Xyzz acc = g1x_identity();
  for (int i = 0; i < k; ++i) {
    G1Aff q;
    q.x = load_ext<FqParams>(q_affs + 8 * i);
    q.y = load_ext<FqParams>(q_affs + 8 * i + 4);
    acc = g1x_madd(acc, q, ((signs >> (i & 63)) & 1) != 0);
  }
  store_jac(out, out_inf, g1x_to_jac(acc));
}

// XYZZ class closure: madd with the accumulator DECLARED at its class maxima must land inside the
// class again, and its conversion to Jacobian inside the Jacobian class.
int hc_xyzz_bounds_closure(const uint64_t* pj, const uint64_t* q_aff, double* report) {
  G1Jac pjac = load_jac(pj, 0);
  G1Xyzz p = g1x_from_jac(pjac);
  force_bounds(p.x, HM_XYZZ_XB);
  force_bounds(p.y, HM_XYZZ_YB);
  force_bounds(p.zz, 2.0);
  force_bounds(p.zzz, 2.0);
  G1Aff q;
  q.x = load_ext<FqParams>(q_aff);
  q.y = load_ext<FqParams>(q_aff + 4);
  force_bounds(q.x, 2.0);
  force_bounds(q.y, 2.0);
  G1Xyzz r = p;
  if (!g1x_madd_fast(r, q, true)) return 0;
  const G1Jac j = g1x_to_jac(p);
  report[0] = r.x.vb; report[1] = r.y.vb; report[2] = r.zz.vb; report[3] = r.zzz.vb;
  report[4] = j.x.vb; report[5] = j.y.vb; report[6] = j.z.vb;
  int ok = 1;
  if (r.x.vb > HM_XYZZ_XB || r.y.vb > HM_XYZZ_YB || r.zz.vb > 2.0 || r.zzz.vb > 2.0) ok = 0;
  if (j.x.vb > HM_G1_XB || j.y.vb > HM_G1_YB || j.z.vb > HM_G1_ZB) ok = 0;
  return ok;
}

// Run every curve formula with inputs DECLARED at the class maxima; any precondition violation
// aborts.  Writes the resulting output bounds (x.vb, y.vb, z.vb per formula) for the report.
int hc_bounds_closure(const uint64_t* pj, const uint64_t* qj, const uint64_t* q_aff, double* report) {
  G1Jac p = load_jac(pj, 0), q2 = load_jac(qj, 0);
  at_class_max(p);
  at_class_max(q2);
  G1Aff q;
  q.x = load_ext<FqParams>(q_aff);
  q.y = load_ext<FqParams>(q_aff + 4);
  force_bounds(q.x, 2.0);
  force_bounds(q.y, 2.0);
  G1Jac r[3] = {g1_madd_nz(p, q, true), g1_add_nz(p, q2), g1_double_nz(p)};
  int ok = 1;
  for (int i = 0; i < 3; ++i) {
    report[3 * i + 0] = r[i].x.vb;
    report[3 * i + 1] = r[i].y.vb;
    report[3 * i + 2] = r[i].z.vb;
    if (r[i].x.vb > HM_G1_XB || r[i].y.vb > HM_G1_YB || r[i].z.vb > HM_G1_ZB) ok = 0;
    if (r[i].x.lb > MASK29 || r[i].y.lb > MASK29 || r[i].z.lb > MASK29) ok = 0;
  }
  return ok;
}

// The Fr vector formulas of ntt.hip / poly.hip / polyops.hip, each run once with its inputs DECLARED at the class
// maxima the kernels rely on (raw 256-bit words: value < 2^256; products: < 2r; reduced sums: < 3r).  Any
// precondition violation aborts; returns 1 when every result is inside the class its consumer expects.
// The GraphEvaluator's lazy classes (graph.hip: GE_CAP = 16, GE_COLUMN_BOUND = 6): every stored intermediate is a
// normalised element of value < 16 r.  Run every operation the interpreter has on operands DECLARED at the class
// maxima (the bound tracking asserts each primitive's precondition) and check the outputs are back inside the class.
int hc_graph_bounds_closure(const uint64_t* a_ext, double* report) {
  typedef Fe<FrParams> F;
  F x = load_ext<FrParams>(a_ext);
  const double CAP = 16.0;
  int ok = 1, r = 0;
  F c16 = x, c16b = x, c3 = x, col = x;
  force_bounds(c16, CAP);
  force_bounds(c16b, CAP);
  force_bounds(c3, 3.0);
  force_bounds(col, 6.0);
  // products of two class-maximum operands (and of column words) come out below 3r
  F m = fe_mul(c16, c16b);
  report[r++] = m.vb;
  ok &= m.vb <= 3.0;
  F q = fe_sqr(c16);
  ok &= q.vb <= 3.0;
  ok &= fe_mul(col, c16).vb <= 3.0;
  // lazy sum at the cap: 8 + 8 -> norm only; a sum past the cap is reduced (16 + 16)
  F h8 = x;
  force_bounds(h8, 8.0);
  F s16 = fe_norm(fe_add(h8, h8));
  report[r++] = s16.vb;
  ok &= s16.vb <= CAP;
  F red = fe_reduce_small(fe_norm(fe_add(c16, c16b)));
  ok &= red.vb <= 3.0;
  F dbl = fe_reduce_small(fe_norm(fe_dbl(c16)));
  ok &= dbl.vb <= 3.0;
  // narrow subtraction (subtrahend < 3r): lazy while minuend + 4 <= cap, reduced above
  F d12 = x;
  force_bounds(d12, 12.0);
  F sub_lazy = fe_norm(fe_sub<4, 29>(d12, c3));
  report[r++] = sub_lazy.vb;
  ok &= sub_lazy.vb <= CAP;
  ok &= fe_reduce_small(fe_norm(fe_sub<4, 29>(c16, c3))).vb <= 3.0;
  // wide subtraction (subtrahend up to the cap), always reduced; negation likewise
  F sub_wide = fe_reduce_small(fe_norm(fe_sub<20, 29>(c16, c16b)));
  report[r++] = sub_wide.vb;
  ok &= sub_wide.vb <= 3.0;
  ok &= fe_reduce_small(fe_norm(fe_sub<20, 29>(fe_zero<FrParams>(), c16))).vb <= 3.0;
  ok &= fe_norm(fe_sub<4, 29>(fe_zero<FrParams>(), c3)).vb <= CAP;
  // Horner step a * b + c: lazy while 3 + c <= cap
  F d13 = x;
  force_bounds(d13, 13.0);
  F ma = fe_norm(fe_add(fe_mul(c16, c16b), d13));
  report[r++] = ma.vb;
  ok &= ma.vb <= CAP;
  ok &= fe_reduce_small(fe_norm(fe_add(fe_mul(c16, c16b), c16))).vb <= 3.0;
  // the result leaves through a reduction, the conversion product and the canonical form
  (void)fe_canonical(fe_mul(fe_reduce_small(fe_norm(c16)), fe_const<FrParams>(FrParams::INT2EXT)));
  return ok;
}

int hc_fr_vector_bounds_closure(const uint64_t* a_ext, const uint64_t* b_ext, double* report) {
  typedef Fe<FrParams> F;
  uint32_t wa[8], wb[8];
  std::memcpy(wa, a_ext, 32);
  std::memcpy(wb, b_ext, 32);
  F raw = fe_unpack<FrParams>(wa);                 // declared: any 256-bit word pattern
  raw.l[8] = (1u << 24) - 1;                       // and take the largest one for the run itself
  for (int i = 0; i < 8; ++i) raw.l[i] = MASK29;
  F z = load_ext<FrParams>(b_ext);                 // a product output
  force_bounds(z, 2.0);
  F k32 = fe_const<FrParams>(FrParams::EXT2INT);   // any canonical constant
  int ok = 1, r = 0;
  // 1. Horner step of eval_polynomial / kate_division: acc <- acc * z + raw, iterated at its fixed point
  F acc = fe_add(fe_mul(raw, z), raw);
  for (int k = 0; k < 4; ++k) acc = fe_add(fe_mul(acc, z), raw);
  report[r++] = acc.vb;
  F s = fe_reduce_small(fe_norm(acc));
  ok &= s.vb <= 3.0;
  // 2. scan step with a uniform multiplier: v <- reduce(v + w * y), v, w < 3r, y < 2r
  F v = s, w = s;
  force_bounds(v, 3.0);
  force_bounds(w, 3.0);
  F t = fe_reduce_small(fe_norm(fe_add(v, fe_mul(w, z))));
  report[r++] = t.vb;
  ok &= t.vb <= 3.0;
  // 3. chunk entry of kate_division: product + stored inclusive value, then the replay step and its canonical store
  F T = fe_add(fe_mul(v, z), w);
  F cur = fe_reduce_small(fe_norm(T));
  cur = fe_reduce_small(fe_norm(fe_add(fe_mul(cur, z), raw)));
  report[r++] = cur.vb;
  (void)fe_canonical(cur);
  // 4. products: factor = raw * K32, running product, scan product, conversion out
  F f = fe_mul(raw, k32);
  F p = fe_mul(z, f);
  p = fe_mul(p, p);
  report[r++] = p.vb;
  ok &= p.vb <= 2.0;
  (void)fe_canonical(p);
  (void)fe_canonical(fe_mul(p, fe_const<FrParams>(FrParams::INT2EXT)));
  F start = raw;                                    // the start value enters as raw words
  (void)fe_mul(p, start);
  // 5. linear combination: four product terms on a value < 3r between reductions
  F lc = v;
  for (int k = 0; k < 4; ++k) lc = fe_add(lc, fe_mul(raw, z));
  lc = fe_reduce_small(fe_norm(lc));
  report[r++] = lc.vb;
  ok &= lc.vb <= 3.0;
  // 6. NTT butterflies: stage 0 on raw inputs, a later trivial-twiddle stage on lazy outputs, a general stage
  F x0 = fe_add(raw, raw), x1 = fe_sub<6, 29>(raw, raw);
  F tt = fe_norm(x1);
  F y0 = fe_add(x0, tt), y1 = fe_sub<13, 29>(x0, tt);
  F g = fe_mul(fe_norm(y1), z);
  F b0 = fe_add(fe_norm(y0), g), b1 = fe_sub<3, 29>(fe_norm(y0), g);
  F o0 = fe_norm(b0), o1 = fe_norm(b1);
  report[r++] = o0.vb;
  report[r++] = o1.vb;
  (void)fe_canonical(fe_reduce_small(o0));
  (void)fe_canonical(fe_mul(o1, z));
  return ok;
}

}  // extern "C"
