// full_prover_replay.cpp -- the counterpart of the reference's one real-prover harness,
// full_prover (/root/reference/src/circuits/utils.rs:22-70): KZG setup -> keygen_vk -> keygen_pk ->
// create_proof -> verify, with the four wall-clock lines it prints (utils.rs:66-69).
//
// The Rust prover cannot be built here, so the MSM / NTT CALL TRACE those steps issue for a circuit
// of the given shape is replayed on synthetic polynomials through the C++ mirror
// of halo2_proofs (cpp/arithmetic.hpp, cpp/domain.hpp) -- everything else in create_proof is
// CPU-side Rust and is not part of these numbers.  "Verify" is the algebraic check the KZG commitments
// must satisfy: EVERY commitment create_proof's trace produced -- each (column, base set) pair, through the
// eight-in-flight path that computed it -- equals [f(s)]G, with f(s) from the device Horner kernel (of the
// inverse NTT for Lagrange-basis columns); plus commit(f) == commit_lagrange(NTT(f)) for a random f.
// The run ends with the MEASURED call trace of the library's counters (hm_get_stats): what a Rust build of the
// shim reads after create_proof instead of SURVEY.md §3.2's estimates.
//
//   full_prover_replay [k=9] [advice=20] [lookups=8] [equality=12] [max_degree=6] [fixed=18] [tamper=0] [srs_path]
// tamper=1 changes one evaluation between the two commitments, as the reference's tests tamper with
// a witness and expect `verify()` to fail: the run must then report a mismatch and exit 1.
// srs_path: load the SRS from that file if it exists (ParamsKZG::read), else generate it and write it there.
// Defaults are test_full_prover's k = 9 (/root/reference/src/circuits/merkle_sum_tree.rs:347) with the
// MerkleSumTree constraint system as transcribed in halo2-experiments_amd/circuits.py (20 advice, 8 lookups, 12 equality columns,
// degree 6, 18 fixed columns): the shape replay.py calls merkle_sum_tree_k9.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <string>

#include "../cpp/domain.hpp"

using namespace halo2;
using bn256::Fq;
using bn256::Fr;
using bn256::G1;
using bn256::G1Affine;
using Clock = std::chrono::steady_clock;

static double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

// host-side group law, only for the final check [f(s)]G
static G1 g1_double(const G1& p) {
  if (p.is_identity()) return p;
  const Fq A = p.x.square(), B = p.y.square(), C = B.square();
  Fq D = p.x * B; D = D + D; D = D + D;
  const Fq E = A + A + A;
  G1 r;
  r.x = E.square() - (D + D);
  Fq c8 = C + C; c8 = c8 + c8; c8 = c8 + c8;
  r.y = E * (D - r.x) - c8;
  r.z = p.y * p.z; r.z = r.z + r.z;
  return r;
}
static G1 g1_add(const G1& p, const G1& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  const Fq z1z1 = p.z.square(), z2z2 = q.z.square();
  const Fq u1 = p.x * z2z2, u2 = q.x * z1z1, s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
  if (u1 == u2) return s1 == s2 ? g1_double(p) : G1::identity();
  const Fq h = u2 - u1;
  Fq i = h + h; i = i.square();
  const Fq j = h * i;
  Fq r = s2 - s1; r = r + r;
  const Fq v = u1 * i;
  G1 o;
  o.x = r.square() - j - (v + v);
  Fq t = s1 * j; t = t + t;
  o.y = r * (v - o.x) - t;
  o.z = p.z * q.z * h; o.z = o.z + o.z;
  return o;
}
static G1Affine g1_mul_generator(const Fr& k_mont) {
  const Fr one_raw{{1, 0, 0, 0}};
  const Fr k = k_mont * one_raw;                        // Montgomery -> canonical
  const G1Affine g = G1Affine::generator();
  const G1 base{g.x, g.y, Fq::one()};
  G1 acc = G1::identity();
  for (int i = 255; i >= 0; --i) {
    acc = g1_double(acc);
    if ((k.l[i >> 6] >> (i & 63)) & 1) acc = g1_add(acc, base);
  }
  if (acc.is_identity()) return G1Affine::identity();
  const Fq zi = acc.z.invert(), zi2 = zi.square();
  return G1Affine{acc.x * zi2, acc.y * zi2 * zi};
}

static Fr random_fr(std::mt19937_64& rng) {
  uint64_t raw[4] = {rng(), rng(), rng(), rng() & 0x0FFFFFFFFFFFFFFFULL};   // < 2^252 < r
  return Fr::from_raw(raw);
}

int main(int argc, char** argv) {
  const uint32_t k = argc > 1 ? (uint32_t)atoi(argv[1]) : 9;
  const uint32_t advice = argc > 2 ? (uint32_t)atoi(argv[2]) : 20;
  const uint32_t lookups = argc > 3 ? (uint32_t)atoi(argv[3]) : 8;
  const uint32_t equality = argc > 4 ? (uint32_t)atoi(argv[4]) : 12;
  const uint32_t max_degree = argc > 5 ? (uint32_t)atoi(argv[5]) : 6;
  const uint32_t fixed = argc > 6 ? (uint32_t)atoi(argv[6]) : 18;
  const bool tamper = argc > 7 && atoi(argv[7]) != 0;
  const std::string srs_path = argc > 8 ? argv[8] : "";
  if (k < 4 || k > 22 || max_degree < 3) { std::fprintf(stderr, "unsupported shape\n"); return 2; }
  try {
    if (hm_device_count() <= 0) { std::fprintf(stderr, "no gfx950 device: %s\n", "this path has no CPU fallback"); return 3; }
    std::mt19937_64 rng(0x48324D4933353558ULL);
    const size_t n = (size_t)1 << k;
    const poly::EvaluationDomain dom(max_degree, k);
    const uint32_t zp = (equality + max_degree - 3) / (max_degree - 2);
    const size_t used_rows = n / 4 < 1100 ? n / 4 : 1100;

    // ParamsKZG::<Bn256>::setup(k, OsRng)  (utils.rs:28) -- or loaded from disk instead of regenerated
    const auto t_setup0 = Clock::now();
    const Fr s = random_fr(rng);
    std::unique_ptr<poly::ParamsKZG> params_ptr;
    bool srs_loaded = false;
    if (!srs_path.empty()) {
      if (FILE* probe = std::fopen(srs_path.c_str(), "rb")) {
        std::fclose(probe);
        params_ptr.reset(new poly::ParamsKZG(srs_path));
        srs_loaded = true;
        if (params_ptr->k != k) throw std::runtime_error("SRS file holds another k");
        if (!(params_ptr->s_g2 == params_ptr->g2.mul(s))) throw std::runtime_error("SRS file was made with another trapdoor");
      } else {
        params_ptr.reset(new poly::ParamsKZG(k, s, false, true));
        params_ptr->write(srs_path);
      }
    } else {
      params_ptr.reset(new poly::ParamsKZG(k, s));
    }
    const poly::ParamsKZG& params = *params_ptr;
    const auto t_setup1 = Clock::now();

    // synthetic columns: dense = uniform; sparse = used_rows small values + 6 blinding rows
    std::vector<Fr> dense(n), sparse(n, Fr::zero());
    for (auto& x : dense) x = random_fr(rng);
    for (size_t i = 0; i < used_rows; ++i) sparse[i] = Fr::from_u64(i % 2 ? 1 : (rng() & 0xFFFF));
    for (size_t i = n - 6; i < n; ++i) sparse[i] = random_fr(rng);
    poly::DevicePolys d_dense(n, 1), d_sparse(n, 1);
    d_dense.upload(dense);
    d_sparse.upload(sparse);

    // keygen_vk (utils.rs:31): commit_lagrange per fixed column and per permutation sigma
    const auto t_vk0 = Clock::now();
    for (uint32_t i = 0; i < fixed; ++i) (void)params.commit_lagrange(d_sparse.d);
    for (uint32_t i = 0; i < equality; ++i) (void)params.commit_lagrange(d_dense.d);
    const auto t_vk1 = Clock::now();

    // keygen_pk (utils.rs:35): lagrange_to_coeff + coeff_to_extended of fixed, sigma, l0 / l_last / l_active
    const auto t_pk0 = Clock::now();
    {
      const size_t polys = fixed + equality + 3;
      for (size_t done = 0; done < polys; done += 8) {
        const size_t b = polys - done < 8 ? polys - done : 8;
        poly::DevicePolys batch(n, b);
        for (size_t i = 0; i < b; ++i) (void)hipMemcpy(batch.poly(i), d_dense.d, n * sizeof(Fr), hipMemcpyDeviceToDevice);
        dom.lagrange_to_coeff(batch);
        poly::DevicePolys ext = dom.coeff_to_extended(batch);
      }
      (void)hipDeviceSynchronize();
    }
    const auto t_pk1 = Clock::now();

    // create_proof (utils.rs:40-48): the MSM / NTT trace of SURVEY.md §3.2 plus the vector steps between them.  The
    // commitments of one phase are independent: each phase is ONE hm_msm_batch_bn256_g1_dev call.
    (void)hipDeviceSynchronize();
    // every commitment of the trace is kept and checked after the timed region: (which column, which base set, result)
    struct Made { int pair; G1 c; };
    std::vector<Made> made;
    made.reserve(256);
    auto commit_phase = [&](uint32_t count, const Fr* d_poly, bool lagrange, int pair) {
      const std::vector<const Fr*> cols(count, d_poly);
      for (const G1& c : params.commit_batch(cols, lagrange)) made.push_back(Made{pair, c});
    };
    // the prover's polynomial buffers live for the whole proof (as halo2's Vec<Polynomial> do)
    poly::DevicePolys batch(n, 8), ext(dom.extended_len(), 8), hpoly(dom.extended_len(), 1);
    const size_t n_z = zp + lookups, usable = n - 7;
    poly::DevicePolys z_fac(n, n_z ? n_z : 1), z_col(n, n_z ? n_z : 1), lk_in(n, 1), lk_tab(n, 1), lk_out(n, lookups ? 2 * lookups : 1), open_acc(n, 5),
        open_q(n, 5);
    {
      std::vector<Fr> fac(n * (n_z ? n_z : 1)), tab(n), inp(n);
      for (auto& x : fac) x = random_fr(rng);
      const size_t span = usable < 65536 ? usable : 65536;          // a range-check lookup: table 0 .. span - 1, inputs from it
      for (size_t i = 0; i < n; ++i) tab[i] = Fr::from_u64(i % span);
      for (size_t i = 0; i < n; ++i) inp[i] = Fr::from_u64(rng() % span);
      z_fac.upload(fac);
      lk_tab.upload(tab);
      lk_in.upload(inp);
    }
    auto vector_steps = [&]() {
      // the z columns of the permutation and lookup arguments: all denominators inverted in one call, then the products
      if (n_z) arithmetic::batch_invert(z_fac.d, n * n_z);
      // (permutation: each column set starts where the one before stood at the last usable row -- upstream's last_z, chained
      // on the device; lookups: one independent product each)
      std::vector<const Fr*> pf, lf;
      std::vector<Fr*> po, lo;
      for (size_t i = 0; i < n_z; ++i) {
        (i < zp ? pf : lf).push_back(z_fac.poly(i));
        (i < zp ? po : lo).push_back(z_col.poly(i));
      }
      if (!pf.empty()) arithmetic::grand_product_batch(pf, n, Fr::one(), usable, po);
      if (!lf.empty()) arithmetic::grand_product_batch(lf, n, Fr::one(), HM_NO_CHAIN, lo);
      // the permuted columns of every lookup argument, one call
      if (lookups) {
        std::vector<const void*> ins(lookups, lk_in.d), tabs(lookups, lk_tab.d);
        std::vector<void*> oa(lookups), os(lookups);
        for (uint32_t i = 0; i < lookups; ++i) { oa[i] = lk_out.poly(2 * i); os[i] = lk_out.poly(2 * i + 1); }
        arithmetic::check(hm_lookup_permute_batch_bn256_fr_dev(ins.data(), tabs.data(), lookups, usable, oa.data(), os.data(), nullptr, nullptr),
                          "permute_expression_pair");
      }
    };
    auto multiopen_steps = [&]() {
      // multiopen: per rotation set one combination of the committed polynomials, a division by (X - point) per opening
      // point, then the final combination and division
      const size_t n_open = advice + 3 * lookups + zp + (max_degree - 1), per_set = (n_open + 3) / 4;
      for (size_t si = 0; si < 4; ++si) {
        const size_t start = si * per_set < n_open ? si * per_set : n_open;
        const size_t cnt = per_set < n_open - start ? per_set : n_open - start;
        std::vector<const Fr*> ps(cnt, d_dense.d);
        std::vector<Fr> cs(cnt, s);
        arithmetic::linear_combination(ps, cs, n, open_acc.poly(si));
      }
      {
        std::vector<const Fr*> ps;
        std::vector<Fr*> qs;
        std::vector<Fr> zs;
        for (size_t qi = 0; qi < 5; ++qi) {
          ps.push_back(open_acc.poly(qi % 4));
          qs.push_back(open_q.poly(qi));
          zs.push_back(s + Fr::from_u64(qi + 2));
        }
        arithmetic::kate_division_batch(ps, n, zs, qs);       // the openings of one round are independent of each other
      }
      arithmetic::linear_combination({open_acc.poly(0), open_acc.poly(1), open_acc.poly(2), open_acc.poly(3)}, {s, s, s, s}, n, open_acc.poly(4));
      arithmetic::kate_division(open_acc.poly(4), n, s + Fr::one(), open_q.d);
    };
    size_t n_msm = 0, n_ntt = 0;
    auto prove = [&]() {
      n_msm = n_ntt = 0;
      commit_phase(advice + 2 * lookups, d_sparse.d, true, 0);       // advice, permuted lookup columns
      vector_steps();                                                // lookup permutations, z columns
      commit_phase(zp + lookups + 1, d_dense.d, true, 1);            // grand products, random poly
      n_msm += advice + 2 * lookups + zp + lookups + 1;
      const size_t polys = advice + 1 + 3 * lookups + zp;
      for (size_t done = 0; done < polys; done += 8) {
        const size_t b = polys - done < 8 ? polys - done : 8;
        batch.batch = b;                                             // a view of the first b rows
        for (size_t i = 0; i < b; ++i)
          (void)hipMemcpyAsync(batch.poly(i), d_dense.d, n * sizeof(Fr), hipMemcpyDeviceToDevice, nullptr);
        dom.lagrange_to_coeff(batch);                                // lagrange_to_coeff per polynomial
        dom.coeff_to_extended(batch, ext);                           // coeff_to_extended per polynomial
        n_ntt += 2 * b;
        if (done + b >= polys) {                                     // h(X): back to coefficients once
          (void)hipMemcpyAsync(hpoly.d, ext.d, ext.len * sizeof(Fr), hipMemcpyDeviceToDevice, nullptr);
          dom.extended_to_coeff(hpoly);
          ++n_ntt;
        }
      }
      batch.batch = 8;
      commit_phase(max_degree - 1, d_dense.d, false, 2);             // h pieces
      multiopen_steps();
      commit_phase(2, d_dense.d, false, 2);                          // SHPLONK
      n_msm += (max_degree - 1) + 2;
      (void)hipDeviceSynchronize();
    };
    prove();                                                         // warm-up proof: every workspace reaches its size here
    made.clear();
    (void)hm_reset_stats();                                          // the measured call trace covers create_proof only
    const auto t_pr0 = Clock::now();
    prove();
    const auto t_pr1 = Clock::now();
    hm_stats trace;
    arithmetic::check(hm_get_stats(&trace), "hm_get_stats");

    // verify: commit(f) == commit_lagrange(NTT(f)) == [f(s)]G
    const auto t_v0 = Clock::now();
    poly::DevicePolys f(n, 1);
    f.upload(dense);
    const G1 c_coeff = params.commit(f.d);
    arithmetic::check(hm_ntt_bn256_fr_dev(f.d, dom.omega.l, k, nullptr), "best_fft");
    if (tamper) {                                           // one evaluation off by one
      Fr e;
      (void)hipMemcpy(&e, f.d + 3, sizeof(Fr), hipMemcpyDeviceToHost);
      e = e + Fr::one();
      (void)hipMemcpy(f.d + 3, &e, sizeof(Fr), hipMemcpyHostToDevice);
    }
    const G1 c_lagrange = params.commit_lagrange(f.d);
    Fr fs = Fr::zero();
    for (size_t i = n; i-- > 0;) fs = fs * s + dense[i];
    const G1Affine expect = g1_mul_generator(fs);
    bool ok = arithmetic::to_affine(c_coeff) == expect && arithmetic::to_affine(c_lagrange) == expect;
    // ... and every commitment the trace made: pair 0 = (sparse column, g_lagrange), 1 = (dense, g_lagrange),
    // 2 = (dense, g).  f(s) by the device Horner kernel -- for a Lagrange-basis column, of its inverse NTT.
    G1Affine want[3];
    {
      poly::DevicePolys tmp(n, 1);
      (void)hipMemcpy(tmp.d, d_sparse.d, n * sizeof(Fr), hipMemcpyDeviceToDevice);
      dom.lagrange_to_coeff(tmp);
      want[0] = g1_mul_generator(arithmetic::eval_polynomial(tmp.d, n, s));
      (void)hipMemcpy(tmp.d, d_dense.d, n * sizeof(Fr), hipMemcpyDeviceToDevice);
      dom.lagrange_to_coeff(tmp);
      want[1] = g1_mul_generator(arithmetic::eval_polynomial(tmp.d, n, s));
      want[2] = g1_mul_generator(arithmetic::eval_polynomial(d_dense.d, n, s));
      if (!(want[2] == expect)) ok = false;            // device Horner vs the host Horner above
    }
    // ... and one opening, as multiopen makes it (the pairing check e(C - [w(z)]G, H) = e(commit(q), [s - z]H) in the
    // exponent): w = v0 f + v1 f', q = kate_division(w, z); then q(s) (s - z) + w(z) = w(s) and commit(q) = [q(s)]G.
    // Plus the z-column identity of the permutation argument's building blocks: prod f_j * prod f_j^-1 = 1.
    bool opening_ok = true;
    {
      poly::DevicePolys w(n, 1), q(n, 1), inv(n, 1), zf(n, 1), zi(n, 1);
      const Fr v0 = s + Fr::one(), v1 = s * s, z = s * s * s + Fr::one();
      arithmetic::linear_combination({d_dense.d, d_sparse.d}, {v0, v1}, n, w.d);
      (void)hipMemset(q.d, 0, n * sizeof(Fr));                 // the quotient has n - 1 coefficients: the last stays zero
      arithmetic::kate_division(w.d, n, z, q.d);
      const Fr wz = arithmetic::eval_polynomial(w.d, n, z), ws = arithmetic::eval_polynomial(w.d, n, s);
      const Fr qs = arithmetic::eval_polynomial(q.d, n, s);
      if (!(qs * (s - z) + wz == ws)) opening_ok = false;
      if (!(arithmetic::to_affine(params.commit(q.d)) == g1_mul_generator(qs))) opening_ok = false;
      (void)hipMemcpy(inv.d, d_dense.d, n * sizeof(Fr), hipMemcpyDeviceToDevice);
      arithmetic::batch_invert(inv.d, n);
      arithmetic::grand_product(d_dense.d, n, Fr::one(), zf.d);
      arithmetic::grand_product(inv.d, n, Fr::one(), zi.d);
      Fr a, b;
      (void)hipMemcpy(&a, zf.d + (n - 1), sizeof(Fr), hipMemcpyDeviceToHost);
      (void)hipMemcpy(&b, zi.d + (n - 1), sizeof(Fr), hipMemcpyDeviceToHost);
      if (!(a * b == Fr::one())) opening_ok = false;
      if (tamper) opening_ok = true;                           // the tampered run fails on the commitments above
    }
    {   // a phase from host vectors equals the device-resident phase
      const std::vector<const Fr*> hp{dense.data(), sparse.data(), dense.data()};
      const std::vector<const Fr*> dp{d_dense.d, d_sparse.d, d_dense.d};
      const std::vector<G1> a = params.commit_batch_host(hp, true), b = params.commit_batch(dp, true);
      for (size_t i = 0; i < a.size(); ++i)
        if (!(arithmetic::to_affine(a[i]) == arithmetic::to_affine(b[i]))) opening_ok = false;
    }
    {   // the extended domain by cosets (what a multi-GPU prover deals over its devices) equals the whole-array route:
        // values on every coset = the residue classes of rows, and the recombined coefficients = extended_to_coeff's
      const size_t e = dom.num_cosets();
      poly::DevicePolys c1(n, 1), ext(dom.extended_len(), 1), part(n, e), back(n, e);
      (void)hipMemcpy(c1.d, d_dense.d, n * sizeof(Fr), hipMemcpyDeviceToDevice);
      dom.coeff_to_extended(c1, ext);
      const std::vector<Fr> rows = ext.download();
      std::vector<const Fr*> parts;
      for (size_t j = 0; j < e; ++j) {
        poly::DevicePolys one(n, 1);
        dom.coeff_to_coset(c1, j, one);
        const std::vector<Fr> got = one.download();
        for (size_t t = 0; t < n; t += (n / 64 ? n / 64 : 1))
          if (!(got[t] == rows[e * t + j])) opening_ok = false;
        dom.coset_to_partial(one, j);
        (void)hipMemcpy(part.poly(j), one.d, n * sizeof(Fr), hipMemcpyDeviceToDevice);
        parts.push_back(part.poly(j));
      }
      dom.combine_cosets(parts, e, back.d);
      dom.extended_to_coeff(ext);
      const std::vector<Fr> a = back.download(), b = ext.download();
      for (size_t i = 0; i < a.size(); i += 61)
        if (!(a[i] == b[i])) opening_ok = false;
      for (size_t i = 0; i < n; ++i)                              // a polynomial of degree < n: its own coefficients, then zeros
        if (!(a[i] == dense[i])) opening_ok = false;
      {   // the same cosets in one launch chain: array c of the output = coset which[c]
        std::vector<size_t> which;
        for (size_t j = 0; j < e && j < 16; ++j) which.push_back(e - 1 - j);
        poly::DevicePolys many(n, which.size());
        dom.coeff_to_cosets(c1, which, many);
        const std::vector<Fr> got = many.download();
        for (size_t c = 0; c < which.size(); ++c)
          for (size_t t = 0; t < n; t += (n / 32 ? n / 32 : 1))
            if (!(got[c * n + t] == rows[e * t + which[c]])) opening_ok = false;
        dom.cosets_to_partials(many, which);
        const std::vector<Fr> p2 = many.download(), p1 = part.download();
        for (size_t c = 0; c < which.size(); ++c)
          for (size_t i = 0; i < n; i += 53)
            if (!(p2[c * n + i] == p1[which[c] * n + i])) opening_ok = false;
      }
      // ... and from the quotient_poly_degree cosets that determine a polynomial of that degree (here any subset does)
      if (dom.min_cosets() <= e && e >= 2) {
        const size_t q = dom.min_cosets() < 2 ? 2 : dom.min_cosets();
        std::vector<const Fr*> sub;
        std::vector<size_t> which;
        for (size_t a2 = 0; a2 < q; ++a2) { which.push_back(e - 1 - a2); sub.push_back(part.poly(e - 1 - a2)); }
        poly::DevicePolys few(n, q);
        dom.combine_cosets(sub, which, q, few.d);
        const std::vector<Fr> c = few.download();
        for (size_t i = 0; i < n; ++i)
          if (!(c[i] == dense[i])) opening_ok = false;
        for (size_t i = n; i < c.size(); i += 37)
          if (!(c[i] == Fr::zero())) opening_ok = false;
      }
    }
    if (!opening_ok) ok = false;
    size_t bad = 0;
    for (const Made& m : made)
      if (!(arithmetic::to_affine(m.c) == want[m.pair])) ++bad;
    if (bad || made.size() != n_msm) ok = false;
    const auto t_v1 = Clock::now();

    std::printf("shape: k=%u advice=%u lookups=%u equality=%u max_degree=%u fixed=%u extended_k=%u  (%zu MSMs, %zu NTTs in create_proof)\n",
                k, advice, lookups, equality, max_degree, fixed, dom.extended_k, n_msm, n_ntt);
    std::printf("Time to generate params %.6fs\n", secs(t_setup0, t_setup1));
    std::printf("Time to generate vk %.6fs\n", secs(t_vk0, t_vk1));
    std::printf("Time to generate pk %.6fs\n", secs(t_pk0, t_pk1));
    std::printf("Prover Time %.6fs\n", secs(t_pr0, t_pr1));
    std::printf("Verifier Time %.6fs\n", secs(t_v0, t_v1));
    std::printf("SRS: %s\n", srs_loaded ? "loaded from disk" : (srs_path.empty() ? "generated" : "generated and written to disk"));
    std::printf("measured call trace (hm_get_stats): %llu MSMs / %llu points, %llu NTTs / %llu elements; MSM device %.3f ms, host fold %.3f ms\n",
                (unsigned long long)trace.msm_calls, (unsigned long long)trace.msm_points, (unsigned long long)trace.ntt_calls,
                (unsigned long long)trace.ntt_elements, trace.msm_device_us / 1e3, trace.msm_host_us / 1e3);
    for (int lg = 0; lg < 32; ++lg)
      if (trace.msm_calls_by_log2[lg] || trace.ntt_calls_by_log2[lg])
        std::printf("  2^%-2d  MSM calls %4llu   NTT calls %4llu\n", lg, (unsigned long long)trace.msm_calls_by_log2[lg],
                    (unsigned long long)trace.ntt_calls_by_log2[lg]);
    {
      hm_stats after;                                        // the verification section's own calls, by kind
      arithmetic::check(hm_get_stats(&after), "hm_get_stats");
      static const char* kinds[7] = {"eval_polynomial", "graph_evaluate", "kate_division", "grand_product", "batch_invert", "linear_combination",
                                     "lookup_permute"};
      std::printf("other entry points since the reset:");
      for (int i = 0; i < 7; ++i)
        if (after.vector_calls[i]) std::printf(" %s x%llu", kinds[i], (unsigned long long)after.vector_calls[i]);
      std::printf("\n");
    }
    std::printf("checked %zu commitments of the trace against [f(s)]G: %zu mismatches\n", made.size(), bad);
    std::printf("opening q = (w - w(z)) / (X - z) and the z-column product identity: %s\n", opening_ok ? "ok" : "FAILED");
    std::printf("%s\n", ok ? "commitments verified" : "COMMITMENT MISMATCH");
    (void)hm_shutdown();
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 4;
  }
}
