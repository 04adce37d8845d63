// domain.hpp -- C++ mirror of halo2_proofs::poly::EvaluationDomain<bn256::Fr> and of the two
// ParamsKZG commit functions, the callers either side of best_fft / best_multiexp in create_proof
// (upstream poly/domain.rs and poly/kzg/commitment.rs at the pinned tag).
// Polynomials live in device buffers (hipMalloc) between calls, the "next row" mode; the scale /
// coset passes are fused into the transforms and same-size transforms can be batched.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "arithmetic.hpp"

namespace halo2 {
namespace poly {

using bn256::Fr;
using bn256::G1;
using bn256::G1Affine;

// `batch` polynomials of `len` Fr each, resident in HBM
struct DevicePolys {
  Fr* d = nullptr;
  size_t len = 0, batch = 0;
  DevicePolys() = default;
  DevicePolys(size_t len_, size_t batch_) : len(len_), batch(batch_) {
    if (hipMalloc((void**)&d, len * batch * sizeof(Fr)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
  }
  DevicePolys(const DevicePolys&) = delete;
  DevicePolys& operator=(const DevicePolys&) = delete;
  DevicePolys(DevicePolys&& o) noexcept : d(o.d), len(o.len), batch(o.batch) { o.d = nullptr; }
  DevicePolys& operator=(DevicePolys&& o) noexcept {
    if (this != &o) { release(); d = o.d; len = o.len; batch = o.batch; o.d = nullptr; }
    return *this;
  }
  ~DevicePolys() { release(); }
  void release() { if (d) (void)hipFree(d); d = nullptr; }
  void upload(const std::vector<Fr>& h) {
    if (h.size() != len * batch) throw std::invalid_argument("DevicePolys::upload: size mismatch");
    if (hipMemcpy(d, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("H2D failed");
  }
  std::vector<Fr> download() const {
    std::vector<Fr> h(len * batch);
    if (hipMemcpy(h.data(), d, h.size() * sizeof(Fr), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("D2H failed");
    return h;
  }
  Fr* poly(size_t i) const { return d + i * len; }
};

class EvaluationDomain {
 public:
  uint32_t k, extended_k;
  size_t n, quotient_poly_degree;
  Fr omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;

  // EvaluationDomain::new(j, k): j = maximum constraint degree
  EvaluationDomain(uint32_t j, uint32_t k_) : k(k_) {
    if (j < 2) throw std::invalid_argument("EvaluationDomain: j must be >= 2");
    n = (size_t)1 << k;
    quotient_poly_degree = j - 1;
    extended_k = k;
    while (((size_t)1 << extended_k) < n * quotient_poly_degree) ++extended_k;
    if (extended_k > bn256::FR_S) throw std::invalid_argument("EvaluationDomain: extended_k exceeds the two-adicity");
    extended_omega = bn256::fr_root_of_unity();
    for (uint32_t i = extended_k; i < bn256::FR_S; ++i) extended_omega = extended_omega.square();
    extended_omega_inv = extended_omega.invert();
    omega = extended_omega;
    for (uint32_t i = k; i < extended_k; ++i) omega = omega.square();
    omega_inv = omega.invert();
    g_coset = bn256::fr_zeta();
    g_coset_inv = g_coset.square();
    ifft_divisor = Fr::from_u64((uint64_t)n).invert();
    extended_ifft_divisor = Fr::from_u64((uint64_t)1 << extended_k).invert();
  }
  size_t extended_len() const { return (size_t)1 << extended_k; }

  // lagrange_to_coeff: ifft(a, omega_inv, k, 1/n), in place, the whole batch in one set of launches
  void lagrange_to_coeff(DevicePolys& a, hipStream_t stream = nullptr) const {
    if (a.len != n) throw std::invalid_argument("lagrange_to_coeff: a.len() != n");
    arithmetic::check(hm_ntt_batch_bn256_fr_dev(a.d, a.batch, omega_inv.l, k, ifft_divisor.l, nullptr, stream), "lagrange_to_coeff");
  }
  // coeff_to_extended: zero-pad, distribute_powers_zeta, best_fft(extended_omega) -- one call; the padded
  // array is never materialised (the first pass reads the n coefficients only)
  DevicePolys coeff_to_extended(const DevicePolys& a, hipStream_t stream = nullptr) const {
    DevicePolys ext(extended_len(), a.batch);
    coeff_to_extended(a, ext, stream);
    return ext;
  }
  // ... into a caller-owned buffer (a prover keeps its extended-domain buffers for the whole proof:
  // hipMalloc / hipFree per polynomial cost more than the transform); the first a.batch rows of ext are written
  void coeff_to_extended(const DevicePolys& a, DevicePolys& ext, hipStream_t stream = nullptr) const {
    if (a.len != n) throw std::invalid_argument("coeff_to_extended: a.len() != n");
    if (ext.len != extended_len() || ext.batch < a.batch) throw std::invalid_argument("coeff_to_extended: ext too small");
    const Fr coset[3] = {Fr::one(), g_coset, g_coset.square()};
    arithmetic::check(hm_coeff_to_extended_bn256_fr_dev(a.d, ext.d, a.batch, extended_omega.l, k, extended_k, coset[0].l, stream),
                      "coeff_to_extended");
  }
  // ... on HOST vectors, upstream's own signature (Polynomial<Coeff> -> Polynomial<ExtendedLagrangeCoeff>): what the patched
  // EvaluationDomain::coeff_to_extended of the drop-in prover calls.  n coefficients cross PCIe upwards, never the zero padding.
  std::vector<Fr> coeff_to_extended(const std::vector<Fr>& a) const {
    if (a.size() != n) throw std::invalid_argument("coeff_to_extended: a.len() != n");
    std::vector<Fr> ext(extended_len());
    const Fr coset[3] = {Fr::one(), g_coset, g_coset.square()};
    arithmetic::check(hm_coeff_to_extended_bn256_fr(reinterpret_cast<const uint64_t*>(a.data()), reinterpret_cast<uint64_t*>(ext.data()),
                                                    extended_omega.l, k, extended_k, coset[0].l),
                      "coeff_to_extended");
    return ext;
  }
  // extended_to_coeff on a HOST vector: consumes the evaluations, returns the n * (j - 1) coefficients upstream keeps (only those
  // come back over PCIe)
  std::vector<Fr> extended_to_coeff(std::vector<Fr> a) const {
    if (a.size() != extended_len()) throw std::invalid_argument("extended_to_coeff: a.len() != extended_len()");
    const Fr c3[3] = {Fr::one(), g_coset_inv, g_coset_inv.square()};
    const size_t keep = n * quotient_poly_degree;
    arithmetic::check(hm_extended_to_coeff_bn256_fr(reinterpret_cast<uint64_t*>(a.data()), keep, extended_omega_inv.l, extended_k,
                                                    extended_ifft_divisor.l, c3[0].l),
                      "extended_to_coeff");
    a.resize(keep);
    return a;
  }
  // divide_by_vanishing_poly: a[i] *= t_evaluations[i % 2^(extended_k - k)] on the extended coset, in place
  void divide_by_vanishing_poly(DevicePolys& a, hipStream_t stream = nullptr) const {
    if (a.len != extended_len()) throw std::invalid_argument("divide_by_vanishing_poly: a.len() != extended_len()");
    const uint32_t period = 1u << (extended_k - k);
    if (period > 64) throw std::invalid_argument("divide_by_vanishing_poly: extension factor above 64");
    std::vector<Fr> t_inv(period);
    Fr x = g_coset;                                          // zeta * extended_omega^i
    for (uint32_t i = 0; i < period; ++i) {
      Fr xn = x;
      for (uint32_t b = 0; b < k; ++b) xn = xn.square();     // x^n, n = 2^k
      t_inv[i] = (xn - Fr::one()).invert();
      x = x * extended_omega;
    }
    arithmetic::check(hm_fr_mul_periodic_dev(a.d, a.len * a.batch, reinterpret_cast<const uint64_t*>(t_inv.data()), period, stream),
                      "divide_by_vanishing_poly");
  }
  // -- the extended domain one coset of <omega> at a time (hm_coeff_to_coset_bn256_fr_dev; later halo2_proofs:
  // coeff_to_extended_part): row E t + j of coeff_to_extended's array is point t of coset j, E = num_cosets()
  size_t num_cosets() const { return (size_t)1 << (extended_k - k); }
  Fr coset_shift(size_t j) const { return g_coset * extended_omega.pow_u64((uint64_t)j); }
  // 1 / (X^n - 1) on coset j: one constant
  Fr coset_vanishing_inverse(size_t j) const {
    Fr xn = coset_shift(j);
    for (uint32_t b = 0; b < k; ++b) xn = xn.square();
    return (xn - Fr::one()).invert();
  }
  // a.batch coefficient arrays -> their values on coset j (out may be a itself); internal: multiplied by 32 (the
  // evaluator's HM_GRAPH_COLUMNS_INTERNAL form)
  void coeff_to_coset(const DevicePolys& a, size_t j, DevicePolys& out, bool internal = false, hipStream_t stream = nullptr) const {
    if (a.len != n || out.len != n || out.batch < a.batch) throw std::invalid_argument("coeff_to_coset: shapes");
    if (j >= num_cosets()) throw std::invalid_argument("coeff_to_coset: no such coset");
    arithmetic::check(hm_coeff_to_coset_bn256_fr_dev(a.d, out.d, a.batch, omega.l, k, coset_shift(j).l, internal ? 1 : 0, stream), "coeff_to_coset");
  }
  // a.batch coefficient arrays onto several cosets in ONE launch chain: out holds a.batch * cosets.size() arrays of n, array
  // b * cosets.size() + c = polynomial b on coset cosets[c] (per input the cosets lie one after the other: the column layout
  // hm_graph_evaluate_segments_dev reads, segment c = coset c).  At most 16 cosets per call.
  void coeff_to_cosets(const DevicePolys& a, const std::vector<size_t>& cosets, DevicePolys& out, bool internal = false,
                       hipStream_t stream = nullptr) const {
    if (a.len != n || out.len != n || out.batch < a.batch * cosets.size()) throw std::invalid_argument("coeff_to_cosets: shapes");
    std::vector<Fr> shifts;
    for (size_t c : cosets) {
      if (c >= num_cosets()) throw std::invalid_argument("coeff_to_cosets: no such coset");
      shifts.push_back(coset_shift(c));
    }
    arithmetic::check(hm_coeff_to_cosets_bn256_fr_dev(a.d, out.d, a.batch, omega.l, k, reinterpret_cast<const uint64_t*>(shifts.data()),
                                                      shifts.size(), internal ? 1 : 0, stream),
                      "coeff_to_cosets");
  }
  // the values ONE polynomial takes on cosets.size() cosets (v.batch arrays of n, in that order) -> its partials, in place
  void cosets_to_partials(DevicePolys& v, const std::vector<size_t>& cosets, hipStream_t stream = nullptr) const {
    if (v.len != n || v.batch != cosets.size()) throw std::invalid_argument("cosets_to_partials: one array of n per coset");
    std::vector<Fr> inv;
    for (size_t c : cosets) inv.push_back(coset_shift(c).invert());
    arithmetic::check(hm_cosets_to_coeff_bn256_fr_dev(v.d, v.batch, omega_inv.l, k, ifft_divisor.l, reinterpret_cast<const uint64_t*>(inv.data()),
                                                      stream),
                      "cosets_to_partials");
  }
  // values of h on coset j -> d_j[i] = sum_q h[i + q n] zeta^(n q) w^(j q), in place
  void coset_to_partial(DevicePolys& v, size_t j, hipStream_t stream = nullptr) const {
    if (v.len != n) throw std::invalid_argument("coset_to_partial: v.len() != n");
    arithmetic::check(hm_coset_to_coeff_bn256_fr_dev(v.d, v.batch, omega_inv.l, k, ifft_divisor.l, coset_shift(j).invert().l, stream),
                      "coset_to_partial");
  }
  // The quotient has fewer than n * quotient_poly_degree coefficients: that many cosets determine it.
  size_t min_cosets() const { return quotient_poly_degree; }
  // partials[a] = coset_to_partial(values on coset cosets[a]) (device pointers to n elements each) -> `pieces` x n
  // coefficients at out: the polynomial of degree < q n through the values on those q cosets.  With all E cosets in order it
  // is what extended_to_coeff returns; with q = quotient_poly_degree cosets it is the same quotient for a satisfied circuit,
  // and the other E - q cosets of evaluate_h need not be computed.  partial_j[i] = sum_t piece_t[i] u_j^t, u_j =
  // coset_shift(j)^n: piece_t = sum_a Vinv[t][a] partial_a, V[a][t] = u_(j_a)^t (Gauss-Jordan on the host, q <= E).
  // divide_by_vanishing: the partials are of the UNDIVIDED numerator; 1 / (X^n - 1) = 1 / (u_c - 1) on coset c rides on the matrix
  void combine_cosets(const std::vector<const Fr*>& partials, const std::vector<size_t>& cosets, size_t pieces, Fr* out,
                      hipStream_t stream = nullptr, bool divide_by_vanishing = false) const {
    const size_t q = cosets.size();
    if (partials.size() != q || q == 0 || pieces == 0 || pieces > q) throw std::invalid_argument("combine_cosets: one partial per coset, at most one piece per coset");
    for (size_t a = 0; a < q; ++a) {
      if (cosets[a] >= num_cosets()) throw std::invalid_argument("combine_cosets: no such coset");
      for (size_t b = 0; b < a; ++b)
        if (cosets[a] == cosets[b]) throw std::invalid_argument("combine_cosets: cosets must be distinct");
    }
    std::vector<std::vector<Fr>> m(q, std::vector<Fr>(2 * q, Fr::zero()));
    for (size_t a = 0; a < q; ++a) {
      Fr u = coset_shift(cosets[a]);
      for (uint32_t b = 0; b < k; ++b) u = u.square();
      Fr p = Fr::one();
      for (size_t t = 0; t < q; ++t) { m[a][t] = p; p = p * u; }
      m[a][q + a] = Fr::one();
    }
    for (size_t col = 0; col < q; ++col) {
      size_t piv = col;
      while (m[piv][col] == Fr::zero()) ++piv;                    // distinct nodes: the matrix is invertible
      std::swap(m[col], m[piv]);
      const Fr inv = m[col][col].invert();
      for (Fr& v : m[col]) v = v * inv;
      for (size_t row = 0; row < q; ++row) {
        if (row == col || m[row][col] == Fr::zero()) continue;
        const Fr f = m[row][col];
        for (size_t c2 = 0; c2 < 2 * q; ++c2) m[row][c2] = m[row][c2] - f * m[col][c2];
      }
    }
    std::vector<Fr> tinv(q, Fr::one());
    if (divide_by_vanishing)
      for (size_t a = 0; a < q; ++a) tinv[a] = coset_vanishing_inverse(cosets[a]);
    for (size_t t = 0; t < pieces; ++t) {
      std::vector<Fr> c(q);
      for (size_t a = 0; a < q; ++a) c[a] = m[t][q + a] * tinv[a];
      arithmetic::linear_combination(partials, c, n, out + t * n, stream);
    }
  }
  void combine_cosets(const std::vector<const Fr*>& partials, size_t pieces, Fr* out, hipStream_t stream = nullptr) const {
    std::vector<size_t> all(num_cosets());
    for (size_t j = 0; j < all.size(); ++j) all[j] = j;
    combine_cosets(partials, all, pieces, out, stream);
  }
  // extended_to_coeff: ifft over the extended domain, undo the coset shift (caller truncates to n*(j-1))
  void extended_to_coeff(DevicePolys& a, hipStream_t stream = nullptr) const {
    if (a.len != extended_len()) throw std::invalid_argument("extended_to_coeff: a.len() != extended_len()");
    const Fr c3[3] = {Fr::one(), g_coset_inv, g_coset_inv.square()};
    arithmetic::check(hm_extended_to_coeff_bn256_fr_dev(a.d, a.batch, extended_omega_inv.l, extended_k, extended_ifft_divisor.l,
                                                        c3[0].l, stream),
                      "extended_to_coeff");
  }
};

// ParamsKZG<Bn256>: g = [s^i]G, g_lagrange = [L_i(s)]G, device resident; g2, s_g2; commit / commit_lagrange;
// write / read of the on-disk SRS (u32 k | n x 64 B g | n x 64 B g_lagrange | 128 B g2 | 128 B s_g2 -- the bytes
// the Rust types hold, the field order of upstream's ParamsKZG::write with raw points)
class ParamsKZG {
 public:
  uint32_t k;
  size_t n;
  uint64_t g_handle = 0, g_lagrange_handle = 0;
  bn256::G2Affine g2, s_g2;
  std::vector<G1Affine> g_points, g_lagrange_points;      // host copies, kept only for write()

  // setup with a caller-provided trapdoor s (the reference draws it from OsRng, utils.rs:28).  All G1 work is on
  // the device: the ladder s^i, its scaled inverse NTT (= the Lagrange scalars L_i(s)), 2 n fixed-base multiples.
  ParamsKZG(uint32_t k_, const Fr& s, bool precompute = false, bool keep_points = false) : k(k_), n((size_t)1 << k_) {
    const EvaluationDomain dom(2, k);
    DevicePolys ladder(n, 1);
    arithmetic::check(hm_fr_powers_dev(ladder.d, n, s.l, nullptr), "ParamsKZG::setup");
    g_handle = fixed_base_set(ladder, precompute, keep_points ? &g_points : nullptr);
    dom.lagrange_to_coeff(ladder);                         // n^-1 NTT_{omega^-1}(s^i) = L_i(s)
    g_lagrange_handle = fixed_base_set(ladder, precompute, keep_points ? &g_lagrange_points : nullptr);
    g2 = bn256::G2Affine::generator();
    s_g2 = g2.mul(s);
  }
  // read(): load instead of regenerate
  explicit ParamsKZG(const std::string& path, bool precompute = false) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("ParamsKZG::read: cannot open " + path);
    uint32_t kk = 0;
    bool ok = std::fread(&kk, 4, 1, f) == 1 && kk <= 28;
    k = kk;
    n = (size_t)1 << (ok ? kk : 0);
    if (ok) {
      g_points.resize(n);
      g_lagrange_points.resize(n);
      ok = std::fread(g_points.data(), 64, n, f) == n && std::fread(g_lagrange_points.data(), 64, n, f) == n &&
           std::fread(&g2, 128, 1, f) == 1 && std::fread(&s_g2, 128, 1, f) == 1;
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("ParamsKZG::read: truncated or malformed file " + path);
    const auto reg = precompute ? hm_register_bases_precomp : hm_register_bases;
    arithmetic::check(reg(reinterpret_cast<const uint64_t*>(g_points.data()), n, &g_handle), "ParamsKZG::read");
    arithmetic::check(reg(reinterpret_cast<const uint64_t*>(g_lagrange_points.data()), n, &g_lagrange_handle), "ParamsKZG::read");
  }
  void write(const std::string& path) const {
    if (g_points.size() != n || g_lagrange_points.size() != n)
      throw std::runtime_error("ParamsKZG::write: the affine points were not kept (keep_points)");
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("ParamsKZG::write: cannot open " + path);
    const bool ok = std::fwrite(&k, 4, 1, f) == 1 && std::fwrite(g_points.data(), 64, n, f) == n &&
                    std::fwrite(g_lagrange_points.data(), 64, n, f) == n && std::fwrite(&g2, 128, 1, f) == 1 &&
                    std::fwrite(&s_g2, 128, 1, f) == 1;
    if (std::fclose(f) != 0 || !ok) throw std::runtime_error("ParamsKZG::write: short write to " + path);
  }
  ParamsKZG(const ParamsKZG&) = delete;
  ParamsKZG& operator=(const ParamsKZG&) = delete;
  ~ParamsKZG() {
    if (g_handle) (void)hm_release_bases(g_handle);
    if (g_lagrange_handle) (void)hm_release_bases(g_lagrange_handle);
  }
  // commit(poly in coefficient form) / commit_lagrange(poly in evaluation form): one best_multiexp each
  G1 commit(const Fr* d_poly) const { return msm(g_handle, d_poly); }
  G1 commit_lagrange(const Fr* d_poly) const { return msm(g_lagrange_handle, d_poly); }
  // asynchronous forms: the commitments of one prover phase are independent, so up to eight are kept
  // in flight on different streams (hm_msm_submit_dev) and awaited in order
  uint64_t commit_submit(const Fr* d_poly, hipStream_t stream) const { return submit(g_handle, d_poly, stream); }
  uint64_t commit_lagrange_submit(const Fr* d_poly, hipStream_t stream) const { return submit(g_lagrange_handle, d_poly, stream); }
  static G1 commit_wait(uint64_t ticket) {
    G1 out;
    arithmetic::check(hm_msm_wait(ticket, reinterpret_cast<uint64_t*>(&out)), "commit_wait");
    return out;
  }
  // a whole phase of commitments in one call (hm_msm_batch_bn256_g1_dev): the library keeps eight chains in flight, groups
  // small ones into shared launch chains, and folds on a second thread
  std::vector<G1> commit_batch(const std::vector<const Fr*>& d_polys, bool lagrange, hipStream_t stream = nullptr) const {
    std::vector<G1> out(d_polys.size());
    arithmetic::check(hm_msm_batch_bn256_g1_dev(lagrange ? g_lagrange_handle : g_handle, 0, reinterpret_cast<const void* const*>(d_polys.data()),
                                                n, d_polys.size(), stream, reinterpret_cast<uint64_t*>(out.data())),
                      "commit_batch");
    return out;
  }
  // the same for polynomials held in host vectors (what halo2's prover holds today): uploads pipelined behind the kernels
  std::vector<G1> commit_batch_host(const std::vector<const Fr*>& polys, bool lagrange) const {
    std::vector<G1> out(polys.size());
    arithmetic::check(hm_msm_batch_bn256_g1_h(lagrange ? g_lagrange_handle : g_handle, 0, reinterpret_cast<const uint64_t* const*>(polys.data()),
                                              n, polys.size(), reinterpret_cast<uint64_t*>(out.data())),
                      "commit_batch_host");
    return out;
  }

 private:
  uint64_t submit(uint64_t handle, const Fr* d_poly, hipStream_t stream) const {
    uint64_t ticket = 0;
    arithmetic::check(hm_msm_submit_dev(handle, 0, d_poly, n, stream, &ticket), "commit_submit");
    return ticket;
  }
  G1 msm(uint64_t handle, const Fr* d_poly) const {
    G1 out;
    arithmetic::check(hm_msm_bn256_g1_dev(handle, 0, d_poly, n, nullptr, reinterpret_cast<uint64_t*>(&out)), "commit");
    return out;
  }
  uint64_t fixed_base_set(const DevicePolys& scalars, bool precompute, std::vector<G1Affine>* keep) const {
    G1Affine* d_pts = nullptr;
    if (hipMalloc((void**)&d_pts, n * sizeof(G1Affine)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
    const G1Affine gen = G1Affine::generator();
    uint64_t handle = 0;
    int rc = hm_g1_fixed_base_mul_dev(scalars.d, n, reinterpret_cast<const uint64_t*>(&gen), d_pts, nullptr);
    if (rc == HM_OK)
      rc = precompute ? hm_register_bases_precomp_dev(d_pts, n, nullptr, &handle) : hm_register_bases_dev(d_pts, n, nullptr, &handle);
    if (rc == HM_OK && keep) {
      keep->resize(n);
      if (hipMemcpy(keep->data(), d_pts, n * sizeof(G1Affine), hipMemcpyDeviceToHost) != hipSuccess) rc = HM_ERR_HIP;
    }
    (void)hipFree(d_pts);
    arithmetic::check(rc, "ParamsKZG::setup");
    return handle;
  }
};

}  // namespace poly
}  // namespace halo2
