// arithmetic.hpp -- C++ mirror of halo2_proofs::arithmetic for BN256 (the crate pinned at
// /root/reference/Cargo.toml:10, tag v2023_02_02), over the C ABI of libhalo2_mi355x.so:
//
//     pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve
//     pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32)
//
// Same names, argument meaning and failure behaviour: upstream `assert_eq!`s on the lengths (a
// panic); here that is std::invalid_argument.  A backend failure (no gfx950 device, HIP error) is
// std::runtime_error carrying hm_last_error() -- there is no CPU fallback in this library.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/halo2_mi355x.h"
#include "bn256.hpp"

namespace halo2 {
namespace arithmetic {

using bn256::Fr;
using bn256::G1;
using bn256::G1Affine;

inline void check(int rc, const char* what) {
  if (rc != HM_OK) throw std::runtime_error(std::string(what) + ": " + hm_last_error());
}

inline G1 best_multiexp(const Fr* coeffs, size_t n_coeffs, const G1Affine* bases, size_t n_bases) {
  if (n_coeffs != n_bases) throw std::invalid_argument("best_multiexp: coeffs.len() != bases.len()");
  G1 out;
  check(hm_msm_bn256_g1_jacobian(reinterpret_cast<const uint64_t*>(coeffs), reinterpret_cast<const uint64_t*>(bases), n_coeffs,
                                 reinterpret_cast<uint64_t*>(&out)),
        "best_multiexp");
  return out;
}
inline G1 best_multiexp(const std::vector<Fr>& coeffs, const std::vector<G1Affine>& bases) {
  return best_multiexp(coeffs.data(), coeffs.size(), bases.data(), bases.size());
}

inline void best_fft(Fr* a, size_t len, const Fr& omega, uint32_t log_n) {
  if (log_n > 63 || len != ((size_t)1 << log_n)) throw std::invalid_argument("best_fft: a.len() != 1 << log_n");
  check(hm_ntt_bn256_fr(reinterpret_cast<uint64_t*>(a), omega.l, log_n), "best_fft");
}
inline void best_fft(std::vector<Fr>& a, const Fr& omega, uint32_t log_n) { best_fft(a.data(), a.size(), omega, log_n); }

// eval_polynomial(poly, point) for a coefficient array resident in HBM (upstream: halo2_proofs::arithmetic::
// eval_polynomial on a host slice); `queries` polynomials / points at once through one pair of launches
inline std::vector<Fr> eval_polynomial(const Fr* d_polys, size_t n, const std::vector<uint32_t>& poly_index,
                                       const std::vector<Fr>& points, void* stream = nullptr) {
  if (!poly_index.empty() && poly_index.size() != points.size())
    throw std::invalid_argument("eval_polynomial: one polynomial index per point");
  std::vector<Fr> out(points.size());
  check(hm_eval_polynomial_bn256_fr_dev(d_polys, n, poly_index.empty() ? nullptr : poly_index.data(),
                                        reinterpret_cast<const uint64_t*>(points.data()), points.size(),
                                        reinterpret_cast<uint64_t*>(out.data()), stream),
        "eval_polynomial");
  return out;
}
inline Fr eval_polynomial(const Fr* d_poly, size_t n, const Fr& point, void* stream = nullptr) {
  return eval_polynomial(d_poly, n, std::vector<uint32_t>{}, std::vector<Fr>{point}, stream)[0];
}

// kate_division(a, z) for a coefficient array resident in HBM: d_quotient receives n - 1 coefficients (upstream returns
// a new Vec; the two arrays must not overlap).  Upstream panics on an empty slice (a.len() - 1 underflows).
inline void kate_division(const Fr* d_poly, size_t n, const Fr& z, Fr* d_quotient, void* stream = nullptr) {
  if (n == 0) throw std::invalid_argument("kate_division: empty polynomial");
  check(hm_kate_division_bn256_fr_dev(d_poly, n, z.l, d_quotient, stream), "kate_division");
}

// the z column of the permutation / lookup arguments: out[0] = start, out[i] = out[i - 1] * factors[i - 1]
inline void grand_product(const Fr* d_factors, size_t n, const Fr& start, Fr* d_out, void* stream = nullptr) {
  check(hm_fr_grand_product_dev(d_factors, n, start.l, d_out, stream), "grand_product");
}

// several kate_divisions of same-length polynomials, each by its own point, in one launch chain (the quotients of one
// multiopen round); no quotient may overlap any polynomial of the call
inline void kate_division_batch(const std::vector<const Fr*>& d_polys, size_t n, const std::vector<Fr>& zs, const std::vector<Fr*>& d_quotients,
                                void* stream = nullptr) {
  if (n == 0) throw std::invalid_argument("kate_division_batch: empty polynomial");
  if (d_polys.size() != zs.size() || d_polys.size() != d_quotients.size())
    throw std::invalid_argument("kate_division_batch: one point and one quotient per polynomial");
  check(hm_kate_division_batch_bn256_fr_dev(reinterpret_cast<const void* const*>(d_polys.data()), n, reinterpret_cast<const uint64_t*>(zs.data()),
                                            reinterpret_cast<void* const*>(d_quotients.data()), d_polys.size(), stream),
        "kate_division_batch");
}

// the z columns of one argument in one launch chain.  chain_row == HM_NO_CHAIN: every column starts from `start` (lookup
// arguments); chain_row = n - (blinding_factors + 1): column j + 1 starts from d_out[j][chain_row], upstream's `last_z`
// of the permutation argument
inline void grand_product_batch(const std::vector<const Fr*>& d_factors, size_t n, const Fr& start, size_t chain_row,
                                const std::vector<Fr*>& d_out, void* stream = nullptr) {
  if (d_factors.size() != d_out.size()) throw std::invalid_argument("grand_product_batch: one output per column");
  if (chain_row != HM_NO_CHAIN && chain_row >= n) throw std::invalid_argument("grand_product_batch: chain_row outside the columns");
  check(hm_fr_grand_product_batch_dev(reinterpret_cast<const void* const*>(d_factors.data()), n, start.l, chain_row,
                                      reinterpret_cast<void* const*>(d_out.data()), d_factors.size(), stream),
        "grand_product_batch");
}

// ff::BatchInvert::batch_invert on a device-resident slice: zero stays zero
inline void batch_invert(Fr* d_values, size_t n, void* stream = nullptr) {
  check(hm_fr_batch_invert_dev(d_values, n, stream), "batch_invert");
}

// sum_j coeffs[j] * polys[j] (upstream `Polynomial * F` and `+`); d_out may be one of the inputs
inline void linear_combination(const std::vector<const Fr*>& d_polys, const std::vector<Fr>& coeffs, size_t n, Fr* d_out,
                               void* stream = nullptr) {
  if (d_polys.size() != coeffs.size()) throw std::invalid_argument("linear_combination: one coefficient per polynomial");
  check(hm_fr_linear_combination_dev(reinterpret_cast<const void* const*>(d_polys.data()), reinterpret_cast<const uint64_t*>(coeffs.data()),
                                     d_polys.size(), n, d_out, stream),
        "linear_combination");
}

// plonk::lookup::prover::permute_expression_pair on device-resident columns: rows [0, usable_rows) of d_permuted_input /
// d_permuted_table are written (the blinding rows beyond are the caller's, as upstream appends random values).  An input
// value missing from the table is upstream's Error::ConstraintSystemFailure: here std::runtime_error.
inline void permute_expression_pair(const Fr* d_input, const Fr* d_table, size_t usable_rows, Fr* d_permuted_input, Fr* d_permuted_table,
                                    void* stream = nullptr) {
  check(hm_lookup_permute_bn256_fr_dev(d_input, d_table, usable_rows, d_permuted_input, d_permuted_table, stream), "permute_expression_pair");
}

// One process, several GPUs (what the reference's prover is: /root/reference/src/circuits/utils.rs:22-70): after this call
// ParamsKZG registrations replicate (or slice) their base sets over `devices` and best_multiexp / commit phases are split
// inside the library (csrc/multi.hip).  An empty list restores the calling thread's device.
inline void use_devices(const std::vector<int>& devices) {
  check(hm_set_msm_devices(devices.empty() ? nullptr : devices.data(), (int)devices.size()), "use_devices");
}

// affine normalisation of a G1 the library returned: it is already (x, y, 1) or the identity
inline G1Affine to_affine(const G1& p) {
  if (p.is_identity()) return G1Affine::identity();
  return G1Affine{p.x, p.y};
}

}  // namespace arithmetic
}  // namespace halo2
