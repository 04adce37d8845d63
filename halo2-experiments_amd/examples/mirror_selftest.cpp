// mirror_selftest.cpp -- checks of the C++ mirror (cpp/*.hpp); the host-only part runs without a GPU, the rest when one is there.
// Field constants against their defining properties, the public alt_bn128 known answer for 2G, the
// EvaluationDomain constants, and the error behaviour of the arithmetic mirror: length mismatch is
// std::invalid_argument (upstream: assert_eq! panic); a missing device is std::runtime_error -- never
// a CPU fallback.
#include <cstdio>

#include "../cpp/domain.hpp"

using namespace halo2;
using bn256::Fq;
using bn256::Fr;

#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main() {
  // Fr: ROOT_OF_UNITY has order exactly 2^28; ZETA is a primitive cube root of unity
  Fr w = bn256::fr_root_of_unity();
  for (int i = 0; i < 27; ++i) w = w.square();
  CHECK(w != Fr::one());
  CHECK(w.square() == Fr::one());
  const Fr z = bn256::fr_zeta();
  CHECK(z != Fr::one() && z * z * z == Fr::one());
  CHECK(Fr::from_u64(5) * Fr::from_u64(7) == Fr::from_u64(35));
  CHECK(Fr::from_u64(5).invert() * Fr::from_u64(5) == Fr::one());
  CHECK(Fr::from_u64(3) - Fr::from_u64(5) + Fr::from_u64(2) == Fr::zero());
  // Fq / curve: (1, 2) is on y^2 = x^3 + 3; doubling it gives the EIP-196 value of 2G
  const Fq one = Fq::from_u64(1), two = Fq::from_u64(2), three = Fq::from_u64(3);
  CHECK(two.square() == one * one * one + three);
  const Fq lam = (three * one.square()) * (two + two).invert();
  const Fq x2 = lam.square() - one - one, y2 = lam * (one - x2) - two;
  const uint64_t ex[4] = {0xd3c208c16d87cfd3ULL, 0xd97816a916871ca8ULL, 0x9b85045b68181585ULL, 0x030644e72e131a02ULL};
  const uint64_t ey[4] = {0xff3ebf7a5a18a2c4ULL, 0x68a6a449e3538fc7ULL, 0xe7845f96b2ae9c0aULL, 0x15ed738c0e0a7c92ULL};
  CHECK(x2 == Fq::from_raw(ex) && y2 == Fq::from_raw(ey));
  // G2: the generator is on y^2 = x^3 + 3 / (9 + u), (r - 1) G2 = -G2, scalar multiples add up
  {
    using bn256::Fq2;
    using bn256::G2Affine;
    const G2Affine g2 = G2Affine::generator();
    const Fq2 b = Fq2{three, Fq::zero()} * Fq2{Fq::from_u64(9), one}.invert();
    CHECK(g2.y.square() == g2.x.square() * g2.x + b);
    const G2Affine m = g2.mul(-Fr::one());
    const Fq2 neg_y{-g2.y.c0, -g2.y.c1};
    CHECK(m.x == g2.x && m.y == neg_y);
    CHECK(m.add(g2).is_identity());
    CHECK(g2.mul(Fr::from_u64(5)).add(g2.mul(Fr::from_u64(7))) == g2.mul(Fr::from_u64(12)));
    CHECK(g2.mul(Fr::from_u64(2)) == g2.add(g2));
  }
  // EvaluationDomain::new(7, 9): MerkleSumTree at test_full_prover's k
  const poly::EvaluationDomain d(7, 9);
  CHECK(d.extended_k == 12 && d.quotient_poly_degree == 6);
  Fr t = d.omega;
  for (int i = 0; i < 9; ++i) t = t.square();
  CHECK(t == Fr::one());
  CHECK(d.omega * d.omega_inv == Fr::one() && d.ifft_divisor * Fr::from_u64(512) == Fr::one());
  CHECK(d.g_coset * d.g_coset_inv == Fr::one());
  // error behaviour
  std::vector<Fr> s(4, Fr::one());
  std::vector<bn256::G1Affine> b(5, bn256::G1Affine::generator());
  bool threw = false;
  try { (void)arithmetic::best_multiexp(s, b); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  threw = false;
  std::vector<Fr> a(6, Fr::one());
  try { arithmetic::best_fft(a, d.omega, 3); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { (void)d.coeff_to_extended(a); } catch (const std::invalid_argument&) { threw = true; }     // a.len() != n
  CHECK(threw);
  threw = false;
  try { (void)d.extended_to_coeff(a); } catch (const std::invalid_argument&) { threw = true; }     // a.len() != extended_len()
  CHECK(threw);
  if (hm_device_count() > 0) {
    // host-vector EvaluationDomain steps: coeff -> extended -> coeff returns the coefficients, zeros above n, n * (j - 1) of them
    std::vector<Fr> c(d.n);
    for (size_t i = 0; i < d.n; ++i) c[i] = Fr::from_u64(3 * i + 1);
    const std::vector<Fr> ext = d.coeff_to_extended(c);
    CHECK(ext.size() == d.extended_len());
    Fr at_zeta = Fr::zero();                                  // row 0 of the extended array is the value at zeta: Horner
    for (size_t i = d.n; i-- > 0;) at_zeta = at_zeta * d.g_coset + c[i];
    CHECK(ext[0] == at_zeta);
    const std::vector<Fr> back = d.extended_to_coeff(ext);
    CHECK(back.size() == d.n * d.quotient_poly_degree);
    for (size_t i = 0; i < back.size(); ++i) CHECK(back[i] == (i < d.n ? c[i] : Fr::zero()));
  }
  if (hm_device_count() == 0) {
    threw = false;
    b.resize(4);
    try { (void)arithmetic::best_multiexp(s, b); } catch (const std::runtime_error& e) {
      threw = std::string(e.what()).find("no CPU fallback") != std::string::npos;
    }
    CHECK(threw);
  }
  std::printf("mirror selftest ok\n");
  return 0;
}
