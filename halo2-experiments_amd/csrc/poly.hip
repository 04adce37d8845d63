// poly.hip -- Fr polynomial kernels next to the NTT (first half): eval_polynomial, the
// Horner evaluations create_proof makes of every committed polynomial at x * omega^rot (upstream
// halo2_proofs::arithmetic::eval_polynomial, called from plonk/prover.rs at the tag pinned by
// /root/reference/Cargo.toml:10; reference call site /root/reference/src/circuits/utils.rs:40-48).
//
// out[q] = sum_i poly[i] * x_q^i for device-resident coefficient arrays.  Like the NTT, the map is linear in
// the coefficients, so the reference's radix-2^256 Montgomery words are read as internal-form values (of
// a / 32) without conversion; only the point is converted (on the host, by value).
//   kernel 1  grid (blocks, queries): a workgroup owns 256 * CH consecutive coefficients; lane t runs Horner in
//             y = x^256 over the coalesced column t, t + 256, ...; the lane results are weighted by x^t and
//             folded by an LDS tree -> one partial per block
//   kernel 2  one workgroup per query: partial_b * (x^(256 CH))^b, tree, canonical external words
// CH is chosen so that a query has at most 256 blocks.  Bound: every intermediate is a product output (< 2r)
// or a reduced sum (< 3r); sums of two are brought back below 3r before the next level.
#include <hip/hip_runtime.h>

#include "g1.h"
#include "hm_internal.h"
#include "host_fr.h"

namespace hm {

constexpr int EV_THREADS = 256;

struct FrInt9 {          // one field element in the 9 x 29-bit internal form, passed by value
  uint32_t l[9];
};

__device__ __forceinline__ Fr fr_from_arg(const FrInt9& a) {
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i];
  HM_DECLARE(r, 1.0);
  return r;
}

// x^e for a small exponent (square-and-multiply from the top bit)
__device__ __forceinline__ Fr fr_pow_small(const Fr& x, uint32_t e) {
  Fr acc = fe_one<FrParams>();
  for (int bit = 31 - __clz(e | 1u); bit >= 0; --bit) {
    acc = fe_sqr(acc);
    if ((e >> bit) & 1u) acc = fe_mul(acc, x);
  }
  return acc;
}

// a + b for values < 3r each, brought back below 3r (normalised limbs)
__device__ __forceinline__ Fr fr_add_reduced(const Fr& a, const Fr& b) { return fe_reduce_small(fe_norm(fe_add(a, b))); }

// sum of one value (< 3r, normalised) per lane over the 256-lane workgroup; valid in lane 0
__device__ __forceinline__ Fr block_sum_fr(uint32_t* lds, Fr v) {
  const uint32_t t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 9; ++i) lds[i * EV_THREADS + t] = v.l[i];
  __syncthreads();
  for (uint32_t off = EV_THREADS / 2; off > 0; off >>= 1) {
    if (t < off) {
      Fr a, b;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        a.l[i] = lds[i * EV_THREADS + t];
        b.l[i] = lds[i * EV_THREADS + t + off];
      }
      HM_DECLARE(a, 3.0);
      HM_DECLARE(b, 3.0);
      const Fr s = fr_add_reduced(a, b);
#pragma unroll
      for (int i = 0; i < 9; ++i) lds[i * EV_THREADS + t] = s.l[i];
    }
    __syncthreads();
  }
  Fr r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = lds[i * EV_THREADS];
  HM_DECLARE(r, 3.0);
  return r;
}

struct FrWordsArg {      // one external field element, passed by value
  uint32_t w[8];
};

struct EvalQuery {
  uint32_t poly;       // index of the coefficient array
  FrInt9 x;            // the point, internal form
};
constexpr int EV_MAX_Q = 48;     // queries per launch (by-value argument block: 48 x 40 B)
struct EvalQueries {
  EvalQuery q[EV_MAX_Q];
};

__global__ __launch_bounds__(EV_THREADS) void fr_eval_partial_kernel(const uint32_t* __restrict__ polys, uint64_t n, EvalQueries qs,
                                                                     uint32_t CH, uint32_t* __restrict__ partial) {
  __shared__ uint32_t lds[9 * EV_THREADS];
  const uint32_t t = threadIdx.x, b = blockIdx.x, qi = blockIdx.y, B = gridDim.x;
  const Fr x = fr_from_arg(qs.q[qi].x);
  Fr y = x;                                        // y = x^256
#pragma unroll 1
  for (int k = 0; k < 8; ++k) y = fe_sqr(y);
  const uint32_t* a = polys + (size_t)qs.q[qi].poly * n * 8;
  const uint64_t base = (uint64_t)b * EV_THREADS * CH;
  Fr acc = fe_zero<FrParams>();
  HM_DECLARE(acc, 0.0);
  for (int k = (int)CH - 1; k >= 0; --k) {
    const uint64_t idx = base + (uint64_t)k * EV_THREADS + t;
    Fr c = fe_zero<FrParams>();
    if (idx < n) {
      const uint4* src = reinterpret_cast<const uint4*>(a + idx * 8);
      const uint4 lo = src[0], hi = src[1];
      const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      c = fe_unpack<FrParams>(w);
    }
    acc = fe_add(fe_mul(acc, y), c);               // limbs < 2^30: fine as the next product's operand
  }
  const Fr term = fe_mul(acc, fr_pow_small(x, t));   // column t carries x^t
  const Fr s = block_sum_fr(lds, term);
  if (t == 0) {
    uint32_t* o = partial + ((size_t)qi * B + b) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = s.l[i];
  }
}

__global__ __launch_bounds__(EV_THREADS) void fr_eval_final_kernel(const uint32_t* __restrict__ partial, EvalQueries qs, uint32_t B,
                                                                   uint32_t log_block, uint32_t CH, uint32_t* __restrict__ out_ext) {
  __shared__ uint32_t lds[9 * EV_THREADS];
  const uint32_t t = threadIdx.x, qi = blockIdx.x;
  const Fr x = fr_from_arg(qs.q[qi].x);
  Fr z = x;                                        // z = x^(256 * CH): 8 squarings, then the CH-th power
#pragma unroll 1
  for (uint32_t k = 0; k < log_block; ++k) z = fe_sqr(z);
  z = fr_pow_small(z, CH);
  Fr v = fe_zero<FrParams>();
  HM_DECLARE(v, 0.0);
  if (t < B) {
    Fr p;
    const uint32_t* src = partial + ((size_t)qi * B + t) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) p.l[i] = src[i];
    HM_DECLARE(p, 3.0);
    v = fe_mul(p, fr_pow_small(z, t));
  }
  const Fr s = block_sum_fr(lds, v);
  if (t == 0) {
    // the coefficients were read as internal values of a / 32, so s is the internal form of f(x) / 32,
    // i.e. exactly the external Montgomery words of f(x)
    const Fr c = fe_canonical(s);
    uint32_t w[8];
    fe_pack(w, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) out_ext[(size_t)qi * 8 + i] = w[i];
  }
}

// out[i] = x^i, external Montgomery words (ParamsKZG::setup's scalar ladder 1, s, s^2, ...: upstream builds it
// serially on the CPU before the fixed-base multiplications)
__global__ __launch_bounds__(EV_THREADS) void fr_powers_kernel(uint32_t* __restrict__ out, uint64_t n, FrInt9 x_int) {
  const uint64_t i = (uint64_t)blockIdx.x * EV_THREADS + threadIdx.x;
  if (i >= n) return;
  const Fr x = fr_from_arg(x_int);
  Fr acc = fe_one<FrParams>();
  for (int bit = 63 - __clzll((unsigned long long)(i | 1ull)); bit >= 0; --bit) {
    acc = fe_sqr(acc);
    if ((i >> bit) & 1ull) acc = fe_mul(acc, x);
  }
  uint32_t w[8];
  fe_to_ext(w, acc);
  uint4* dst = reinterpret_cast<uint4*>(out + i * 8);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

int fr_powers_run(uint32_t* d_out, uint64_t n, const uint64_t x_ext[4], hipStream_t stream) {
  if (n == 0) return HM_OK;
  FrInt9 x;
  host::fr_to_internal9(host::fr_load(x_ext), x.l);
  hipLaunchKernelGGL(fr_powers_kernel, dim3((uint32_t)((n + EV_THREADS - 1) / EV_THREADS)), dim3(EV_THREADS), 0, stream, d_out, n, x);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

// ---- inner product, arithmetic progression, uniform sampling: the inputs and the known answer of the
// benchmark of SURVEY.md §8d (bases [a + i b]G, scalars uniform in [0, r) from xoshiro256**, expected result
// [sum_i s_i (a + i b)]G) -- produced and checked without leaving the device ------------------------------
__global__ __launch_bounds__(EV_THREADS) void fr_dot_partial_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                                    uint64_t n, uint32_t per, uint32_t* __restrict__ partial) {
  __shared__ uint32_t lds[9 * EV_THREADS];
  const uint32_t t = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * EV_THREADS * per;
  Fr acc = fe_zero<FrParams>();
  HM_DECLARE(acc, 0.0);
  for (uint32_t k = 0; k < per; ++k) {
    const uint64_t idx = base + (uint64_t)k * EV_THREADS + t;
    if (idx < n) {
      const uint4* pa = reinterpret_cast<const uint4*>(a + idx * 8);
      const uint4* pb = reinterpret_cast<const uint4*>(b + idx * 8);
      const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
      const uint32_t wa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const uint32_t wb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      acc = fr_add_reduced(acc, fe_mul(fe_unpack<FrParams>(wa), fe_unpack<FrParams>(wb)));
    }
  }
  const Fr s = block_sum_fr(lds, acc);
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) partial[(size_t)blockIdx.x * 9 + i] = s.l[i];
  }
}

__global__ __launch_bounds__(EV_THREADS) void fr_dot_final_kernel(const uint32_t* __restrict__ partial, uint32_t B, FrInt9 k32_int,
                                                                  uint32_t* __restrict__ out_ext) {
  __shared__ uint32_t lds[9 * EV_THREADS];
  const uint32_t t = threadIdx.x;
  Fr acc = fe_zero<FrParams>();
  HM_DECLARE(acc, 0.0);
  for (uint32_t i = t; i < B; i += EV_THREADS) {
    Fr p;
#pragma unroll
    for (int j = 0; j < 9; ++j) p.l[j] = partial[(size_t)i * 9 + j];
    HM_DECLARE(p, 3.0);
    acc = fr_add_reduced(acc, p);
  }
  const Fr s = block_sum_fr(lds, acc);
  if (t == 0) {
    // a_ext * b_ext * 2^-261 = a b 2^251: times 32 (internal form) gives a b 2^256, the external words
    const Fr c = fe_canonical(fe_mul(s, fr_from_arg(k32_int)));
    uint32_t w[8];
    fe_pack(w, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) out_ext[i] = w[i];
  }
}

// out[i] = a + i * b (external words); bx = b's external words re-read as a value and converted (host) to internal
__global__ __launch_bounds__(EV_THREADS) void fr_affine_sequence_kernel(uint32_t* __restrict__ out, uint64_t n, FrWordsArg a_ext,
                                                                        FrInt9 bx_int) {
  const uint64_t i = (uint64_t)blockIdx.x * EV_THREADS + threadIdx.x;
  if (i >= n) return;
  Fr iv = fe_zero<FrParams>();                         // the raw integer i in 29-bit limbs
  iv.l[0] = (uint32_t)(i & MASK29);
  iv.l[1] = (uint32_t)((i >> 29) & MASK29);
  iv.l[2] = (uint32_t)(i >> 58);
  HM_DECLARE(iv, 1.0);
  uint32_t wa[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) wa[k] = a_ext.w[k];
  // (b 2^256 2^261) * i * 2^-261 = i b 2^256 = ext(i b); plus ext(a) (canonical, < r): < 3r
  const Fr y = fe_canonical(fe_reduce_small(fe_norm(fe_add(fe_mul(fr_from_arg(bx_int), iv), fe_unpack<FrParams>(wa)))));
  uint32_t w[8];
  fe_pack(w, y);
  uint4* dst = reinterpret_cast<uint4*>(out + i * 8);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

// out[i] uniform in [0, r) as external words: element i runs its own xoshiro256** stream, seeded by splitmix64 from
// (seed, i); 254-bit candidates are rejected until one is below r (p(accept) = 0.76).  The words are stored as they
// are: a uniform canonical integer read as Montgomery words is a uniform field element.
__global__ __launch_bounds__(EV_THREADS) void fr_random_kernel(uint32_t* __restrict__ out, uint64_t n, uint64_t seed) {
  const uint64_t i = (uint64_t)blockIdx.x * EV_THREADS + threadIdx.x;
  if (i >= n) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1), st[4];
  for (int k = 0; k < 4; ++k) {                        // splitmix64
    z += 0x9E3779B97F4A7C15ull;
    uint64_t x = z;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    st[k] = x ^ (x >> 31);
  }
  uint64_t v[4];
  for (int attempt = 0; attempt < 64; ++attempt) {
    for (int k = 0; k < 4; ++k) {                      // xoshiro256**
      v[k] = rotl64(st[1] * 5, 7) * 9;
      const uint64_t t = st[1] << 17;
      st[2] ^= st[0];
      st[3] ^= st[1];
      st[1] ^= st[2];
      st[0] ^= st[3];
      st[2] ^= t;
      st[3] = rotl64(st[3], 45);
    }
    v[3] &= 0x3FFFFFFFFFFFFFFFull;                     // 254 bits
    const uint64_t m[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    bool below = false;
    for (int k = 3; k >= 0; --k) {
      if (v[k] != m[k]) {
        below = v[k] < m[k];
        break;
      }
    }
    if (below) break;
    if (attempt == 63) v[3] = 0;                       // unreachable in practice (0.24^64); keeps the output in range
  }
  uint4* dst = reinterpret_cast<uint4*>(out + i * 8);
  dst[0] = make_uint4((uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32));
  dst[1] = make_uint4((uint32_t)v[2], (uint32_t)(v[2] >> 32), (uint32_t)v[3], (uint32_t)(v[3] >> 32));
}

int fr_dot_run(DeviceCtx& ctx, const uint32_t* d_a, const uint32_t* d_b, uint64_t n, uint64_t out_ext[4], hipStream_t stream) {
  if (n == 0) {
    std::memset(out_ext, 0, 32);
    return HM_OK;
  }
  uint32_t per = (uint32_t)((n + (uint64_t)EV_THREADS * 4096 - 1) / ((uint64_t)EV_THREADS * 4096));   // <= 4096 blocks
  if (per < 8) per = 8;
  const uint32_t B = (uint32_t)((n + (uint64_t)EV_THREADS * per - 1) / ((uint64_t)EV_THREADS * per));
  AuxSlot* slot = aux_acquire(ctx, stream);
  if (!slot) return HM_ERR_HIP;
  const size_t need = (size_t)B * 36 + 32;
  uint8_t* buf = (uint8_t*)slot->table.ensure(need > ((size_t)64 * 15 * 28 * 4) ? need : ((size_t)64 * 15 * 28 * 4));
  if (!buf) return hm_fail(HM_ERR_HIP, "fr_dot: scratch allocation failed");
  uint32_t* d_partial = (uint32_t*)buf;
  uint32_t* d_out = (uint32_t*)(buf + (size_t)B * 36);
  FrInt9 k32;
  host::fr_to_internal9(host::FR_32, k32.l);
  hipLaunchKernelGGL(fr_dot_partial_kernel, dim3(B), dim3(EV_THREADS), 0, stream, d_a, d_b, n, per, d_partial);
  hipLaunchKernelGGL(fr_dot_final_kernel, dim3(1), dim3(EV_THREADS), 0, stream, (const uint32_t*)d_partial, B, k32, d_out);
  HM_HIP_CHECK(hipGetLastError());
  HM_HIP_CHECK(hipMemcpyAsync(out_ext, d_out, 32, hipMemcpyDeviceToHost, stream));
  HM_HIP_CHECK(hipStreamSynchronize(stream));
  return aux_release(ctx, slot, stream);
}

int fr_affine_sequence_run(uint32_t* d_out, uint64_t n, const uint64_t a_ext[4], const uint64_t b_ext[4], hipStream_t stream) {
  if (n == 0) return HM_OK;
  FrWordsArg a;
  std::memcpy(a.w, a_ext, 32);
  // The kernel multiplies by the raw integer i, so it wants the integer b_ext * 2^261 (then the product is the
  // integer i * b_ext = ext(i b)).  fr_to_internal9 maps an integer u to 32 u; fed with b_ext * 2^256 (one more
  // Montgomery factor: a product by 2^512 mod r) it yields b_ext * 2^261.
  const host::Fr4 R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};   // 2^512 mod r
  const host::Fr4 b_twice = host::fr_mul(host::fr_load(b_ext), R2);       // b_ext * 2^512 / 2^256 = b_ext * 2^256
  FrInt9 bx;
  host::fr_to_internal9(b_twice, bx.l);
  hipLaunchKernelGGL(fr_affine_sequence_kernel, dim3((uint32_t)((n + EV_THREADS - 1) / EV_THREADS)), dim3(EV_THREADS), 0, stream, d_out, n,
                     a, bx);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int fr_random_run(uint32_t* d_out, uint64_t n, uint64_t seed, hipStream_t stream) {
  if (n == 0) return HM_OK;
  hipLaunchKernelGGL(fr_random_kernel, dim3((uint32_t)((n + EV_THREADS - 1) / EV_THREADS)), dim3(EV_THREADS), 0, stream, d_out, n, seed);
  HM_HIP_CHECK(hipGetLastError());
  return HM_OK;
}

int fr_eval_polynomial_run(DeviceCtx& ctx, const uint32_t* d_polys, uint64_t n, const uint32_t* poly_index, const uint64_t* points_ext,
                           size_t q, uint64_t* out_ext, hipStream_t stream) {
  if (q == 0) return HM_OK;
  if (n == 0) {
    std::memset(out_ext, 0, q * 32);
    return HM_OK;
  }
  // at most 256 blocks of 256 * CH coefficients per query
  uint32_t CH = (uint32_t)((n + (uint64_t)EV_THREADS * 256 - 1) / ((uint64_t)EV_THREADS * 256));
  if (CH < 16) CH = n < 16 * EV_THREADS ? (uint32_t)((n + EV_THREADS - 1) / EV_THREADS) : 16;
  if (CH == 0) CH = 1;
  const uint32_t B = (uint32_t)((n + (uint64_t)EV_THREADS * CH - 1) / ((uint64_t)EV_THREADS * CH));
  if (B > EV_THREADS) return hm_fail(HM_ERR_INTERNAL, "eval_polynomial: block plan exceeds one final workgroup");
  AuxSlot* slot = aux_acquire(ctx, stream);
  if (!slot) return HM_ERR_HIP;
  const size_t per_launch = (size_t)EV_MAX_Q * B * 36 + (size_t)EV_MAX_Q * 32;
  uint8_t* buf = (uint8_t*)slot->table.ensure(per_launch > ((size_t)64 * 15 * 28 * 4) ? per_launch : ((size_t)64 * 15 * 28 * 4));
  if (!buf) return hm_fail(HM_ERR_HIP, "eval_polynomial: scratch allocation failed");
  uint32_t* d_partial = (uint32_t*)buf;
  uint32_t* d_out = (uint32_t*)(buf + (size_t)EV_MAX_Q * B * 36);
  for (size_t q0 = 0; q0 < q; q0 += EV_MAX_Q) {
    const uint32_t cnt = (uint32_t)(q - q0 < (size_t)EV_MAX_Q ? q - q0 : (size_t)EV_MAX_Q);
    EvalQueries qs;
    std::memset(&qs, 0, sizeof qs);
    for (uint32_t i = 0; i < cnt; ++i) {
      qs.q[i].poly = poly_index ? poly_index[q0 + i] : (uint32_t)(q0 + i);
      host::fr_to_internal9(host::fr_load(points_ext + (q0 + i) * 4), qs.q[i].x.l);
    }
    // Coefficients per lane of THIS launch.  A lane pays ~25 products whatever its share (x^256, x^t, the block sum), so with
    // many queries in the launch -- the 94 openings of a k = 18 proof go 48 at a time -- longer shares are cheaper as long as
    // the launch still has ~1 000 workgroups: 16 per lane costs 2.5 products per coefficient, 64 per lane 1.4.
    uint32_t CHq = CH;
    {
      const uint32_t want_blocks = 1024 / cnt ? 1024 / cnt : 1;                        // workgroups per query
      const uint64_t ch = (n + (uint64_t)EV_THREADS * want_blocks - 1) / ((uint64_t)EV_THREADS * want_blocks);
      if (ch > CHq) CHq = (uint32_t)(ch > 256 ? 256 : ch);
    }
    const uint32_t Bq = (uint32_t)((n + (uint64_t)EV_THREADS * CHq - 1) / ((uint64_t)EV_THREADS * CHq));     // <= B
    hipLaunchKernelGGL(fr_eval_partial_kernel, dim3(Bq, cnt), dim3(EV_THREADS), 0, stream, d_polys, n, qs, CHq, d_partial);
    hipLaunchKernelGGL(fr_eval_final_kernel, dim3(cnt), dim3(EV_THREADS), 0, stream, (const uint32_t*)d_partial, qs, Bq, 8u, CHq, d_out);
    HM_HIP_CHECK(hipGetLastError());
    HM_HIP_CHECK(hipMemcpyAsync(out_ext + q0 * 4, d_out, (size_t)cnt * 32, hipMemcpyDeviceToHost, stream));
    HM_HIP_CHECK(hipStreamSynchronize(stream));      // the results go to the caller's (pageable) memory
  }
  return aux_release(ctx, slot, stream);
}

}  // namespace hm
