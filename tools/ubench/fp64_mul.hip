// Microbenchmark: a 254-bit Montgomery product built from v_fma_f64 on 52-bit limbs
// (hi = fma(a, b, 2^104); lo = fma(a, b, (2^104 + 2^52) - hi); 64-bit integer accumulation of the bit patterns:
// the scheme of Emmart et al.) against ff29's fe_mul (v_mad_u64_u32 on 29-bit limbs), same harness, full occupancy.
// The FP64 kernel executes the INSTRUCTION SEQUENCE of such a product (5 x 5 partial products of a * b, five reduction
// rounds of 5 partial products m * p plus the low product that forms m); it runs in the default rounding mode, so its
// values are not a correct product -- round-toward-zero is a mode-register setting, not an instruction -- but its
// instruction count and dependencies are those of the real thing.  Not part of the product path.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../halo2-experiments_amd/csrc/g1.h"

using namespace hm;

constexpr int ITERS = 256;

__device__ __forceinline__ void dp_partial(double a, double b, uint64_t& col_lo, uint64_t& col_hi) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  const double hi = __builtin_fma(a, b, C1);
  const double lo = __builtin_fma(a, b, C2 - hi);
  col_lo += (uint64_t)__double_as_longlong(lo);
  col_hi += (uint64_t)__double_as_longlong(hi);
}

__global__ __launch_bounds__(256) void fp64_product_kernel(double* out, double seed) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  double a[5], b[5], p[5];
  for (int i = 0; i < 5; ++i) {
    a[i] = seed + t + i;
    b[i] = seed * 3 + t * 7 + i;
    p[i] = 4503599627370495.0 - i;
  }
  uint64_t sink = 0;
  for (int it = 0; it < ITERS; ++it) {
    uint64_t col[11] = {};
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) dp_partial(a[i], b[j], col[i + j], col[i + j + 1]);
#pragma unroll
    for (int i = 0; i < 5; ++i) {      // reduction round i: m = low 52 bits of col[i] * inv, then col += m * p
      uint64_t mlo = 0, mhi = 0;
      dp_partial((double)(col[i] & 0xFFFFFFFFFFFFFull), 4503599627370493.0, mlo, mhi);    // int -> fp, low product
      const double m = (double)(mlo & 0xFFFFFFFFFFFFFull);
#pragma unroll
      for (int j = 0; j < 5; ++j) dp_partial(m, p[j], col[i + j], col[i + j + 1]);
      col[i + 1] += col[i] >> 52;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {      // result limbs back to doubles for the next product
      a[i] = (double)(col[5 + i] & 0xFFFFFFFFFFFFFull);
      col[6 + i > 10 ? 10 : 6 + i] += col[5 + i] >> 52;
    }
    sink += col[10];
  }
  out[t] = a[0] + a[1] + a[2] + a[3] + a[4] + (double)sink;
}

__global__ __launch_bounds__(256) void ff29_product_kernel(uint32_t* out, uint32_t seed) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a, b;
  for (int i = 0; i < 9; ++i) {
    a.l[i] = (seed + t * 31 + i * 7) & MASK29;
    b.l[i] = (seed * 5 + t * 17 + i * 3) & MASK29;
  }
  a.l[8] &= 0xFFFF;
  b.l[8] &= 0xFFFF;
  for (int it = 0; it < ITERS; ++it) a = fe_mul(a, b);
  uint32_t s = 0;
  for (int i = 0; i < 9; ++i) s ^= a.l[i];
  out[t] = s;
}

int main() {
  const int blocks = 256 * 16, threads = 256;        // 4 waves per SIMD of work in flight several times over
  double* d_out = nullptr;
  uint32_t* u_out = nullptr;
  (void)hipMalloc(&d_out, (size_t)blocks * threads * 8);
  (void)hipMalloc(&u_out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    float ms_f = 0, ms_i = 0;
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fp64_product_kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 1.5);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms_f, e0, e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(ff29_product_kernel, dim3(blocks), dim3(threads), 0, 0, u_out, 12345u);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms_i, e0, e1);
    const double products = (double)blocks * threads * ITERS;
    printf("fp64 52-bit-limb product sequence: %.3f ms = %.3e products/s;  ff29 fe_mul: %.3f ms = %.3e products/s;  ratio fp64/ff29 time = %.2f\n",
           ms_f, products / ms_f * 1e3, ms_i, products / ms_i * 1e3, ms_f / ms_i);
  }
  return 0;
}
