// Microbenchmark: per-instruction issue rate of the integer/FP64 VALU ops a 256-bit
// Montgomery multiplier can be built from on gfx950. Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a0 = seed ^ t, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 11;
  uint32_t b = t * 2654435761u + 12345u, c = ~t;
  uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0000001, dc = 0.5;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {  // v_mad_u64_u32
      REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
                         "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b), "v"(c) : "vcc");)
    } else if constexpr (OP == 1) {  // v_mul_lo_u32
      REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if constexpr (OP == 2) {  // v_mul_hi_u32
      REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if constexpr (OP == 3) {  // v_mad_u32_u24
      REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n\tv_mad_u32_u24 %1, %1, %4, %5\n\tv_mad_u32_u24 %2, %2, %4, %5\n\tv_mad_u32_u24 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    } else if constexpr (OP == 4) {  // v_mul_hi_u32_u24
      REP64(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\tv_mul_hi_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if constexpr (OP == 5) {  // v_fma_f64
      REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
    } else if constexpr (OP == 6) {  // v_add_co_u32 + v_addc_co_u32 pair
      REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
    } else if constexpr (OP == 7) {  // v_add3_u32
      REP64(asm volatile("v_add3_u32 %0, %0, %4, %5\n\tv_add3_u32 %1, %1, %4, %5\n\tv_add3_u32 %2, %2, %4, %5\n\tv_add3_u32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    } else if constexpr (OP == 8) {  // v_mul_u32_u24
      REP64(asm volatile("v_mul_u32_u24 %0, %0, %4\n\tv_mul_u32_u24 %1, %1, %4\n\tv_mul_u32_u24 %2, %2, %4\n\tv_mul_u32_u24 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
    } else if constexpr (OP == 9) {  // v_mad_u64_u32 dependent chain of 1 (latency)
      REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                         : "+v"(q0) : "v"(b), "v"(c) : "vcc");)
    } else if constexpr (OP == 10) {  // v_lshl_add_u64 (64-bit add)
      REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0 ^ b));)
    } else if constexpr (OP == 12) {  // v_fma_f32 calibration
      REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    } else if constexpr (OP == 13) {  // mixed: 1 mad_u64_u32 : 1 add_co : 1 addc
      REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_add_co_u32 %2, vcc, %2, %4\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1"
                         : "+v"(q0), "+v"(q1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
    } else if constexpr (OP == 11) {  // v_mad_i32_i24
      REP64(asm volatile("v_mad_i32_i24 %0, %0, %4, %5\n\tv_mad_i32_i24 %1, %1, %4, %5\n\tv_mad_i32_i24 %2, %2, %4, %5\n\tv_mad_i32_i24 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    }
  }
  out[t] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)q0 ^ (uint32_t)q1 ^ (uint32_t)q2 ^ (uint32_t)q3 ^
           (uint32_t)(q0 >> 32) ^ (uint32_t)d0 ^ (uint32_t)d1 ^ (uint32_t)d2 ^ (uint32_t)d3;
}

template <int OP>
void run(const char* name, uint32_t* d, int wavesPerSimd) {
  const int iters = 4000;
  int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves = 1 wave per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 64 * 4;
  double waves = (double)blocks * 4;
  double total = instr_per_wave * waves;                 // wave-instructions
  double per_simd_per_s = total / 1024.0 / (ms * 1e-3);  // wave-instr per SIMD per s
  double cyc = 2.4e9 / per_simd_per_s;                   // cycles per wave-instr at nominal 2.4 GHz
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cyc/wave-instr(@2.4GHz)  %.2f Tlaneops/s\n", name, wavesPerSimd, ms, cyc,
         total * 64 / (ms * 1e-3) / 1e12);
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
  for (int w : {1, 4, 8}) {
    run<0>("v_mad_u64_u32", d, w);
    run<1>("v_mul_lo_u32", d, w);
    run<2>("v_mul_hi_u32", d, w);
    run<3>("v_mad_u32_u24", d, w);
    run<4>("v_mul_hi_u32_u24", d, w);
    run<5>("v_fma_f64", d, w);
    run<6>("v_add_co/addc_co_u32", d, w);
    run<7>("v_add3_u32", d, w);
    run<8>("v_mul_u32_u24", d, w);
    run<9>("v_mad_u64_u32 (dep chain)", d, w);
    run<10>("v_lshl_add_u64", d, w);
    run<11>("v_mad_i32_i24", d, w);
    run<12>("v_fma_f32", d, w);
    run<13>("mix mad64/add_co/addc", d, w);
  }
  return 0;
}
