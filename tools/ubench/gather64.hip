// Microbenchmark: what do the TCC counters report for 64-byte GATHERS?  MI355X_MICROARCH.md calibrates the gfx950
// FETCH_SIZE correction (x2) for wide coalesced streams only; msm_accumulate_kernel reads its bases as 64-byte
// gathers (four dwordx4 per lane at an arbitrary 64-byte-aligned address).  This program issues a KNOWN number of
// such gathers over a table far larger than L2 + MALL and prints the algorithmic byte counts; run it under
// `rocprofv3 --pmc FETCH_SIZE` (tools/pmc_gather.sh) and compare.  Not part of the product path.
//   gather64 [log2 table points = 25] [log2 gathers = 26] [mode: 0 random | 1 sequential]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void gather64_kernel(const uint4* __restrict__ table, const uint32_t* __restrict__ idx,
                                                      uint32_t* __restrict__ out, size_t m) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= m) return;
  const uint4* p = table + (size_t)idx[i] * 4;
  const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  out[i] = a.x ^ b.y ^ c.z ^ d.w;
}

__global__ void fill_idx(uint32_t* idx, size_t m, uint32_t mask, int sequential) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  idx[i] = sequential ? (uint32_t)(i & mask) : (uint32_t)((z ^ (z >> 31)) & mask);
}

int main(int argc, char** argv) {
  const int lt = argc > 1 ? atoi(argv[1]) : 25, lm = argc > 2 ? atoi(argv[2]) : 26, seq = argc > 3 ? atoi(argv[3]) : 0;
  const size_t npts = (size_t)1 << lt, m = (size_t)1 << lm;
  uint4* table = nullptr;
  uint32_t *idx = nullptr, *out = nullptr;
  if (hipMalloc(&table, npts * 64) != hipSuccess || hipMalloc(&idx, m * 4) != hipSuccess || hipMalloc(&out, m * 4) != hipSuccess) {
    fprintf(stderr, "allocation failed\n");
    return 1;
  }
  (void)hipMemset(table, 1, npts * 64);
  hipLaunchKernelGGL(fill_idx, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, idx, m, (uint32_t)(npts - 1), seq);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gather64_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, 0, (const uint4*)table, (const uint32_t*)idx, out, m);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("gather64 mode=%s table=2^%d points (%.1f GiB) gathers=2^%d: %.3f ms; algorithmic bytes per launch: gathered %zu + index %zu = %zu read, %zu written; %.1f GB/s of gathered bytes\n",
           seq ? "sequential" : "random", lt, (double)(npts * 64) / (1 << 30), lm, ms, m * 64, m * 4, m * 68, m * 4, (double)(m * 64) / ms / 1e6);
  }
  return 0;
}
