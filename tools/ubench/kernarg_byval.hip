// kernarg_byval.hip -- what happens to a 2.9 KiB BY-VALUE kernel argument that the kernel indexes dynamically?
// (round 2's evaluate_h kernel took its column table -- 256 pointers, 16 per-proof constants, a byte table of
// column periods -- as one struct by value; "a variant with a byte table in it aborted at run time on ROCm 7.2 and the cause
// was never pinned down".  This reproduces the SHAPE of that argument in a stand-alone kernel, so that the compiler's
// treatment can be read from the code object (kernarg segment size, private segment = scratch per lane) and, with `run`,
// one launch can be tried outside the library.)
//   hipcc -O3 --offload-arch=gfx950 --save-temps -c tools/ubench/kernarg_byval.hip      # read .kernarg_segment_size / .private_segment_fixed_size
//   (launching: see main -- only the variant without the byte table, and only on request)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

struct Columns {
  const uint32_t* p[256];
  uint32_t dyn[16 * 9];
  uint32_t n_static;
  uint8_t log_rows[256];      // the "byte table"
};

// index comes from memory (the program's source words): dynamic, and as far as the compiler knows per-lane.
// BYTES = false: the same kernel without the byte table (pointer and word tables only: the compiler reads those with
// scalar loads from the kernarg segment).
template <bool BYTES>
__global__ __launch_bounds__(256) void byval_kernel(Columns c, const uint32_t* __restrict__ prog, uint32_t n_prog, uint32_t* __restrict__ out,
                                                     uint32_t rows) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < n_prog; ++i) {
    const uint32_t src = prog[i], col = src & 255u;
    const uint32_t lr = BYTES ? c.log_rows[col] : 0u;
    const uint32_t r = lr ? (row & ((1u << lr) - 1u)) : row;
    acc += c.p[col][r] + c.dyn[(src >> 8) % (16 * 9)] + c.n_static;
  }
  out[row] = acc;
}

// COMPILE-ONLY by default: the byte-table variant faulted on the shared pool once and is never launched again --
// `main` refuses it; its code stays in the object so that the ISA can be read (profiles/r05_kernarg_isa.txt, tests/test_isa.py).
// `kernarg_byval run-w` launches the variant WITHOUT the byte table (scalar loads only), which is known to run.
int main(int argc, char** argv) {
  if (argc < 2 || std::string(argv[1]) != "run-w") {
    printf("compile-only micro-benchmark: read the ISA (hipcc --save-temps); `run-w` launches the variant without the byte table\n");
    return 0;
  }
  const bool bytes = false;
  const uint32_t rows = 1u << 21, n_prog = 64;
  std::vector<uint32_t> prog(n_prog);
  for (uint32_t i = 0; i < n_prog; ++i) prog[i] = (i * 37u) & 0xffffu;
  uint32_t *d_prog, *d_out, *d_col;
  if (hipMalloc(&d_prog, n_prog * 4) != hipSuccess || hipMalloc(&d_out, rows * 4) != hipSuccess || hipMalloc(&d_col, rows * 4) != hipSuccess) return 2;
  hipMemcpy(d_prog, prog.data(), n_prog * 4, hipMemcpyHostToDevice);
  hipMemset(d_col, 1, rows * 4);
  Columns c{};
  for (int i = 0; i < 256; ++i) { c.p[i] = d_col; c.log_rows[i] = (i % 7 == 0) ? 3 : 0; }
  c.n_static = 5;
  printf("sizeof(Columns) = %zu, variant: %s\n", sizeof(Columns), bytes ? "with the byte table" : "without the byte table");
  fflush(stdout);
  if (bytes) hipLaunchKernelGGL(byval_kernel<true>, dim3(rows / 256), dim3(256), 0, 0, c, (const uint32_t*)d_prog, n_prog, d_out, rows);
  else hipLaunchKernelGGL(byval_kernel<false>, dim3(rows / 256), dim3(256), 0, 0, c, (const uint32_t*)d_prog, n_prog, d_out, rows);
  const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
  printf("launch: %s, sync: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  uint32_t v = 0;
  hipMemcpy(&v, d_out, 4, hipMemcpyDeviceToHost);
  printf("out[0] = %u\n", v);
  return e1 == hipSuccess && e2 == hipSuccess ? 0 : 1;
}
