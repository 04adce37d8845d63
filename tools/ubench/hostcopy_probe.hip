// hostcopy_probe.hip -- what hipMemcpy on a caller's pageable memory costs, and what it leaves behind (csrc/xfer.hip's policy rests on this).
//
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/hostcopy_probe tools/ubench/hostcopy_probe.hip && /tmp/hostcopy_probe
//
// For 8 MiB and 64 MiB (a k = 18 column / an extended-domain array), on anonymous mappings whose pages have all been touched (no
// page fault is in any figure):
//   A  hipHostRegister + hipHostUnregister of a fresh mapping                      -> what pinning a range costs per 4 KiB page
//   B  hipMemcpy H2D from a fresh mapping: first, second, third copy               -> is the first copy the pin (per-page cost), are later ones free of it
//   C  munmap of that mapping, then the NEXT GPU operation (an empty kernel + sync, nothing to do with the range): its latency
//      against the same operation after unmapping a range HIP never saw (E)         -> does unmapping a range the runtime has pinned stall the queues
//   D  mmap again (same size; whether the address comes back is printed), touch, hipMemcpy first / second
//   F  the same cycle (map, touch, copy, unmap) ten times: per-cycle copy time and the empty kernel after the unmap   -> does it grow
//   G  D2H into a fresh mapping: first, second
// Prints one table; exits 0.  Touches nothing of the library.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      std::exit(1);                                                                           \
    }                                                                                         \
  } while (0)

__global__ void empty_kernel(int* p) {
  if (p && threadIdx.x == 1024) *p = 1;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void* map_touched(size_t bytes, double* t_map_us = nullptr) {
  const double t0 = now_us();
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (t_map_us) *t_map_us = now_us() - t0;
  if (p == MAP_FAILED) {
    std::perror("mmap");
    std::exit(1);
  }
  std::memset(p, 0x5a, bytes);
  return p;
}

static double timed_h2d(void* d, const void* h, size_t bytes) {
  const double t0 = now_us();
  CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
  return now_us() - t0;
}
static double timed_d2h(void* h, const void* d, size_t bytes) {
  const double t0 = now_us();
  CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
  return now_us() - t0;
}
static double timed_empty_kernel() {
  const double t0 = now_us();
  hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, (int*)nullptr);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  return now_us() - t0;
}
static double timed_unmap(void* p, size_t bytes) {
  const double t0 = now_us();
  munmap(p, bytes);
  return now_us() - t0;
}

int main() {
  CK(hipSetDevice(0));
  void* d = nullptr;
  CK(hipMalloc(&d, (size_t)64 << 20));
  for (int i = 0; i < 3; ++i) timed_empty_kernel();
  double base = 1e30;
  for (int i = 0; i < 10; ++i) {
    const double t = timed_empty_kernel();
    base = t < base ? t : base;
  }
  std::printf("empty kernel + hipDeviceSynchronize, idle process: %.1f us (best of 10)\n", base);
  const char* env = std::getenv("GPU_PINNED_MIN_XFER_SIZE");
  std::printf("GPU_PINNED_MIN_XFER_SIZE=%s GPU_PINNED_XFER_SIZE=%s\n", env ? env : "(unset)",
              std::getenv("GPU_PINNED_XFER_SIZE") ? std::getenv("GPU_PINNED_XFER_SIZE") : "(unset)");
  for (size_t mib : {(size_t)8, (size_t)64}) {
    const size_t bytes = mib << 20, pages = bytes / 4096;
    std::printf("\n== %zu MiB (%zu pages of 4 KiB)\n", mib, pages);
    {  // A
      void* p = map_touched(bytes);
      double t0 = now_us();
      CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
      const double t_reg = now_us() - t0;
      const double c1 = timed_h2d(d, p, bytes), c2 = timed_h2d(d, p, bytes);
      t0 = now_us();
      CK(hipHostUnregister(p));
      const double t_unreg = now_us() - t0;
      const double t_um = timed_unmap(p, bytes), k = timed_empty_kernel();
      std::printf("A  hipHostRegister %9.1f us = %6.3f us/page   copies from the registered range %8.1f %8.1f us (%.1f GB/s)   "
                  "hipHostUnregister %8.1f us   munmap %7.1f us   next empty kernel %7.1f us\n",
                  t_reg, t_reg / pages, c1, c2, bytes / c2 / 1e3, t_unreg, t_um, k);
    }
    void* first_addr = nullptr;
    {  // B, C
      void* p = map_touched(bytes);
      first_addr = p;
      const double c1 = timed_h2d(d, p, bytes), c2 = timed_h2d(d, p, bytes), c3 = timed_h2d(d, p, bytes);
      std::printf("B  hipMemcpy H2D from a fresh touched mapping: first %9.1f us = %6.3f us/page (%.2f GB/s)   second %8.1f us (%.1f GB/s)   third %8.1f us\n",
                  c1, c1 / pages, bytes / c1 / 1e3, c2, bytes / c2 / 1e3, c3);
      const double t_um = timed_unmap(p, bytes);
      const double k1 = timed_empty_kernel(), k2 = timed_empty_kernel(), k3 = timed_empty_kernel();
      std::printf("C  munmap of the range hipMemcpy read from: %8.1f us   then empty kernels: %8.1f %8.1f %8.1f us\n", t_um, k1, k2, k3);
    }
    {  // E (control)
      void* p = map_touched(bytes);
      const double t_um = timed_unmap(p, bytes);
      const double k1 = timed_empty_kernel(), k2 = timed_empty_kernel();
      std::printf("E  munmap of a range HIP never saw:          %8.1f us   then empty kernels: %8.1f %8.1f us\n", t_um, k1, k2);
    }
    {  // D
      double t_map = 0;
      void* p = map_touched(bytes, &t_map);
      const double c1 = timed_h2d(d, p, bytes), c2 = timed_h2d(d, p, bytes);
      std::printf("D  mapped again (mmap %.1f us, %s address): first %9.1f us = %6.3f us/page   second %8.1f us\n", t_map,
                  p == first_addr ? "the SAME" : "another", c1, c1 / pages, c2);
      timed_unmap(p, bytes);
      timed_empty_kernel();
    }
    std::printf("F  ten cycles of map + touch + ONE copy + unmap + empty kernel [copy us / munmap us / kernel us]:\n   ");
    for (int i = 0; i < 10; ++i) {
      void* p = map_touched(bytes);
      const double c = timed_h2d(d, p, bytes);
      const double u = timed_unmap(p, bytes);
      const double k = timed_empty_kernel();
      std::printf(" [%.0f/%.0f/%.0f]", c, u, k);
    }
    std::printf("\n");
    {  // G
      void* p = map_touched(bytes);
      const double c1 = timed_d2h(p, d, bytes), c2 = timed_d2h(p, d, bytes);
      const double u = timed_unmap(p, bytes), k = timed_empty_kernel();
      std::printf("G  hipMemcpy D2H into a fresh touched mapping: first %9.1f us = %6.3f us/page   second %8.1f us (%.1f GB/s)   munmap %7.1f us   next empty kernel %7.1f us\n",
                  c1, c1 / pages, c2, bytes / c2 / 1e3, u, k);
    }
    {  // H: pinned staging of our own for comparison (one 2 MiB pinned slot, memcpy + DMA, serial: a floor for a staged copy, not the lanes)
      void* pin = nullptr;
      CK(hipHostMalloc(&pin, (size_t)2 << 20, hipHostMallocDefault));
      void* p = map_touched(bytes);
      const double t0 = now_us();
      for (size_t off = 0; off < bytes; off += (size_t)2 << 20) {
        std::memcpy(pin, (char*)p + off, (size_t)2 << 20);
        CK(hipMemcpy((char*)d + off, pin, (size_t)2 << 20, hipMemcpyHostToDevice));
      }
      const double t = now_us() - t0;
      std::printf("H  staged through one 2 MiB pinned slot, serial: %9.1f us (%.1f GB/s)\n", t, bytes / t / 1e3);
      timed_unmap(p, bytes);
      CK(hipHostFree(pin));
    }
  }
  CK(hipFree(d));
  return 0;
}
