// randline_probe.hip -- what a scattered, dependent read costs on this part (design input, not product code).
//
// Every lane runs a chain of `chain` dependent reads at pseudo-random granule-aligned addresses inside a working set of W
// bytes and loads `bytes` bytes of each granule (16-byte pieces).  Reported: granules per second, and the same as
// 128-byte-line traffic.  Questions it answers for the rank kernels:
//   * does a working set that fits the 256 MiB Infinity Cache (or the 4 MiB L2s) serve scattered lines faster than HBM?
//   * does a 32- / 64-byte granule cost less than a 128-byte one (is the memory-side request always 128 bytes)?
//   * what do address-translation misses cost (W = 64 GB against 1 GB)?
//   * does loading the whole line (8 pieces) cost more than one piece of it?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/randline_probe tools/randline_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// load flavours: 0 plain global_load_dwordx4, 1 nontemporal (nt), 2 sc0 sc1 (system-scope), 3 sc1, 4 sc0 sc1 nt
template <int kFlavour>
__device__ __forceinline__ uint4 load16(const uint4* p) {
  if (kFlavour == 0) return *p;
  uint4 v;
  if (kFlavour == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (kFlavour == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (kFlavour == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int kFlavour>
__global__ __launch_bounds__(256) void probe_flavour(const uint8_t* __restrict__ base, const uint64_t ngran, const int gran_shift, const int chain,
                                                     const int64_t n, uint64_t* __restrict__ out) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  uint64_t s = mix(uint64_t(q) * 0x9e3779b97f4a7c15ULL + 12345);
  uint64_t acc = 0;
  for (int c = 0; c < chain; c++) {
    const uint64_t g = s % ngran;
    const uint4 w = load16<kFlavour>(reinterpret_cast<const uint4*>(base + (g << gran_shift)));
    const uint64_t v = w.x + w.y + w.z + w.w;
    acc += v;
    s = mix(s + v);
  }
  if (acc == 0x1234567) out[0] = acc;
}

template <int kPieces>
__global__ __launch_bounds__(256) void probe(const uint8_t* __restrict__ base, const uint64_t ngran, const int gran_shift, const int chain,
                                             const int64_t n, uint64_t* __restrict__ out, const int spread) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  uint64_t s = mix(uint64_t(q) * 0x9e3779b97f4a7c15ULL + 12345);
  uint64_t acc = 0;
  for (int c = 0; c < chain; c++) {
    uint64_t g = s % ngran;
    if (spread) g = (g & ~uint64_t(spread - 1)) | (uint64_t(q) & uint64_t(spread - 1));   // neighbouring lanes -> neighbouring granules
    const uint4* p = reinterpret_cast<const uint4*>(base + (g << gran_shift));
    uint64_t v = 0;
#pragma unroll
    for (int k = 0; k < kPieces; k++) {
      const uint4 w = p[k];
      v += w.x + w.y + w.z + w.w;
    }
    acc += v;
    s = mix(s + v);    // the next address depends on what was loaded
  }
  if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char** argv) {
  const int64_t n = 10'000'000;
  uint64_t* d_out;
  CK(hipMalloc(&d_out, 8));
  const double ws_gb[] = {0.002, 0.016, 0.064, 0.128, 0.192, 0.256, 0.384, 0.512, 1.0, 4.0, 16.0, 64.0, 200.0};
  size_t max_bytes = size_t(200.0 * (1ull << 30));
  if (argc > 1) max_bytes = size_t(atof(argv[1]) * (1ull << 30));
  uint8_t* buf = nullptr;
  while (hipMalloc(&buf, max_bytes) != hipSuccess) max_bytes /= 2;
  CK(hipMemset(buf, 0, max_bytes));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("# n=%lld lanes, buffer %.1f GiB\n", (long long)n, double(max_bytes) / (1ull << 30));
  printf("# W_GiB gran pieces chain spread ms Ggran/s TB/s_as_128B_lines\n");
  struct Cfg { int gran_shift, pieces, chain, spread; };
  const Cfg cfgs[] = {{7, 1, 4, 0}, {7, 8, 4, 0}, {7, 2, 4, 0}, {6, 1, 4, 0}, {6, 4, 4, 0}, {5, 1, 4, 0}, {5, 2, 4, 0}, {7, 1, 1, 0}, {7, 1, 8, 0}, {7, 1, 4, 4}, {8, 8, 4, 0}};
  if (argc > 2) {   // load flavours at 1 GiB: does any cache policy make the memory-side request smaller than 128 bytes?
    printf("# flavour gran chain ms Ggran/s   (0 plain, 1 nt, 2 sc0 sc1, 3 sc1, 4 sc0 sc1 nt)\n");
    const size_t W = size_t(1) << 30;
    for (int fl = 0; fl < 5; fl++)
      for (int gs : {5, 6, 7}) {
        const uint64_t ngran = W >> gs;
        const int blocks = int((n + 255) / 256), chain = 4;
        auto launch = [&]() {
          switch (fl) {
            case 0: hipLaunchKernelGGL(probe_flavour<0>, dim3(blocks), dim3(256), 0, 0, buf, ngran, gs, chain, n, d_out); break;
            case 1: hipLaunchKernelGGL(probe_flavour<1>, dim3(blocks), dim3(256), 0, 0, buf, ngran, gs, chain, n, d_out); break;
            case 2: hipLaunchKernelGGL(probe_flavour<2>, dim3(blocks), dim3(256), 0, 0, buf, ngran, gs, chain, n, d_out); break;
            case 3: hipLaunchKernelGGL(probe_flavour<3>, dim3(blocks), dim3(256), 0, 0, buf, ngran, gs, chain, n, d_out); break;
            default: hipLaunchKernelGGL(probe_flavour<4>, dim3(blocks), dim3(256), 0, 0, buf, ngran, gs, chain, n, d_out); break;
          }
        };
        launch();
        launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        printf("%d %4d %d %8.3f %8.2f\n", fl, 1 << gs, chain, ms, double(n) * chain / (ms * 1e-3) / 1e9);
        fflush(stdout);
      }
    return 0;
  }
  for (double w : ws_gb) {
    const size_t W = size_t(w * (1ull << 30));
    if (W > max_bytes) continue;
    for (const Cfg& c : cfgs) {
      const uint64_t ngran = W >> c.gran_shift;
      auto launch = [&]() {
        const int blocks = int((n + 255) / 256);
        switch (c.pieces) {
          case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, buf, ngran, c.gran_shift, c.chain, n, d_out, c.spread); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, buf, ngran, c.gran_shift, c.chain, n, d_out, c.spread); break;
          case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, buf, ngran, c.gran_shift, c.chain, n, d_out, c.spread); break;
          default: hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, buf, ngran, c.gran_shift, c.chain, n, d_out, c.spread); break;
        }
      };
      launch();
      launch();
      CK(hipEventRecord(e0));
      const int reps = 5;
      for (int r = 0; r < reps; r++) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      const double g = double(n) * c.chain / (ms * 1e-3) / 1e9;
      printf("%8.3f %4d %d %d %d %8.3f %8.2f %6.2f\n", w, 1 << c.gran_shift, c.pieces, c.chain, c.spread, ms, g, g * 128 / 1000);
      fflush(stdout);
    }
  }
  return 0;
}
