// L2 -> CU streaming rate for a read-only table that every workgroup walks in the SAME order (the packed weights of
// the MLP kernels), versus walks that are rotated per workgroup.  256-thread workgroups, 1 or 2 per CU, every wave
// issues 16-byte-per-lane loads (1 KiB per instruction) over a 2 MiB table.  Reports bytes per clock and CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE, int NG>
__global__ void __launch_bounds__(256, 2) k(long long* out, const uint4* __restrict__ wts, int iters, int rot) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned x = 0;
  const unsigned base = MODE == 0 ? 0u : (MODE == 1 ? blockIdx.x * rot : (blockIdx.x * 2654435761u) >> 8);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    uint4 g[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = wts[(base + ((it * NG + i) * 4 + w) * 64 + lane) & 0x1ffff];   // 2 MiB window, 4 KiB per WG and step
#pragma unroll
    for (int i = 0; i < NG; ++i) x ^= g[i].x;
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; }
  if (x == 0x12345) out[0] = x;
}

template <int MODE, int NG>
void run(long long* out, long long* h, const uint4* wts, int wpc, int rot, const char* what) {
  const int grid = 256 * wpc, iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, NG>), dim3(grid), dim3(256), 0, 0, out, wts, iters, rot);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost);
  double t = 0, w = 0;
  for (int b = 0; b < grid; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; }
  t /= grid; w /= grid;
  const double bytes_cu = (double)iters * NG * 4096 * wpc;
  printf("%-52s WG/CU %d: %5.1f B/clk/CU  (%.1f TB/s chip, sclk %.2f GHz)\n", what, wpc, bytes_cu / t, bytes_cu * 256 / (w * 1e-8) / 1e12, t / (w * 10));
}

// other access widths / paths over the same 2 MiB table: WIDTH 4, 8 (bytes per lane, plain loads), 16 = LDS-DMA (global_load_lds_dwordx4)
template <int WIDTH, int NG>
__global__ void __launch_bounds__(256, 2) kw(long long* out, const char* __restrict__ wts, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[4 * NG * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned x = 0;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (WIDTH == 16) {
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const char* g = wts + (((size_t)((it * NG + i) * 4 + w) * 1024 + lane * 16) & 0x1fffff);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds + (w * NG + i) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (WIDTH == 8) {
      uint2 g[NG];
#pragma unroll
      for (int i = 0; i < NG; ++i) g[i] = *reinterpret_cast<const uint2*>(wts + (((size_t)((it * NG + i) * 4 + w) * 512 + lane * 8) & 0x1fffff));
#pragma unroll
      for (int i = 0; i < NG; ++i) x ^= g[i].x;
    } else {
      unsigned g[NG];
#pragma unroll
      for (int i = 0; i < NG; ++i) g[i] = *reinterpret_cast<const unsigned*>(wts + (((size_t)((it * NG + i) * 4 + w) * 256 + lane * 4) & 0x1fffff));
#pragma unroll
      for (int i = 0; i < NG; ++i) x ^= g[i];
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; }
  if (x == 0x12345) out[0] = x + lds[lane];
}
template <int WIDTH, int NG>
void runw(long long* out, long long* h, const uint4* wts, int wpc, const char* what) {
  const int grid = 256 * wpc, iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kw<WIDTH, NG>), dim3(grid), dim3(256), 0, 0, out, (const char*)wts, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost);
  double t = 0, w = 0;
  for (int b = 0; b < grid; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; }
  t /= grid; w /= grid;
  const double bytes_cu = (double)iters * NG * 4 * 64 * WIDTH * wpc;
  printf("%-52s WG/CU %d: %5.1f B/clk/CU\n", what, wpc, bytes_cu / t);
}

// 512-thread workgroups (one per CU): waves w and w + 4 read the SAME 1 KiB pieces -- does the second reader hit L1?
template <int NG, int LAG>
__global__ void __launch_bounds__(512, 1) k8(long long* out, const uint4* __restrict__ wts, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned x = 0;
  const long long t0 = clock64(), w0 = wall_clock64();
  if (LAG && w >= 4) __builtin_amdgcn_s_sleep(LAG);
  for (int it = 0; it < iters; ++it) {
    uint4 g[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = wts[(((it * NG + i) * 4 + (w & 3)) * 64 + lane) & 0x1ffff];
#pragma unroll
    for (int i = 0; i < NG; ++i) x ^= g[i].x;
    if ((it & 15) == 15) __syncthreads();   // the layer barriers keep the two halves within a few k-steps
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; }
  if (x == 0x12345) out[0] = x;
}
template <int NG, int LAG>
void run8(long long* out, long long* h, const uint4* wts, const char* what) {
  const int grid = 256, iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k8<NG, LAG>), dim3(grid), dim3(512), 0, 0, out, wts, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost);
  double t = 0, w = 0;
  for (int b = 0; b < grid; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; }
  t /= grid; w /= grid;
  const double bytes_cu = (double)iters * NG * 4096 * 2;   // delivered to the waves (each unique byte twice)
  printf("%-52s 8 waves : %5.1f B/clk/CU delivered (%.1f unique)\n", what, bytes_cu / t, bytes_cu / t / 2);
}

int main() {
  long long *out, *h = (long long*)malloc(512 * 4 * 8);
  uint4* wts;
  (void)hipMalloc(&out, 512 * 4 * 8);
  (void)hipMalloc(&wts, 2 << 20);
  (void)hipMemset(wts, 1, 2 << 20);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    run<0, 8>(out, h, wts, wpc, 0, "same walk in every workgroup");
    run<1, 8>(out, h, wts, wpc, 256, "rotated by 4 KiB per workgroup");
    run<1, 8>(out, h, wts, wpc, 2048, "rotated by 32 KiB per workgroup");
    run<1, 8>(out, h, wts, wpc, 264, "rotated by 4 KiB + 128 B per workgroup");
    run<1, 8>(out, h, wts, wpc, 8, "rotated by 128 B per workgroup");
    run<2, 8>(out, h, wts, wpc, 0, "hashed start per workgroup");
  }
  for (int wpc = 1; wpc <= 2; ++wpc) {
    runw<4, 8>(out, h, wts, wpc, "global_load_dword (4 B per lane)");
    runw<8, 8>(out, h, wts, wpc, "global_load_dwordx2 (8 B per lane)");
    runw<16, 8>(out, h, wts, wpc, "global_load_lds_dwordx4 (LDS-DMA, 16 B per lane)");
  }
  run8<4, 0>(out, h, wts, "pairs of waves read the same pieces, 4 per step");
  run8<8, 0>(out, h, wts, "pairs of waves read the same pieces, 8 per step");
  run8<4, 20>(out, h, wts, "same, second half starts ~1300 clk later");
  return 0;
}
