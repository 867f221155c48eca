// What drags the shader clock under a split-bf16 MLP-like instruction mix?  Persistent 256-thread workgroups (WPC per
// CU) run, per iteration, 12 v_mfma_f32_32x32x16_bf16 plus optional ds_read_b128 / L2-resident global_load_dwordx4 /
// VALU / streaming 16-byte stores, for tens of milliseconds; sclk = s_memtime ticks / wall-clock (100 MHz) ticks.
// Build: hipcc --offload-arch=gfx950 -O3 -o power_probe power_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NL, int NG, int NV, int NS>
__global__ void __launch_bounds__(256, 2) k(long long* out, const uint4* __restrict__ wts, uint4* __restrict__ sink, int iters) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(i, 1, 2, 3);
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  uint4 av = make_uint4(threadIdx.x, 1, 2, 3), bv = make_uint4(4, 5, 6, threadIdx.x);
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
  const long long t0 = clock64(), w0 = wall_clock64();
  unsigned x = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 g[NG > 0 ? NG : 1], l[NL > 0 ? NL : 1];
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = wts[(((it * 4 + w) * NG + i) * 64 + lane) & 0x1ffff];   // 2 MiB window
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = lds[(it * 64 + i * 64 + lane * 1) & 4095];
#pragma unroll
    for (int q = 0; q < NM; ++q) {
      acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[q & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / NM; ++v) f[(q + v) & 7] = __builtin_fmaf(f[(q + v) & 7], 1.0001f, 0.5f);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) x ^= g[i].x;
#pragma unroll
    for (int i = 0; i < NL; ++i) x ^= l[i].y;
    if (NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const u32x4 t = {x, (unsigned)it, 0u, 0u};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(sink + ((((size_t)blockIdx.x * iters + it) * NS + i) * 256 + threadIdx.x)));
      }
    }
    av.x ^= x & 1;
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  for (int i = 0; i < 8; ++i) s += f[i];
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; out[blockIdx.x * 4 + 2] = (long long)s + x; }
}

template <int NM, int NL, int NG, int NV, int NS>
void run(long long* out, long long* h, const uint4* wts, uint4* sink, int wpc, const char* what) {
  const int grid = 256 * wpc;
  int iters = NS ? 6000 : 40000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NM, NL, NG, NV, NS>), dim3(grid), dim3(256), 65536, 0, out, wts, sink, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost);
  double t = 0, w = 0;
  for (int b = 0; b < grid; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; }
  t /= grid; w /= grid;
  const double secs = w * 1e-8, sclk = t / secs * 1e-9;
  const double mf = (double)iters * NM * 4 * grid / secs;   // wave-MFMAs per second, chip
  printf("%-44s WG/CU %d: %6.2f ms  sclk %.3f GHz  %6.0f TFLOP/s (%.1f clk per MFMA and SIMD)\n", what, wpc, secs * 1e3, sclk,
         mf * 32768 / 1e12, t / ((double)iters * NM * wpc));
}

int main() {
  long long *out, *h = (long long*)malloc(512 * 4 * 8);
  uint4 *wts, *sink;
  (void)hipMalloc(&out, 512 * 4 * 8);
  (void)hipMalloc(&wts, 2 << 20);
  (void)hipMemset(wts, 1, 2 << 20);
  (void)hipMalloc(&sink, (size_t)512 * 6000 * 2 * 256 * 16);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<12, 0, 0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    run<12, 0, 0, 0, 0>(out, h, wts, sink, wpc, "12 MFMA");
    run<12, 8, 0, 0, 0>(out, h, wts, sink, wpc, "12 MFMA + 8 ds_read_b128");
    run<12, 0, 8, 0, 0>(out, h, wts, sink, wpc, "12 MFMA + 8 global_load_dwordx4 (L2)");
    run<12, 8, 8, 0, 0>(out, h, wts, sink, wpc, "12 MFMA + 8 ds_read + 8 L2 loads");
    run<12, 0, 0, 48, 0>(out, h, wts, sink, wpc, "12 MFMA + 48 v_fma_f32");
    run<12, 8, 8, 48, 0>(out, h, wts, sink, wpc, "12 MFMA + 8 ds_read + 8 L2 + 48 VALU");
    run<12, 8, 8, 48, 1>(out, h, wts, sink, wpc, "... + one 16-B/lane streaming store");
    run<12, 8, 8, 84, 1>(out, h, wts, sink, wpc, "12 MFMA + 8 + 8 + 84 VALU + 1 store");
  }
  return 0;
}
