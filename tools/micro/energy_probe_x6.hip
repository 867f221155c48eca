// Which activity costs the clock?  A saturated-ish v_mfma_f32_32x32x16_bf16 stream on RANDOM bf16 operands (one
// 256-thread workgroup per CU = one wave per SIMD, or two) plus, per 12 MFMAs, NL ds_read_b128 of random data, NG
// L2-resident 16-byte loads, NV fp32 VALU instructions, NS 16-byte streaming stores.  Reports shader clock
// (s_memtime / s_memrealtime), MFMA rate and time per iteration: the power management trades clock for activity.
// Build: hipcc --offload-arch=gfx950 -O3 -o energy_probe energy_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NL, int NG, int NV, int NS, int ND = 0>
__global__ void __launch_bounds__(256, 2) k(long long* out, const uint4* __restrict__ data, uint4* __restrict__ sink, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = data[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = data[2048 + i * 256 + threadIdx.x]; b[i] = data[4096 + i * 256 + threadIdx.x]; }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = __uint_as_float((data[threadIdx.x + i * 256].x & 0x007fffffu) | 0x3f800000u);
  unsigned x = 0;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    uint4 l[NL > 0 ? NL : 1], g[NG > 0 ? NG : 1];
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = lds[(it * 64 + i * 256 + threadIdx.x) & 2047];
#pragma unroll
    for (int i = 0; i < ND; ++i)   // LDS-DMA of 1 KiB per wave instruction from the L2-resident table into a private ring slot
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(data + ((((it * 4 + w) * (ND > 0 ? ND : 1) + i) * 64 + lane) & 0x1ffff)),
                                       (__attribute__((address_space(3))) void*)((char*)lds + 32768 + ((w * 4 + ((it * ND + i) & 3)) * 1024)), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = data[(((it * 4 + w) * (NG > 0 ? NG : 1) + i) * 64 + lane) & 0x1ffff];
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q & 3]), __builtin_bit_cast(bf16x8, b[(q * 3) & 3]), acc[q & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / 12; ++v) f[(q + v) & 7] = __builtin_fmaf(f[(q + v) & 7], 1.0001f, 0.37f);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) { a[i & 3].x ^= l[i].x & 0x00010001u; }   // consume: keeps operands random-ish
#pragma unroll
    for (int i = 0; i < NG; ++i) { b[i & 3].y ^= g[i].y & 0x00010001u; }
    if (NS) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const u32x4 t = {a[0].x, b[0].y, (unsigned)it, x};
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(sink + ((((size_t)blockIdx.x * iters + it) * NS + i) * 256 + threadIdx.x)));
      }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) s += acc[q][0];
  for (int i = 0; i < 8; ++i) s += f[i];
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; out[blockIdx.x * 4 + 2] = (long long)s + x; }
}

template <int NL, int NG, int NV, int NS, int ND = 0>
void run(long long* out, long long* h, const uint4* data, uint4* sink, int wpc, const char* what) {
  const int grid = 256 * wpc, iters = NS ? 8000 : 30000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NL, NG, NV, NS, ND>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NL, NG, NV, NS, ND>), dim3(grid), dim3(256), 65536, 0, out, data, sink, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost);
  double t = 0, w = 0;
  for (int b = 0; b < grid; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; }
  t /= grid; w /= grid;
  const double secs = w * 1e-8;
  printf("%-52s WG/CU %d: sclk %.3f GHz  %5.0f clk and %6.1f ns per iteration (12 MFMA / wave)  %5.0f TFLOP/s\n", what, wpc, t / secs * 1e-9,
         t / iters, secs / iters * 1e9, (double)iters * 12 * 4 * grid * 32768 / secs / 1e12);
}

int main() {
  long long *out, *h = (long long*)malloc(512 * 4 * 8);
  uint4 *data, *sink;
  const size_t nd = 2 << 20;
  unsigned short* hd = (unsigned short*)malloc(nd);
  srand(1);
  for (size_t i = 0; i < nd / 2; ++i) { float f = ((rand() % 2001) - 1000) * 1e-3f; unsigned v; memcpy(&v, &f, 4); hd[i] = v >> 16; }
  (void)hipMalloc(&out, 512 * 4 * 8);
  (void)hipMalloc(&data, nd);
  (void)hipMemcpy(data, hd, nd, hipMemcpyHostToDevice);
  (void)hipMalloc(&sink, (size_t)512 * 8000 * 2 * 256 * 16);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    run<0, 0, 0, 0>(out, h, data, sink, wpc, "12 MFMA (random bf16 operands)");
    run<4, 4, 36, 0>(out, h, data, sink, wpc, "x3 inference-like: 4 ds_read + 4 L2 + 36 VALU");
    run<4, 4, 72, 1>(out, h, data, sink, wpc, "x3 training-like: 4 ds_read + 4 L2 + 72 VALU + 1 store");
    // six-product split (per 12 MFMAs = half a k-step of a 64 x 64 wave tile)
    run<3, 3, 0, 0>(out, h, data, sink, wpc, "x6 planes in LDS: 3 ds_read + 3 L2");
    run<3, 3, 24, 0>(out, h, data, sink, wpc, "x6 planes in LDS: 3 ds_read + 3 L2 + 24 VALU (inference-like)");
    run<3, 3, 48, 1>(out, h, data, sink, wpc, "x6 planes in LDS: 3 ds_read + 3 L2 + 48 VALU + 1 store (training-like)");
    run<2, 3, 48, 0>(out, h, data, sink, wpc, "x6 split at read: 2 ds_read + 3 L2 + 48 VALU");
    run<2, 3, 60, 0>(out, h, data, sink, wpc, "x6 split at read: 2 ds_read + 3 L2 + 60 VALU (inference-like)");
    run<2, 3, 72, 1>(out, h, data, sink, wpc, "x6 split at read: 2 ds_read + 3 L2 + 72 VALU + 1 store (training-like)");
    run<6, 3, 24, 0>(out, h, data, sink, wpc, "x6 planes in LDS, 64 x 32 wave tile: 6 ds_read + 3 L2 + 24 VALU");
    run<3, 0, 24, 0, 3>(out, h, data, sink, wpc, "x6 planes in LDS, weights via LDS-DMA: 3(+3) ds_read + 3 DMA + 24 VALU");
  }
  return 0;
}
