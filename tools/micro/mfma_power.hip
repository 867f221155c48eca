// Does a saturated v_mfma_f32_32x32x16_bf16 stream hold its clock with REAL (random) operand data?  Persistent
// workgroups, one wave per SIMD, bare MFMA loop over 8 operand pairs held in registers: constant bits vs random bf16
// values.  sclk = s_memtime / s_memrealtime.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256, 2) k(long long* out, const uint4* __restrict__ data, int iters) {
  uint4 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = data[(i * 256 + threadIdx.x)]; b[i] = data[((8 + i) * 256 + threadIdx.x)]; }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q & 7]), __builtin_bit_cast(bf16x8, b[(q * 3) & 7]), acc[q & 3], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) s += acc[q][0];
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; out[blockIdx.x * 4 + 2] = (long long)s; }
}

int main() {
  const int iters = 40000;
  long long *out, *h = (long long*)malloc(512 * 4 * 8);
  uint4* data;
  uint4* hd = (uint4*)malloc(16 * 256 * 16);
  (void)hipMalloc(&out, 512 * 4 * 8);
  (void)hipMalloc(&data, 16 * 256 * 16);
  for (int grid = 256; grid <= 256; grid += 256)   // one workgroup per CU = one wave per SIMD: exact 32 clk / MFMA accounting
  for (int mode = 0; mode < 3; ++mode) {
    unsigned short* u = (unsigned short*)hd;
    srand(1);
    for (int i = 0; i < 16 * 256 * 8; ++i) {
      if (mode == 0) u[i] = 0x3f80;                                   // all ones (1.0)
      else if (mode == 1) { float f = ((rand() % 2001) - 1000) * 1e-3f; unsigned v; memcpy(&v, &f, 4); u[i] = v >> 16; }   // uniform(-1, 1) as bf16
      else u[i] = (unsigned short)(rand() & 0xbfff) ;                 // random bits, exponent kept finite
    }
    (void)hipMemcpy(data, hd, 16 * 256 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, data, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost);
    double t = 0, w = 0;
    for (int b = 0; b < grid; ++b) { t += h[b * 4]; w += h[b * 4 + 1]; }
    t /= grid; w /= grid;
    const double secs = w * 1e-8;
    printf("WG/CU %d  %-28s %6.2f ms  sclk %.3f GHz  %6.0f TFLOP/s  (%.1f clk per MFMA and SIMD)\n",
           grid / 256, mode == 0 ? "operands all 1.0" : mode == 1 ? "operands uniform(-1,1) bf16" : "operands random bits", secs * 1e3, t / secs * 1e-9,
           (double)iters * 16 * 4 * grid * 32768 / secs / 1e12, t / ((double)iters * 16 * (grid / 256)));
  }
  return 0;
}
