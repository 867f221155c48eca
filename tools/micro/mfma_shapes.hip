// Power-limited rate of the two bf16 MFMA shapes on random data: v_mfma_f32_32x32x16_bf16 (C = 16 registers per lane, read + written per
// 32 Kflop) against v_mfma_f32_16x16x32_bf16 (C = 4 registers per lane per 16 Kflop: half the accumulator traffic per flop, twice the
// operand traffic).  Bare streams, one wave per SIMD, 8 independent accumulators.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const uint4* __restrict__ src, float* out, unsigned long long* ticks, int iters) {
  uint4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 1023]; b[i] = src[(threadIdx.x + 64 * i + 256) & 1023]; }
  f32x16 c32[4]; f32x4 c16[16];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c32[i][r] = 0.f;
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) c16[i][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 32) {
#pragma unroll
      for (int u = 0; u < 16; ++u)      // 16 x 32 Kflop
        c32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u & 3]), __builtin_bit_cast(bf16x8, b[(u >> 2) & 3]), c32[u & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 32; ++u)      // 32 x 16 Kflop
        c16[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[u & 3]), __builtin_bit_cast(bf16x8, b[(u >> 2) & 3]), c16[u & 15], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c32[i][0];
  for (int i = 0; i < 16; ++i) s += c16[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

static uint4* g_src; static float* g_out; static unsigned long long* g_t;
template <int SHAPE, int WPS> void run(const char* what) {
  const int iters = 40000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256 * WPS), dim3(256), 0, 0, g_src, g_out, g_t, 500);
  float best = 1e30f; unsigned long long tk = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256 * WPS), dim3(256), 0, 0, g_src, g_out, g_t, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; (void)hipMemcpy(&tk, g_t, 8, hipMemcpyDeviceToHost); }
  }
  const double flop = (double)iters * 16 * 32768.0 * 1024 * WPS;   // per launch: 1024 SIMD x WPS waves x 512 Kflop per iteration
  printf("%-28s waves/SIMD %d: %.3f ms  %.0f TFLOP/s issued  %.1f cycles per 32 Kflop per wave  clock %.2f GHz\n", what, WPS, best, flop / best / 1e9,
         (double)tk / (iters * 16.0), (double)tk / (best * 1e-3) / 1e9);
}
int main() {
  (void)hipMalloc(&g_src, 1024 * 16); (void)hipMalloc(&g_out, 2 * 256 * 256 * 4); (void)hipMalloc(&g_t, 8);
  unsigned h[4096]; unsigned x = 12345;
  for (auto& w : h) { x = x * 1664525u + 1013904223u; w = (x & 0x807f807fu) | 0x3f003f00u; }
  (void)hipMemcpy(g_src, h, sizeof(h), hipMemcpyHostToDevice);
  for (int r = 0; r < 2; ++r) {
    run<32, 1>("v_mfma_f32_32x32x16_bf16"); run<16, 1>("v_mfma_f32_16x16x32_bf16");
    run<32, 2>("v_mfma_f32_32x32x16_bf16"); run<16, 2>("v_mfma_f32_16x16x32_bf16");
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
