// How many instructions of which kind fit into the shadow of one v_mfma_f32_32x32x16_bf16 (32 cycles of matrix pipe) when they come
// from the SAME wave, at one and at two waves per SIMD?  (profiles/r03_mfma_valu_exclusion.md: from the PARTNER wave nothing fits.)
// Random bf16 operands (the clock the chip grants depends on the data).  Prints ns and shader cycles (s_memtime) per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o gap_probe gap_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { K_AND = 0, K_SUB = 1, K_CVT = 2, K_DOT2C = 3, K_CHAIN = 4, K_SPLIT = 5, K_DSREAD = 6, K_NONE = 7 };

template <int KIND, int N, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const uint4* __restrict__ src, float* out, unsigned long long* ticks, int iters) {
  __shared__ uint4 lds[1024];
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const uint4 av = src[threadIdx.x], bv = src[256 + threadIdx.x];
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = __uint_as_float(src[512 + threadIdx.x].x + i * 977u) * 1e-3f + i;
  unsigned c10, c01;
  asm volatile("s_mov_b32 %0, 0xbf80" : "=s"(c10));
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(c01));
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = src[i & 511];
  __syncthreads();
  uint4 l0 = av;
  const unsigned la = (threadIdx.x & 255) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[u & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (KIND == K_AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(f[n & 7]));
        if (KIND == K_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[n & 7]) : "v"(f[(n + 1) & 7]));
        if (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[n & 7]) : "v"(f[(n + 3) & 7]));
        if (KIND == K_DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[n & 7]) : "s"(c10), "v"(f[(n + 3) & 7]));
        if (KIND == K_CHAIN) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[1]));
        if (KIND == K_DSREAD) { uint4 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(la), "n"((n & 3) * 4096)); l0.x ^= t.x; }
      }
      if (KIND == K_SPLIT) {   // N pairs, the 11-instruction subtract form of csrc/mlp.hip (split3_pair_p), independent pairs
#pragma unroll
        for (int n = 0; n < N; ++n) {
          float &x0 = f[(2 * n) & 7], &x1 = f[(2 * n + 1) & 7];
          unsigned h, m, l, t0_, t1_;
          float r0, r1;
          asm volatile(
              "v_cvt_pk_bf16_f32 %2, %0, %1\n\t"
              "v_lshlrev_b32 %5, 16, %2\n\t"
              "v_and_b32 %6, 0xffff0000, %2\n\t"
              "v_sub_f32 %7, %0, %5\n\t"
              "v_sub_f32 %8, %1, %6\n\t"
              "v_cvt_pk_bf16_f32 %3, %7, %8\n\t"
              "v_lshlrev_b32 %5, 16, %3\n\t"
              "v_and_b32 %6, 0xffff0000, %3\n\t"
              "v_sub_f32 %7, %7, %5\n\t"
              "v_sub_f32 %8, %8, %6\n\t"
              "v_cvt_pk_bf16_f32 %4, %7, %8"
              : "+v"(x0), "+v"(x1), "=&v"(h), "=&v"(m), "=&v"(l), "=&v"(t0_), "=&v"(t1_), "=&v"(r0), "=&v"(r1));
          l0.y ^= h ^ m ^ l;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + l0.x + l0.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

static uint4* g_src; static float* g_out; static unsigned long long* g_ticks;
template <int KIND, int N, int WPS>
void run(const char* name) {
  const int iters = 12000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, N, WPS>), dim3(256 * WPS), dim3(256), 0, 0, g_src, g_out, g_ticks, 200);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, N, WPS>), dim3(256 * WPS), dim3(256), 0, 0, g_src, g_out, g_ticks, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t;
  (void)hipMemcpy(&t, g_ticks, 8, hipMemcpyDeviceToHost);
  const double per = ms * 1e-3 / (iters * 8.0 * WPS);   // seconds per MFMA per SIMD
  const int ninstr = KIND == K_SPLIT ? 11 * N : N;
  printf("%-7s x%2d (%2d instr / MFMA)  waves/SIMD %d : %6.2f ns / MFMA, %5.1f s_memtime ticks / MFMA of one wave (100 MHz ticks x clock ratio), %5.0f TFLOP/s issued\n",
         name, N, ninstr, WPS, per * 1e9, (double)t / (iters * 8.0), 32768.0 * 1024 / per / 1e12);
}

int main() {
  (void)hipMalloc(&g_src, 1024 * 16); (void)hipMalloc(&g_out, 2 * 256 * 256 * 4); (void)hipMalloc(&g_ticks, 8);
  unsigned h[4096];
  unsigned x = 12345;
  for (auto& w : h) { x = x * 1664525u + 1013904223u; w = (x & 0x807f807fu) | 0x3f003f00u; }   // bf16 pairs in +-[0.5, 1)
  (void)hipMemcpy(g_src, h, sizeof(h), hipMemcpyHostToDevice);
#define ROW(K, NAME) run<K, 1, 1>(NAME); run<K, 2, 1>(NAME); run<K, 3, 1>(NAME); run<K, 4, 1>(NAME); run<K, 5, 1>(NAME); run<K, 6, 1>(NAME); run<K, 8, 1>(NAME); \
                     run<K, 2, 2>(NAME); run<K, 4, 2>(NAME); run<K, 6, 2>(NAME);
  run<K_NONE, 0, 1>("none"); run<K_NONE, 0, 2>("none");
  ROW(K_AND, "and") ROW(K_SUB, "sub") ROW(K_CVT, "cvt_pk") ROW(K_DOT2C, "dot2c") ROW(K_CHAIN, "chain")
  run<K_SPLIT, 1, 1>("split"); run<K_SPLIT, 1, 2>("split");
  run<K_DSREAD, 1, 1>("ds_read"); run<K_DSREAD, 2, 1>("ds_read"); run<K_DSREAD, 1, 2>("ds_read");
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
