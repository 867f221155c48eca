// How much independent VALU / LDS work fits behind one v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles) when there is a
// single wave per SIMD?  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_window mfma_window.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NL, int NACC, int WPS>
__global__ void __launch_bounds__(256, WPS) k(float* out, int iters) {
  __shared__ uint4 lds[1024];
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  uint4 av = make_uint4(threadIdx.x, 1, 2, 3), bv = make_uint4(4, 5, 6, threadIdx.x);
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
  lds[threadIdx.x] = av; lds[threadIdx.x + 256] = bv; lds[threadIdx.x + 512] = av; lds[threadIdx.x + 768] = bv;
  __syncthreads();
  uint4 l0 = av;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[u % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) f[v & 7] = __builtin_fmaf(f[v & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int l = 0; l < NL; ++l) { uint4 t = lds[(threadIdx.x + 64 * (u + l)) & 1023]; l0.x ^= t.x; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) s += acc[a][0];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + l0.x;
}

template <int NV, int NL, int NACC, int WPS = 1>
void run(float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NL, NACC, WPS>), dim3(256 * WPS), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NL, NACC, WPS>), dim3(256 * WPS), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e-3 / (iters * 8.0 * WPS);   // seconds per MFMA per SIMD
  printf("waves/SIMD %d  VALU/mfma %2d  ds_read_b128/mfma %d  accumulators %d : %.1f ns per MFMA  (%.1f cycles @2.4GHz)  %.0f TFLOP/s chip\n", WPS, NV, NL, NACC,
         per * 1e9, per * 2.4e9, 32768.0 * 1024 / per / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 2 * 256 * 256 * 4);
  run<0, 0, 4>(out); run<0, 0, 2>(out); run<0, 0, 1>(out);
  run<2, 0, 4>(out); run<4, 0, 4>(out); run<6, 0, 4>(out); run<8, 0, 4>(out); run<12, 0, 4>(out);
  run<0, 0, 4, 2>(out); run<4, 0, 4, 2>(out); run<8, 0, 4, 2>(out); run<12, 0, 4, 2>(out); run<16, 0, 4, 2>(out); run<24, 0, 4, 2>(out);
  return 0;
}
