// Streaming-read ceiling of the box: how fast can HBM be read with N workgroups per CU and U 16-byte loads in flight
// per lane?  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int U>
__global__ void __launch_bounds__(256) rd(const uint4* __restrict__ src, size_t n_u4, unsigned* out) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i + (U - 1) * stride < n_u4; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// contiguous-chunk variant: every workgroup streams its own contiguous range (like the dW kernels)
template <int U>
__global__ void __launch_bounds__(256) rd_chunk(const uint4* __restrict__ src, size_t n_u4, unsigned* out) {
  const size_t per = n_u4 / gridDim.x;
  const uint4* p = src + (size_t)blockIdx.x * per + threadIdx.x;
  unsigned acc = 0;
  for (size_t i = 0; i + U * 256 <= per; i += U * 256) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// LDS-DMA variant: the same contiguous chunks copied global -> LDS with global_load_lds_dwordx4 (1 KiB per wave
// instruction), U pieces in flight per wave, nothing consumed
template <int U>
__global__ void __launch_bounds__(256) rd_dma(const uint4* __restrict__ src, size_t n_u4, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const size_t per = n_u4 / gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint4* p = src + (size_t)blockIdx.x * per + wave * 64 + lane;
  for (size_t i = 0; i + U * 256 <= per; i += U * 256) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i + u * 256),
                                       (__attribute__((address_space(3))) void*)(lds + ((u * 4 + wave) * 1024)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(U / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 0x7b && threadIdx.x == 999) out[0] = 1;
}
template <int U>
void run_dma(const uint4* src, size_t n_u4, unsigned* out, int wgs) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rd_dma<U>), hipFuncAttributeMaxDynamicSharedMemorySize, U * 4096);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((rd_dma<U>), dim3(wgs), dim3(256), U * 4096, 0, src, n_u4, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("lds-dma U=%2d WGs=%5d : %.2f TB/s\n", U, wgs, 3.0 * n_u4 * 16 / (ms * 1e-3) / 1e12);
}
template <int U, bool CHUNK>
void run(const uint4* src, size_t n_u4, unsigned* out, int wgs) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    for (int k = 0; k < 3; ++k) {
      if (CHUNK) hipLaunchKernelGGL((rd_chunk<U>), dim3(wgs), dim3(256), 0, 0, src, n_u4, out);
      else hipLaunchKernelGGL((rd<U>), dim3(wgs), dim3(256), 0, 0, src, n_u4, out);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s U=%2d WGs=%5d : %.2f TB/s\n", CHUNK ? "chunk " : "stride", U, wgs, 3.0 * n_u4 * 16 / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t bytes = (size_t)8 << 30;
  uint4* src; unsigned* out;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 4);
  (void)hipMemset(src, 1, bytes);
  const size_t n = bytes / 16;
  for (int wgs : {256, 512, 1024, 2048, 8192}) { run<4, false>(src, n, out, wgs); run<8, false>(src, n, out, wgs); }
  for (int wgs : {256, 512, 1024}) { run<8, true>(src, n, out, wgs); run<16, true>(src, n, out, wgs); }
  for (int wgs : {256, 512}) { run_dma<8>(src, n, out, wgs); run_dma<16>(src, n, out, wgs); run_dma<32>(src, n, out, wgs); }
  return 0;
}
