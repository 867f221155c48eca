// common.h -- shared helpers for the gfx950 kernels (error plumbing, launch geometry).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/fastnerf.h"

namespace fn {
void set_error(const char* fmt, ...);
inline hipStream_t S(fn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace fn

#define FN_CHECK_ARG(cond, msg)                          \
  do {                                                   \
    if (!(cond)) {                                       \
      fn::set_error("%s: bad argument: %s", __func__, msg); \
      return -1;                                         \
    }                                                    \
  } while (0)

#define FN_HIP(expr)                                                               \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      fn::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
      return -2;                                                                   \
    }                                                                              \
  } while (0)

#define FN_LAUNCH_CHECK() FN_HIP(hipGetLastError())

// exact (non-contracted) fp32 helpers: the reference computes mul and add as
// separate roundings (e.g. pts = o + d*z, render.py:268).
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }

// Philox4x32-10 counter RNG (used only when the caller injects no randoms).
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// U[0,1) with 24 random bits, like torch.rand for fp32.
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
