// train.hip -- MSE loss + its gradient + the per-(image, leaf) error table that feeds the
// quadtree, and the Adam update over the flat parameter buffer (gfx950).
//
// Reference semantics (nerf-ours/): img2mse run_nerf_helpers.py:9; loss = mse(fine)+mse(coarse)
// run_nerf.py:482-490; epoch loss map run_nerf.py:505-506 + tree.py:538,632-642 (max over the
// rays of a leaf and the 3 channels of |gt-pred|) -- reduced here on the device with an
// atomicMax on the uint view of non-negative floats (exact and order independent);
// torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) run_nerf.py:99,494.
#include "common.h"

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = (threadIdx.x < 4) ? red[threadIdx.x] : 0.f;
  if (w == 0) {
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
  }
  __syncthreads();
  return t;  // valid in thread 0
}

__global__ void __launch_bounds__(256) mse_leafmax_kernel(int64_t n, const float* __restrict__ rgb,
                                                           const float* __restrict__ rgb0,
                                                           const float* __restrict__ target, float gscale,
                                                           float inv_count, float* __restrict__ g_rgb,
                                                           float* __restrict__ g_rgb0, float* __restrict__ loss2,
                                                           const int32_t* __restrict__ tag, int max_leaves,
                                                           uint32_t* __restrict__ table) {
  __shared__ float red[4];
  float se = 0.f, se0 = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float emax = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = target[i * 3 + c];
      const float d = fsub(rgb[i * 3 + c], t);
      se += d * d;
      if (g_rgb) g_rgb[i * 3 + c] = gscale * d;
      emax = fmaxf(emax, fabsf(fsub(t, rgb[i * 3 + c])));
      if (rgb0) {
        const float d0 = fsub(rgb0[i * 3 + c], t);
        se0 += d0 * d0;
        if (g_rgb0) g_rgb0[i * 3 + c] = gscale * d0;
      }
    }
    if (table && tag) {
      const int64_t slot = (int64_t)tag[i * 2] * max_leaves + tag[i * 2 + 1];
      atomicMax(table + slot, __float_as_uint(emax));
    }
  }
  const float s = block_sum_256(se, red);
  const float s0 = block_sum_256(se0, red);
  if (threadIdx.x == 0 && loss2) {
    atomicAdd(loss2 + 0, s * inv_count);
    atomicAdd(loss2 + 1, s0 * inv_count);
  }
}

__global__ void __launch_bounds__(256) adam_kernel(int64_t n4, int64_t n, float* __restrict__ p,
                                                    const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float b1, float b2, float omb1,
                                                    float omb2, float eps, float step_size, float bc2_sqrt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      M[k] = fadd(fmul(M[k], b1), fmul(omb1, G[k]));
      V[k] = fadd(fmul(V[k], b2), fmul(omb2, fmul(G[k], G[k])));
      const float denom = fadd(sqrtf(V[k]) / bc2_sqrt, eps);
      P[k] = fsub(P[k], fmul(step_size, M[k] / denom));
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  const int64_t t = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && t < n) {
    const float gk = g[t];
    const float mk = fadd(fmul(m[t], b1), fmul(omb1, gk));
    const float vk = fadd(fmul(v[t], b2), fmul(omb2, fmul(gk, gk)));
    const float denom = fadd(sqrtf(vk) / bc2_sqrt, eps);
    p[t] = fsub(p[t], fmul(step_size, mk / denom));
    m[t] = mk;
    v[t] = vk;
  }
}

extern "C" int fastnerf_mse_leafmax(int64_t n, const float* rgb, const float* rgb0, const float* target,
                                    float grad_scale, float* g_rgb, float* g_rgb0, float* loss2,
                                    const int32_t* leaf_tag, int max_leaves, uint32_t* table, fn_stream_t stream) {
  FN_CHECK_ARG(n > 0 && rgb && target, "n>0 and non-null rgb/target");
  FN_CHECK_ARG(!(table && !leaf_tag) && (!table || max_leaves > 0), "table needs leaf_tag and max_leaves>0");
  if (loss2) FN_HIP(hipMemsetAsync(loss2, 0, 2 * sizeof(float), fn::S(stream)));
  // up to 64 k rays (every training batch) one workgroup does the whole reduction, so the reported losses are
  // bit-reproducible like the parameters; beyond that the block sums meet in fp32 atomics (last-bit order dependent)
  int64_t g = (n <= 65536) ? 1 : (n + 255) / 256;
  if (g > 1024) g = 1024;
  const float inv_count = (float)(1.0 / (3.0 * (double)n));
  const float gscale = (float)(2.0 / (3.0 * (double)n) * (double)grad_scale);
  hipLaunchKernelGGL(mse_leafmax_kernel, dim3((int)g), dim3(256), 0, fn::S(stream), n, rgb, rgb0, target, gscale,
                     inv_count, g_rgb, g_rgb0, loss2, leaf_tag, max_leaves, table);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_adam_step(int64_t n, float* params, const float* grads, float* m, float* v, double lr,
                                  double beta1, double beta2, double eps, int step, fn_stream_t stream) {
  FN_CHECK_ARG(n > 0 && params && grads && m && v && step >= 1, "n>0, step>=1, non-null pointers");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int64_t n4 = n / 4;
  int64_t g = (n4 + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((int)g), dim3(256), 0, fn::S(stream), n4, n, params, grads, m, v, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt);
  FN_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------------------
// nerf++ quadtree fork: split criterion = MEAN of |gt - pred| over a leaf's rays and channels
// (nerf++-ours/tree.py:622).  Accumulated per (image, leaf) as an fp64 sum + a count so that the
// result does not depend on the order rays arrive in (fp32 atomics would).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) leaf_sumcount_kernel(int64_t n, const float* __restrict__ rgb,
                                                             const float* __restrict__ target,
                                                             const int32_t* __restrict__ tag, int max_leaves,
                                                             double* __restrict__ sum, int32_t* __restrict__ count) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double e = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) e += (double)fabsf(fsub(target[i * 3 + c], rgb[i * 3 + c]));
    const int64_t slot = (int64_t)tag[i * 2] * max_leaves + tag[i * 2 + 1];
    atomicAdd(sum + slot, e);
    atomicAdd(count + slot, 1);
  }
}

extern "C" int fastnerf_leaf_sumcount(int64_t n, const float* rgb, const float* target, const int32_t* leaf_tag,
                                      int max_leaves, double* sum, int32_t* count, fn_stream_t stream) {
  FN_CHECK_ARG(n > 0 && rgb && target && leaf_tag && sum && count && max_leaves > 0, "n>0, non-null pointers");
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(leaf_sumcount_kernel, dim3((int)g), dim3(256), 0, fn::S(stream), n, rgb, target, leaf_tag,
                     max_leaves, sum, count);
  FN_LAUNCH_CHECK();
  return 0;
}
