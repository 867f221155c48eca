// train.hip -- MSE loss + its gradient + the per-(image, leaf) error table that feeds the
// quadtree, and the Adam update over the flat parameter buffer (gfx950).
//
// Reference semantics (nerf-ours/): img2mse run_nerf_helpers.py:9; loss = mse(fine)+mse(coarse)
// run_nerf.py:482-490; epoch loss map run_nerf.py:505-506 + tree.py:538,632-642 (max over the
// rays of a leaf and the 3 channels of |gt-pred|) -- reduced here on the device with an
// atomicMax on the uint view of non-negative floats (exact and order independent);
// torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) run_nerf.py:99,494.
#include "common.h"

#define MSE_THREADS 1024   // 16 waves: a training batch (<= 64 k rays) is reduced by ONE workgroup (bit-reproducible loss) in n / 1024 rounds
__device__ __forceinline__ float block_sum(float v, float* red) {      // fixed order: wave butterflies, then the 16 wave sums pairwise
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = (threadIdx.x < MSE_THREADS / 64) ? red[threadIdx.x] : 0.f;
  if (w == 0) {
#pragma unroll
    for (int o = 1; o < MSE_THREADS / 64; o <<= 1) t += __shfl_xor(t, o, 64);
  }
  __syncthreads();
  return t;  // valid in thread 0
}

__global__ void __launch_bounds__(MSE_THREADS) mse_leafmax_kernel(int64_t n, const float* __restrict__ rgb,
                                                           const float* __restrict__ rgb0,
                                                           const float* __restrict__ target, float gscale,
                                                           float inv_count, float* __restrict__ g_rgb,
                                                           float* __restrict__ g_rgb0, float* __restrict__ loss2,
                                                           const int32_t* __restrict__ tag, int max_leaves,
                                                           uint32_t* __restrict__ table) {
  __shared__ float red[MSE_THREADS / 64];
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
  const float s = block_sum(se, red);
  const float s0 = block_sum(se0, red);
  if (threadIdx.x == 0 && loss2) {
    if (gridDim.x == 1) {      // the whole batch in this workgroup: plain stores, no memset before the launch
      loss2[0] = s * inv_count;
      loss2[1] = s0 * inv_count;
    } else {
      atomicAdd(loss2 + 0, s * inv_count);
      atomicAdd(loss2 + 1, s0 * inv_count);
    }
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
  // up to 64 k rays (every training batch) one workgroup does the whole reduction, so the reported losses are
  // bit-reproducible like the parameters; beyond that the block sums meet in fp32 atomics (last-bit order dependent)
  int64_t g = (n <= 65536) ? 1 : (n + MSE_THREADS - 1) / MSE_THREADS;
  if (g > 1024) g = 1024;
  if (loss2 && g > 1) FN_HIP(hipMemsetAsync(loss2, 0, 2 * sizeof(float), fn::S(stream)));
  const float inv_count = (float)(1.0 / (3.0 * (double)n));
  const float gscale = (float)(2.0 / (3.0 * (double)n) * (double)grad_scale);
  hipLaunchKernelGGL(mse_leafmax_kernel, dim3((int)g), dim3(MSE_THREADS), 0, fn::S(stream), n, rgb, rgb0, target, gscale,
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
// (nerf++-ours/tree.py:622).  Accumulated per (image, leaf) as an fp64 sum + a count.  A ray's term (three |differences| added
// in fp64: exact) is rounded ONCE to a multiple of 2^-30 before it is added: sums of such terms below 2^23 are exact in fp64, hence
// independent of the order the atomics arrive in AND of how the rays are sharded over ranks -- the all-reduced (SUM) tables of N
// ranks are bit-identical to the single-rank tables (fastnerf_allreduce_leaf_sumcount, csrc/comm.cpp).  The rounding moves a
// leaf's mean by <= 4.7e-10, three orders below fp32's own resolution of the reference's torch.mean at the 1e-3 thresholds.
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
    e = rint(e * 1073741824.0) * (1.0 / 1073741824.0);
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


// ---------------------------------------------------------------------------------------
// Exact zero-gradient point compaction.  A sample whose upstream gradient d(loss)/d(raw) is exactly zero in all four
// components (sigma + noise <= 0  =>  alpha = 0, weight = 0, relu' = 0: 40-60 % of the samples of a batch, at
// initialisation and on trained scenes alike) contributes exactly nothing to any parameter gradient of
// loss.backward() (run_nerf.py:493): every pre-activation gradient of the point is a sum of products with those
// zeros.  The training backward therefore runs on the list of the other ("live") points only.
//   live_idx[0 .. count) = indices p (ascending: the compaction is stable, so results do not depend on timing) of the
//   points with draw[p] != 0;  count_out[0] = count, count_out[1] = n_points (for the caller's statistics).
// Three small launches: per-block counts -> exclusive scan of the block counts -> scatter.
// ---------------------------------------------------------------------------------------
#define CP_PTS 1024   // points per block (256 threads x 4)
__device__ __forceinline__ unsigned cp_flags(const float* __restrict__ draw, int64_t p0, int64_t n) {
  unsigned f = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t p = p0 + k;
    if (p < n) {
      const uint4 v = *reinterpret_cast<const uint4*>(draw + p * 4);
      if (((v.x | v.y | v.z | v.w) & 0x7fffffffu) != 0u) f |= 1u << k;   // +-0 in all four components = dead
    }
  }
  return f;
}
__global__ void __launch_bounds__(256) cp_count_kernel(int64_t n, const float* __restrict__ draw, int* __restrict__ blk) {
  __shared__ int red[4];
  const unsigned f = cp_flags(draw, (int64_t)blockIdx.x * CP_PTS + threadIdx.x * 4, n);
  int c = __popc(f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// one workgroup: exclusive scan of nb block counts (in place), total -> count_out
__global__ void __launch_bounds__(1024) cp_scan_kernel(int nb, int* __restrict__ blk, int* __restrict__ count_out, int n_points) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = (i < nb) ? blk[i] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(x, o, 64); if (lane >= o) x += t; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int pre = carry_s;
    for (int k = 0; k < w; ++k) pre += wsum[k];
    if (i < nb) blk[i] = pre + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = pre + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) { count_out[0] = carry_s; count_out[1] = n_points; }
}
__global__ void __launch_bounds__(256) cp_scatter_kernel(int64_t n, const float* __restrict__ draw, const int* __restrict__ blk,
                                                          int* __restrict__ live_idx) {
  __shared__ int wsum[4];
  const int64_t p0 = (int64_t)blockIdx.x * CP_PTS + threadIdx.x * 4;
  const unsigned f = cp_flags(draw, p0, n);
  const int c = __popc(f);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(x, o, 64); if (lane >= o) x += t; }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  int pos = blk[blockIdx.x] + x - c;
  for (int k = 0; k < w; ++k) pos += wsum[k];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (f & (1u << k)) live_idx[pos++] = (int)(p0 + k);
}

extern "C" int64_t fastnerf_compact_ws_ints(int64_t n_points) { return (n_points + CP_PTS - 1) / CP_PTS; }

extern "C" int fastnerf_compact_live(int64_t n_points, const float* draw, int32_t* live_idx, int32_t* count_out,
                                     int32_t* ws, fn_stream_t stream) {
  FN_CHECK_ARG(n_points > 0 && n_points < ((int64_t)1 << 31) && draw && live_idx && count_out && ws,
               "0 < n_points < 2^31, non-null pointers");
  const int nb = (int)((n_points + CP_PTS - 1) / CP_PTS);
  hipLaunchKernelGGL(cp_count_kernel, dim3(nb), dim3(256), 0, fn::S(stream), n_points, draw, ws);
  hipLaunchKernelGGL(cp_scan_kernel, dim3(1), dim3(1024), 0, fn::S(stream), nb, ws, count_out, (int)n_points);
  hipLaunchKernelGGL(cp_scatter_kernel, dim3(nb), dim3(256), 0, fn::S(stream), n_points, draw, ws, live_idx);
  FN_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// sigma noise of one render_rays call (render.py:162: noise = torch.randn(raw[..., 3].shape) * raw_noise_std): ONE launch fills the noise of
// BOTH passes (coarse [n, S0] and fine [n, S0 + Ni], one allocation) -- Philox4x32-10 keyed by the caller's seed, counter = the float4 index,
// Box-Muller on the four words -- instead of two torch.randn + two multiplications per step (the LLFF configs train with raw_noise_std = 1).
// The values are N(0, std^2) draws of this library's own stream (the reference draws from torch's global generator: distribution parity, like
// the jitter streams; the `pytest=True` hook of render_rays still injects the reference's deterministic numpy numbers).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gauss_noise_kernel(int64_t n4, int64_t n, float sd, uint32_t k0, uint32_t k1, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t o[4];
    philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), 0x6e6f6973u, 0u, k0, k1, o);
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(o[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1): the logarithm stays finite
      const float u2 = (float)(o[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
      const float r = sqrtf(-2.0f * logf(u1)) * sd;
      float sn, cs;
      sincosf(6.28318530717958647692f * u2, &sn, &cs);
      v[2 * h] = r * cs;
      v[2 * h + 1] = r * sn;
    }
    const int64_t e = i * 4;
    if (e + 4 <= n) {
      *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e + q < n) out[e + q] = v[q];
    }
  }
}
extern "C" int fastnerf_gauss_noise(int64_t n, float std, uint64_t seed, float* out, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && std >= 0.f, "n>=0, std>=0");
  if (n == 0) return 0;
  FN_CHECK_ARG(out && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "out: non-null, 16-byte aligned");
  const int64_t n4 = (n + 3) / 4;
  int64_t grid = (n4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gauss_noise_kernel, dim3((unsigned)grid), dim3(256), 0, fn::S(stream), n4, n, std, (uint32_t)seed, (uint32_t)(seed >> 32), out);
  FN_LAUNCH_CHECK();
  return 0;
}
