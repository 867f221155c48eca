// composite.hip -- alpha compositing (raw2outputs) forward/backward and the hierarchical
// sampler (inverse-CDF + sort-merge) for gfx950.
//
// Mapping: ONE 64-lane wavefront per ray.  Each lane owns a contiguous chunk of C =
// ceil(S/64) samples (S=64 -> 1, S=192 -> 3); the transmittance product and the backward
// suffix sums are wave-level scans done with DPP/`__shfl_up` (no LDS round trip, no barrier).
// The sampler keeps the ray's cdf/bins/new samples in a per-wave LDS slice so that the
// per-sample binary searches and the bitonic sort are bank-parallel.
//
// Reference semantics (nerf-ours/): raw2outputs render.py:149-192; sample_pdf
// run_nerf_helpers.py:112-155; merge render.py:279-283,299.
#include "common.h"

#define WAVE 64
#define MAXC 8  // up to 512 samples per ray

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// inclusive scans over lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    float t = __shfl_up(v, o, WAVE);
    if (lane >= o) v *= t;
  }
  return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
  for (int o = 1; o < WAVE; o <<= 1) {
    float t = __shfl_up(v, o, WAVE);
    if (lane >= o) v += t;
  }
  return v;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

struct SampleVals {
  float alpha, t, dist, z, sig;
  float c[3];
};

// Variants of the compositing rule:
//   nerf-ours raw2outputs (render.py:149-192)      : relu, eps 1e-10, last dist 1e10, dists * |d|
//   nerf++ foreground (ddp_model.py:97-107)         : abs,  eps 1e-6,  last dist fg_far - z_last, * |d|
//   nerf++ background (ddp_model.py:118-135)        : abs,  eps 1e-6,  z flipped (1 -> 0), last 1e10, no |d|
struct CompCfg {
  int act_abs;    // 0: relu(sigma)   1: |sigma|
  float eps;      // added to 1 - alpha
  int last_mode;  // 0: 1e10   1: far[r] - z_last
  int use_dnorm;  // multiply dists by |rays_d|
  int flip;       // samples are stored near->far but consumed far->near (z index S-1-s)
};
__device__ __forceinline__ CompCfg nerf_cfg() { return CompCfg{0, 1e-10f, 0, 1, 0}; }

// per-lane chunk evaluation shared by forward and backward
template <bool WITH_RGB>
__device__ __forceinline__ void eval_chunk(int S, int C, int lane, const float* __restrict__ raw,
                                           const float* __restrict__ z, const float* __restrict__ noise, float dnorm,
                                           SampleVals v[MAXC], int& cnt, const CompCfg cfg, float far) {
  const int s0 = lane * C;
  cnt = 0;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    if (j >= C) break;
    const int s = s0 + j;
    if (s >= S) break;
    const float4 r = *reinterpret_cast<const float4*>(raw + (int64_t)s * 4);
    const float zs = cfg.flip ? z[S - 1 - s] : z[s];
    float dist;
    if (s + 1 < S) {
      const float zn = cfg.flip ? z[S - 2 - s] : z[s + 1];
      dist = cfg.flip ? fsub(zs, zn) : fsub(zn, zs);
    } else {
      dist = cfg.last_mode ? fsub(far, zs) : 1e10f;
    }
    if (cfg.use_dnorm) dist = fmul(dist, dnorm);
    float sig = r.w;
    if (noise) sig = fadd(sig, noise[s]);
    const float rs = cfg.act_abs ? fabsf(sig) : fmaxf(sig, 0.0f);
    const float alpha = fsub(1.0f, expf(-fmul(rs, dist)));
    v[j].alpha = alpha;
    v[j].t = fadd(fsub(1.0f, alpha), cfg.eps);
    v[j].dist = dist;
    v[j].z = zs;
    v[j].sig = sig;
    if (WITH_RGB) {
      v[j].c[0] = sigmoidf(r.x);
      v[j].c[1] = sigmoidf(r.y);
      v[j].c[2] = sigmoidf(r.z);
    }
    cnt = j + 1;
  }
}

__global__ void __launch_bounds__(256) raw2outputs_fwd_kernel(int64_t n, int S, const float* __restrict__ raw,
                                                               const float* __restrict__ z,
                                                               const float* __restrict__ rays,
                                                               const float* __restrict__ noise, int white,
                                                               float* __restrict__ rgb_map, float* __restrict__ disp,
                                                               float* __restrict__ acc, float* __restrict__ weights,
                                                               float* __restrict__ depth, const CompCfg cfg,
                                                               const float* __restrict__ far,
                                                               float* __restrict__ lambda) {
  const int lane = threadIdx.x & 63;
  const int C = (S + WAVE - 1) / WAVE;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < n; r += nwaves) {
    const float* rr = rays + r * 11;
    const float dnorm = sqrtf(fadd(fadd(fmul(rr[3], rr[3]), fmul(rr[4], rr[4])), fmul(rr[5], rr[5])));
    SampleVals v[MAXC];
    int cnt;
    eval_chunk<true>(S, C, lane, raw + r * S * 4, z + r * S, noise ? noise + r * S : nullptr, dnorm, v, cnt, cfg,
                     far ? far[r] : 0.f);
    float prod = 1.0f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < cnt) prod *= v[j].t;
    const float incl = wave_scan_mul(prod, lane);
    float T = __shfl_up(incl, 1, WAVE);
    if (lane == 0) T = 1.0f;
    if (lambda) {   // product over ALL samples (nerf++ bg_lambda, ddp_model.py:103)
      const float tot = __shfl(incl, WAVE - 1, WAVE);
      if (lane == 0) lambda[r] = tot;
    }
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      if (j < cnt) {
        const float w = v[j].alpha * T;
        if (weights) weights[r * S + lane * C + j] = w;
        sr += w * v[j].c[0];
        sg += w * v[j].c[1];
        sb += w * v[j].c[2];
        sd += w * v[j].z;
        sa += w;
        T *= v[j].t;
      }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
    if (lane == 0) {
      if (white) {
        const float bg = fsub(1.0f, sa);
        sr = fadd(sr, bg); sg = fadd(sg, bg); sb = fadd(sb, bg);
      }
      rgb_map[r * 3 + 0] = sr; rgb_map[r * 3 + 1] = sg; rgb_map[r * 3 + 2] = sb;
      if (disp) {
        // torch.max propagates NaN (render.py:186): a ray that hits nothing (acc == 0 exactly) has depth / acc = 0 / 0 and the
        // reference's disparity is NaN there; fmaxf alone would return 1e-10 and report 1e10
        const float q = sd / sa;
        disp[r] = (q != q) ? q : 1.0f / fmaxf(1e-10f, q);
      }
      if (acc) acc[r] = sa;
      if (depth) depth[r] = sd;
    }
  }
}

__global__ void __launch_bounds__(256) raw2outputs_bwd_kernel(int64_t n, int S, const float* __restrict__ raw,
                                                               const float* __restrict__ z,
                                                               const float* __restrict__ rays,
                                                               const float* __restrict__ noise, int white,
                                                               const float* __restrict__ g_rgb,
                                                               float* __restrict__ draw, const CompCfg cfg,
                                                               const float* __restrict__ far,
                                                               const float* __restrict__ g_lambda) {
  const int lane = threadIdx.x & 63;
  const int C = (S + WAVE - 1) / WAVE;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < n; r += nwaves) {
    const float* rr = rays + r * 11;
    const float dnorm = sqrtf(fadd(fadd(fmul(rr[3], rr[3]), fmul(rr[4], rr[4])), fmul(rr[5], rr[5])));
    const float g0 = g_rgb[r * 3], g1 = g_rgb[r * 3 + 1], g2 = g_rgb[r * 3 + 2];
    const float gbg = white ? (g0 + g1 + g2) : 0.0f;
    SampleVals v[MAXC];
    int cnt;
    eval_chunk<true>(S, C, lane, raw + r * S * 4, z + r * S, noise ? noise + r * S : nullptr, dnorm, v, cnt, cfg,
                     far ? far[r] : 0.f);
    float prod = 1.0f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < cnt) prod *= v[j].t;
    const float incl = wave_scan_mul(prod, lane);
    float T0 = __shfl_up(incl, 1, WAVE);
    if (lane == 0) T0 = 1.0f;
    // d(lambda)/d(alpha_i) = -lambda / t_i : folds into the suffix term
    const float lam_term = g_lambda ? g_lambda[r] * __shfl(incl, WAVE - 1, WAVE) : 0.0f;
    // pass 1: weights, G*w chunk sums
    float w[MAXC], G[MAXC], Tj[MAXC];
    float gw_chunk = 0.f;
    float T = T0;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      if (j < cnt) {
        Tj[j] = T;
        w[j] = v[j].alpha * T;
        G[j] = g0 * v[j].c[0] + g1 * v[j].c[1] + g2 * v[j].c[2] - gbg;
        gw_chunk += G[j] * w[j];
        T *= v[j].t;
      }
    }
    const float incl_gw = wave_scan_add(gw_chunk, lane);
    const float total = __shfl(incl_gw, WAVE - 1, WAVE);
    float suffix = total - incl_gw;  // sum over lanes > this lane
    // walk the chunk backwards: suffix = sum_{j' > j} G w
    float4 out[MAXC];
#pragma unroll
    for (int j = MAXC - 1; j >= 0; --j) {
      if (j < cnt) {
        const float dalpha = G[j] * Tj[j] - (suffix + lam_term) / v[j].t;
        const float dact = dalpha * v[j].dist * (1.0f - v[j].alpha);
        const float dsig = cfg.act_abs ? ((v[j].sig > 0.0f) ? dact : ((v[j].sig < 0.0f) ? -dact : 0.0f))
                                       : ((v[j].sig > 0.0f) ? dact : 0.0f);
        out[j].x = g0 * w[j] * v[j].c[0] * (1.0f - v[j].c[0]);
        out[j].y = g1 * w[j] * v[j].c[1] * (1.0f - v[j].c[1]);
        out[j].z = g2 * w[j] * v[j].c[2] * (1.0f - v[j].c[2]);
        out[j].w = dsig;
        suffix += G[j] * w[j];
      }
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < cnt) *reinterpret_cast<float4*>(draw + (r * S + lane * C + j) * 4) = out[j];
  }
}

// ---------------------------------------------------------------------------------------
// sample_pdf + merge
// ---------------------------------------------------------------------------------------
// number of elements of sorted a[0..n) that are <  x  (lower_bound) / <= x (upper_bound)
__device__ __forceinline__ int lower_bound_f(const float* a, int n, float x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int upper_bound_f(const float* a, int n, float x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// LDS per wave: zs[S] | cdf[S] | bins[S] | smp[NP]   (NP = Ni rounded up to a power of two)
// bins_mode != 0: `z` holds the bins themselves ([n,S-1]) and `weights` is [n,S-2]; no merge.
// variant 1 = nerf++ sampler (ddp_train_nerf.py:84-133): eps 1e-6, index = #(u >= cdf[:M]), +1e-6 bin width
__global__ void __launch_bounds__(256) sample_pdf_merge_kernel(int64_t n, int S, int Ni, int NP, int bins_mode,
                                                                int variant,
                                                                const float* __restrict__ z,
                                                                const float* __restrict__ weights, int det,
                                                                const float* __restrict__ u_in, uint64_t seed,
                                                                float* __restrict__ z_out,
                                                                float* __restrict__ z_samples,
                                                                float* __restrict__ z_std) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int per_wave = 3 * S + NP;
  float* zs = smem + wib * per_wave;
  float* cdf = zs + S;
  float* bins = cdf + S;
  float* smp = bins + S;
  const int M = S - 1;   // len(bins) == len(cdf)
  const int NW = S - 2;  // len(weights[1:-1])
  const int CW = (NW + WAVE - 1) / WAVE;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < n; r += nwaves) {
    if (bins_mode) {
      for (int k = lane; k < M; k += WAVE) bins[k] = z[r * M + k];
    } else {
      for (int s = lane; s < S; s += WAVE) zs[s] = z[r * S + s];
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < M; k += WAVE) bins[k] = fmul(0.5f, fadd(zs[k + 1], zs[k]));
    }
    // pdf / cdf : lane owns weights k in [lane*CW, lane*CW+CW).  The sum and the running cdf are
    // accumulated in fp64 and rounded once per entry (torch's CPU cumsum accumulates float in
    // double as well), which keeps the cdf within 1 ulp of the reference's.
    float wl[MAXC];
    double csum = 0.0;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      const int k = lane * CW + j;
      wl[j] = (j < CW && k < NW) ? fadd(bins_mode ? weights[r * NW + k] : weights[r * S + k + 1], variant ? 1e-6f : 1e-5f) : 0.f;
      csum += (double)wl[j];
    }
    double totd = csum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) totd += __shfl_xor(totd, o, WAVE);
    const float tot = (float)totd;
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      wl[j] = wl[j] / tot;
      run += (double)wl[j];
    }
    double incl = run;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
      const double t = __shfl_up(incl, o, WAVE);
      if (lane >= o) incl += t;
    }
    double base = incl - run;  // exclusive prefix of this lane's chunk
    if (lane == 0) cdf[0] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
      const int k = lane * CW + j;
      if (j < CW && k < NW) {
        base += (double)wl[j];
        cdf[k + 1] = (float)base;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // inverse cdf
    float lsum = 0.f;
    for (int i = lane; i < NP; i += WAVE) {
      float smpl = __builtin_inff();
      if (i < Ni) {
        float u;
        if (u_in) {
          u = u_in[r * Ni + i];
        } else if (det) {
          const float step = 1.0f / (float)(Ni - 1);
          u = (Ni == 1) ? 0.0f : ((i < Ni / 2) ? fmul(step, (float)i) : fsub(1.0f, fmul(step, (float)(Ni - 1 - i))));
        } else {
          uint32_t o[4];
          const uint64_t idx = (uint64_t)r * Ni + i;
          philox4x32((uint32_t)idx, (uint32_t)(idx >> 32), 0x70646673u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
          u = u01(o[0]);
        }
        const int inds = upper_bound_f(cdf, variant ? NW : M, u);  // searchsorted(right=True) / count(u >= cdf[:M])
        const int below = inds - 1 > 0 ? inds - 1 : 0;
        const int above = inds < M - 1 ? inds : M - 1;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = bins[below], b1 = bins[above];
        float denom = fsub(c1, c0);
        if (denom < (variant ? 1e-6f : 1e-5f)) denom = 1.0f;
        const float t = fsub(u, c0) / denom;
        smpl = variant ? fadd(b0, fmul(t, fadd(fsub(b1, b0), 1e-6f))) : fadd(b0, fmul(t, fsub(b1, b0)));
        if (z_samples) z_samples[r * Ni + i] = smpl;
        lsum += smpl;
      }
      smp[i] = smpl;
    }
    if (z_std) {
      const float mean = wave_sum(lsum) / (float)Ni;
      float lv = 0.f;
      for (int i = lane; i < Ni; i += WAVE) {
        const float d = smp[i] - mean;
        lv += d * d;
      }
      const float var = wave_sum(lv) / (float)Ni;
      if (lane == 0) z_std[r] = sqrtf(var);
    }
    __builtin_amdgcn_wave_barrier();
    if (bins_mode) continue;
    // bitonic sort of smp[0..NP) inside the wave
    for (int k = 2; k <= NP; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < NP; i += WAVE) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const float a = smp[i], b = smp[ixj];
            const bool up = ((i & k) == 0);
            if ((a > b) == up) { smp[i] = b; smp[ixj] = a; }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // merge by rank: coarse z first on ties
    const int So = S + Ni;
    for (int s = lane; s < S; s += WAVE) {
      const float x = zs[s];
      z_out[r * So + s + lower_bound_f(smp, Ni, x)] = x;
    }
    for (int i = lane; i < Ni; i += WAVE) {
      const float x = smp[i];
      z_out[r * So + i + upper_bound_f(zs, S, x)] = x;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static inline int grid_waves(int64_t n) {
  int64_t g = (n + 3) / 4;  // 4 waves (rays) per 256-thread block
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;
  return (int)g;
}

extern "C" int fastnerf_raw2outputs_fwd(int64_t n, int S, const float* raw, const float* z, const float* rays11,
                                        const float* noise, int white_bkgd, float* rgb_map, float* disp, float* acc,
                                        float* weights, float* depth, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 1 && S <= WAVE * MAXC, "n>=0, 1<=S<=512");
  FN_CHECK_ARG(n == 0 || (raw && z && rays11 && rgb_map), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(raw2outputs_fwd_kernel, dim3(grid_waves(n)), dim3(256), 0, fn::S(stream), n, S, raw, z, rays11,
                     noise, white_bkgd, rgb_map, disp, acc, weights, depth, CompCfg{0, 1e-10f, 0, 1, 0},
                     (const float*)nullptr, (float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_raw2outputs_bwd(int64_t n, int S, const float* raw, const float* z, const float* rays11,
                                        const float* noise, int white_bkgd, const float* g_rgb, float* draw,
                                        fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 1 && S <= WAVE * MAXC, "n>=0, 1<=S<=512");
  FN_CHECK_ARG(n == 0 || (raw && z && rays11 && g_rgb && draw), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(raw2outputs_bwd_kernel, dim3(grid_waves(n)), dim3(256), 0, fn::S(stream), n, S, raw, z, rays11,
                     noise, white_bkgd, g_rgb, draw, CompCfg{0, 1e-10f, 0, 1, 0}, (const float*)nullptr,
                     (const float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_sample_pdf_merge(int64_t n, int S, int Ni, const float* z, const float* weights, int det,
                                         const float* u, uint64_t seed, float* z_out, float* z_samples, float* z_std,
                                         fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 3 && S <= WAVE * MAXC && Ni >= 1 && Ni <= 1024, "n>=0, 3<=S<=512, 1<=Ni<=1024");
  FN_CHECK_ARG(n == 0 || (z && weights && z_out), "null pointer");
  if (n == 0) return 0;
  int NP = 1;
  while (NP < Ni) NP <<= 1;
  const size_t lds = (size_t)4 * (3 * S + NP) * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3(grid_waves(n)), dim3(256), lds, fn::S(stream), n, S, Ni, NP, 0, 0, z,
                     weights, det, u, seed, z_out, z_samples, z_std);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_sample_pdf(int64_t n, int M, int Ni, const float* bins, const float* weights, int det,
                                   const float* u, uint64_t seed, float* samples, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && M >= 2 && M + 1 <= WAVE * MAXC && Ni >= 1 && Ni <= 1024, "n>=0, 2<=M<=511, 1<=Ni<=1024");
  FN_CHECK_ARG(n == 0 || (bins && weights && samples), "null pointer");
  if (n == 0) return 0;
  int NP = 1;
  while (NP < Ni) NP <<= 1;
  const int S = M + 1;
  const size_t lds = (size_t)4 * (3 * S + NP) * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3(grid_waves(n)), dim3(256), lds, fn::S(stream), n, S, Ni, NP, 1, 0,
                     bins, weights, det, u, seed, (float*)nullptr, samples, (float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------------------
// nerf++ compositing (ddp_model.py:97-135): part 0 = foreground, part 1 = background
// ---------------------------------------------------------------------------------------
static inline CompCfg pp_cfg(int part) { return part == 0 ? CompCfg{1, 1e-6f, 1, 1, 0} : CompCfg{1, 1e-6f, 0, 0, 1}; }

extern "C" int fastnerf_pp_composite_fwd(int64_t n, int S, int part, const float* raw, const float* z,
                                         const float* rays11, const float* fg_far, float* rgb_map, float* weights,
                                         float* depth, float* lambda, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 2 && S <= WAVE * MAXC && (part == 0 || part == 1), "n>=0, 2<=S<=512, part in {0,1}");
  FN_CHECK_ARG(n == 0 || (raw && z && rays11 && rgb_map && (part == 1 || fg_far)), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(raw2outputs_fwd_kernel, dim3(grid_waves(n)), dim3(256), 0, fn::S(stream), n, S, raw, z, rays11,
                     (const float*)nullptr, 0, rgb_map, (float*)nullptr, (float*)nullptr, weights, depth, pp_cfg(part),
                     part == 0 ? fg_far : (const float*)nullptr, part == 0 ? lambda : (float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_pp_composite_bwd(int64_t n, int S, int part, const float* raw, const float* z,
                                         const float* rays11, const float* fg_far, const float* g_rgb,
                                         const float* g_lambda, float* draw, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 2 && S <= WAVE * MAXC && (part == 0 || part == 1), "n>=0, 2<=S<=512, part in {0,1}");
  FN_CHECK_ARG(n == 0 || (raw && z && rays11 && g_rgb && draw && (part == 1 || fg_far)), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(raw2outputs_bwd_kernel, dim3(grid_waves(n)), dim3(256), 0, fn::S(stream), n, S, raw, z, rays11,
                     (const float*)nullptr, 0, g_rgb, draw, pp_cfg(part), part == 0 ? fg_far : (const float*)nullptr,
                     part == 0 ? g_lambda : (const float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}

// nerf++: sample_pdf on mid(z) / weights[1:-1] + sort(cat) (ddp_train_nerf.py:369-382)
extern "C" int fastnerf_pp_sample_pdf_merge(int64_t n, int S, int Ni, const float* z, const float* weights, int det,
                                            const float* u, uint64_t seed, float* z_out, float* z_samples,
                                            fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 3 && S <= WAVE * MAXC && Ni >= 1 && Ni <= 1024, "n>=0, 3<=S<=512, 1<=Ni<=1024");
  FN_CHECK_ARG(n == 0 || (z && weights && z_out), "null pointer");
  if (n == 0) return 0;
  int NP = 1;
  while (NP < Ni) NP <<= 1;
  const size_t lds = (size_t)4 * (3 * S + NP) * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3(grid_waves(n)), dim3(256), lds, fn::S(stream), n, S, Ni, NP, 0, 1, z,
                     weights, det, u, seed, z_out, z_samples, (float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}

// nerf++ stand-alone sample_pdf(bins [n,M], weights [n,M-1]) -> samples [n,Ni] (ddp_train_nerf.py:84-133 as its callers outside
// train_step use it): eps 1e-6, index = #(u >= cdf[:M]), +1e-6 bin width.
extern "C" int fastnerf_pp_sample_pdf(int64_t n, int M, int Ni, const float* bins, const float* weights, int det, const float* u,
                                      uint64_t seed, float* samples, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && M >= 2 && M + 1 <= WAVE * MAXC && Ni >= 1 && Ni <= 1024, "n>=0, 2<=M<=511, 1<=Ni<=1024");
  FN_CHECK_ARG(n == 0 || (bins && weights && samples), "null pointer");
  if (n == 0) return 0;
  int NP = 1;
  while (NP < Ni) NP <<= 1;
  const int S = M + 1;
  const size_t lds = (size_t)4 * (3 * S + NP) * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_merge_kernel, dim3(grid_waves(n)), dim3(256), lds, fn::S(stream), n, S, Ni, NP, 1, 1,
                     bins, weights, det, u, seed, (float*)nullptr, samples, (float*)nullptr);
  FN_LAUNCH_CHECK();
  return 0;
}
