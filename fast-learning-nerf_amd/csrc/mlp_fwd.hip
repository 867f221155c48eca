// mlp_fwd.hip -- the fused forward of the 8 x 256 NeRF MLP (model.py:38-63; nerf++ MLPNet: nerf_network.py:70-120): positional encoding ->
// 8 layers (skip at 5) -> alpha head -> feature -> view branch -> rgb, one persistent kernel, three math modes (mlp_common.h).
#include "mlp_common.h"

// =========================================================================================
// forward
// =========================================================================================
// inverted-sphere background point (x', y', z', 1/r) of nerf++ (ddp_model.py:16-45)
__device__ __forceinline__ void bg_point(const float* __restrict__ o, const float* __restrict__ d, float depth,
                                         float x[4]) {
  const float dd = fadd(fadd(fmul(d[0], d[0]), fmul(d[1], d[1])), fmul(d[2], d[2]));
  const float od = fadd(fadd(fmul(d[0], o[0]), fmul(d[1], o[1])), fmul(d[2], o[2]));
  const float d1 = -od / dd;
  float pm_[3], ps[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pm_[c] = fadd(o[c], fmul(d1, d[c]));
  const float pmn = sqrtf(fadd(fadd(fmul(pm_[0], pm_[0]), fmul(pm_[1], pm_[1])), fmul(pm_[2], pm_[2])));
  const float dcos = 1.0f / sqrtf(dd);
  const float d2 = fmul(sqrtf(fsub(1.0f, fmul(pmn, pmn))), dcos);
  const float d12 = fadd(d1, d2);
#pragma unroll
  for (int c = 0; c < 3; ++c) ps[c] = fadd(o[c], fmul(d12, d[c]));
  float ax[3] = {fsub(fmul(o[1], ps[2]), fmul(o[2], ps[1])), fsub(fmul(o[2], ps[0]), fmul(o[0], ps[2])),
                 fsub(fmul(o[0], ps[1]), fmul(o[1], ps[0]))};
  const float an = sqrtf(fadd(fadd(fmul(ax[0], ax[0]), fmul(ax[1], ax[1])), fmul(ax[2], ax[2])));
#pragma unroll
  for (int c = 0; c < 3; ++c) ax[c] = ax[c] / an;
  const float ang = fsub(asinf(pmn), asinf(fmul(pmn, depth)));
  const float ca = cosf(ang), sa = sinf(ang);
  const float cr[3] = {fsub(fmul(ax[1], ps[2]), fmul(ax[2], ps[1])), fsub(fmul(ax[2], ps[0]), fmul(ax[0], ps[2])),
                       fsub(fmul(ax[0], ps[1]), fmul(ax[1], ps[0]))};
  const float dot = fadd(fadd(fmul(ax[0], ps[0]), fmul(ax[1], ps[1])), fmul(ax[2], ps[2]));
  const float omc = fsub(1.0f, ca);
  float pn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pn[c] = fadd(fadd(fmul(ps[c], ca), fmul(cr[c], sa)), fmul(fmul(ax[c], dot), omc));
  const float nn = sqrtf(fadd(fadd(fmul(pn[0], pn[0]), fmul(pn[1], pn[1])), fmul(pn[2], pn[2])));
  x[0] = pn[0] / nn; x[1] = pn[1] / nn; x[2] = pn[2] / nn; x[3] = depth;
}

__device__ __forceinline__ int x2idx(int m, int k) { return m * 32 + ((((k >> 2) ^ ((m >> 1) & 7)) << 2) | (k & 3)); }

// BG == false: points o + d*z with the 3-D encoding (63 channels -> E).
// BG == true : nerf++ background net: inverted-sphere points (4-D), samples in flipped order
//              (ddp_model.py:118-124), 84 channels = 64 in E + 20 (padded to 32) in the X2 block that
//              borrows the first 8 KiB of H while H is free (L0) or after it has been consumed (L5).
#define FN_SIN(a) (ABL_NOPE ? (a) : sinf(a))
#define FN_COS(a) (ABL_NOPE ? (a) : cosf(a))
template <bool SAVE, bool BG, int MM = MM_F32>
__global__ void __launch_bounds__(NTHR, 2 * NTHR / 512 * WG_PER_CU)
mlp_fwd_kernel(int64_t P, int S, const float* __restrict__ rays, const float* __restrict__ zv,
               const float* __restrict__ params, const float* __restrict__ packed, float* __restrict__ raw,
               float* __restrict__ act, NetLayout lay, unsigned* __restrict__ sched, const int* __restrict__ live_idx,
               const int* __restrict__ live_cnt, int flags) {
  // live-list mode (exact zero-gradient point compaction, see mlp_bf16.hip / train.hip): row j of the launch is point
  // live_idx[j], the row count is a device value; the saved tensors keep the strides of the capacity PL they were sized for
  const int64_t PL = P;
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;
  float* Es = smem + LDS_H;
  float* X2 = smem;   // [TM][32], aliases the head of H (BG only)
  // tile scheduler word (sched.h): the last two floats of H = columns >= 128 of the last row, stale feature values at
  // the end of a tile and next written by the following tile's layer-0 epilogue, one barrier after everybody read it
  volatile int* sched_word = reinterpret_cast<volatile int*>(smem + LDS_H - 2);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t ntiles = (P + TM - 1) / TM;
  const int PEP = BG ? 96 : 64;
  const int dbg = 0;
  stagger_start();

  for (int64_t tile = blockIdx.x; tile < ntiles;) {
    const int64_t p0 = tile * TM;
    const int valid = (int)((P - p0) < TM ? (P - p0) : TM);
    unsigned long long* maskw =
        SAVE ? reinterpret_cast<unsigned long long*>(act + act_mask(PL, PEP)) + tile * (8 * NWAVES * 64) : nullptr;
    // ---- phase A: points + positional encoding -> Es (+ X2) ---------------------------
    const int pm = tid >> 2, pq = tid & 3;
    int64_t pp = p0 + pm;
    if (pp >= P) pp = P - 1;
    if (live_idx) pp = live_idx[pp];
    const int64_t ray = pp / S;
    const float* rr = rays + ray * 11;
    float x4[4] = {0.f, 0.f, 0.f, 0.f};   // BG: kept live for the L5 re-encode of channels 64..83
    auto write_x2 = [&]() {               // channels 64..95 of the 4-D encoding, dimension pq of row pm
      const float xv = x4[pq];
      X2[x2idx(pm, 0 + pq)] = FN_COS(fmul(xv, 128.0f));
      X2[x2idx(pm, 4 + pq)] = FN_SIN(fmul(xv, 256.0f));
      X2[x2idx(pm, 8 + pq)] = FN_COS(fmul(xv, 256.0f));
      X2[x2idx(pm, 12 + pq)] = FN_SIN(fmul(xv, 512.0f));
      X2[x2idx(pm, 16 + pq)] = FN_COS(fmul(xv, 512.0f));
      X2[x2idx(pm, 20 + pq)] = 0.f; X2[x2idx(pm, 24 + pq)] = 0.f; X2[x2idx(pm, 28 + pq)] = 0.f;
    };
    if (!BG) {
      const float zz = zv[pp];
      float x[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = fadd(rr[c], fmul(rr[3 + c], zz));
      if (pq == 0) {
        Es[eidx(pm, 0)] = x[0]; Es[eidx(pm, 1)] = x[1]; Es[eidx(pm, 2)] = x[2];
        Es[eidx(pm, 63)] = 0.f;
      }
      for (int j = pq; j < 30; j += 4) {
        const int k = j / 3, dim = j - 3 * k;
        const float a = fmul(x[dim], (float)(1 << k));
        Es[eidx(pm, 3 + 6 * k + dim)] = FN_SIN(a);
        Es[eidx(pm, 6 + 6 * k + dim)] = FN_COS(a);
      }
    } else {
      const int sidx = (int)(pp - ray * S);
      const float zz = zv[ray * S + (S - 1 - sidx)];   // flipped sample order
      bg_point(rr, rr + 3, zz, x4);
      const float xv = x4[pq];
      Es[eidx(pm, pq)] = xv;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float a = fmul(xv, (float)(1 << k));
        Es[eidx(pm, 4 + 8 * k + pq)] = FN_SIN(a);
        Es[eidx(pm, 8 + 8 * k + pq)] = FN_COS(a);
      }
      Es[eidx(pm, 60 + pq)] = FN_SIN(fmul(xv, 128.0f));
      write_x2();
    }
    __syncthreads();
    if (SAVE) {
      float* ape = act + act_pe(PL, PEP) + p0 * PEP;
      for (int i = tid; i < TM * 16; i += NTHR) {
        const int m = i >> 4, sl = i & 15;
        if (m < valid)
          store_nt(ape + m * PEP + sl * 4, *reinterpret_cast<const float4*>(Es + m * 64 + ((sl ^ (m & 15)) << 2)));
      }
      if (BG) {
        for (int i = tid; i < TM * 8; i += NTHR) {
          const int m = i >> 3, sl = i & 7;
          if (m < valid)
            store_nt(ape + m * PEP + 64 + sl * 4,
                     *reinterpret_cast<const float4*>(X2 + m * 32 + ((sl ^ ((m >> 1) & 7)) << 2)));
        }
      }
    }
    constexpr bool L16 = MM != MM_F32;    // the bf16x6 kernels: 16 x 16 accumulator tiles
    constexpr bool FOLD = L16;            // ... whose accumulators start from the bias (no bias add in the epilogue)
    // (the forward loads a segment's first weights in its own prologue: a one-layer look-ahead as in dX cost 3.99 -> 4.28 ms, r04 ab_chain2/3)
    AccT<L16, 2> acc;
    constexpr int KS5 = (BG ? 96 + 256 : 64 + 256) / 8;
    // ---- L0 : pe -> 256 -----------------------------------------------------------------
    float bv2[L16 ? 4 : 2];
    load_bias<2>(bv2, params + lay.LB[0], wn, lane);
    init_acc<2, FOLD>(acc, bv2);
    gemm<MM, 2, 1>(acc, Es, 0, 8, wblock<MM>(packed, lay.PF[0]), PEP / 8, 0, wn * 2, wm, lane, dbg);
    if (BG) {
      gemm<MM, 2, 2>(acc, X2, 0, 4, wblock<MM>(packed, lay.PF[0]), PEP / 8, 8, wn * 2, wm, lane, dbg);
      __syncthreads();   // X2 lives in H: everyone must be done with it before H is written
    }
    epilogue_fwd<2, true, SAVE, FOLD>(acc, bv2, Hs, wm, wn, lane, nullptr, 256, valid,
                                      SAVE ? maskw + (0 * NWAVES + wave) * 64 : nullptr);
    __syncthreads();
    // ---- L1..L7 -----------------------------------------------------------------------
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
      const void* B = wblock<MM>(packed, lay.PF[l]);
      load_bias<2>(bv2, params + lay.LB[l], wn, lane);
      init_acc<2, FOLD>(acc, bv2);
      float* sv = SAVE ? act + act_h(PL, PEP, l - 1) + p0 * 256 : nullptr;   // h_{l-1} is what this loop reads
      if (l == 5) {
        if (!BG) {
          gemm<MM, 2, 1>(acc, Es, 0, 8, B, KS5, 0, wn * 2, wm, lane, dbg);
          gemm<MM, 2, 0>(acc, Hs, 0, 32, B, KS5, 8, wn * 2, wm, lane, dbg, sv, valid, wave);
        } else {
          gemm<MM, 2, 0>(acc, Hs, 0, 32, B, KS5, 12, wn * 2, wm, lane, dbg, sv, valid, wave);
          __syncthreads();   // h4 consumed: its first 8 KiB become X2 again
          write_x2();
          __syncthreads();
          gemm<MM, 2, 1>(acc, Es, 0, 8, B, KS5, 0, wn * 2, wm, lane, dbg);
          gemm<MM, 2, 2>(acc, X2, 0, 4, B, KS5, 8, wn * 2, wm, lane, dbg);
        }
      } else {
        gemm<MM, 2, 0>(acc, Hs, 0, 32, B, 32, 0, wn * 2, wm, lane, dbg, sv, valid, wave);
      }
      __syncthreads();  // every wave has finished reading H
      epilogue_fwd<2, true, SAVE, FOLD>(acc, bv2, Hs, wm, wn, lane, nullptr, 256, valid,
                                        SAVE ? maskw + (l * NWAVES + wave) * 64 : nullptr);
      __syncthreads();
    }
    // ---- alpha head (VALU) + view-direction encoding -> Es ---------------------------
    float alpha_val = 0.f;
    {
      const float* wa = params + lay.AW;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = pq * 64 + i * 4;
        const float4 h = *reinterpret_cast<const float4*>(Hs + pm * 256 + ((((k >> 2) ^ (pm & 15))) << 2));
        const float4 w = *reinterpret_cast<const float4*>(wa + k);
        s = fmaf(h.x, w.x, s); s = fmaf(h.y, w.y, s); s = fmaf(h.z, w.z, s); s = fmaf(h.w, w.w, s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      alpha_val = s + params[lay.AB];
      float v[3] = {rr[8], rr[9], rr[10]};
      if (pq == 0) {
        Es[eidx(pm, 0)] = v[0]; Es[eidx(pm, 1)] = v[1]; Es[eidx(pm, 2)] = v[2];
#pragma unroll
        for (int c = 27; c < 32; ++c) Es[eidx(pm, c)] = 0.f;
      }
      for (int j = pq; j < 12; j += 4) {
        const int k = j / 3, dim = j - 3 * k;
        const float a = fmul(v[dim], (float)(1 << k));
        Es[eidx(pm, 3 + 6 * k + dim)] = FN_SIN(a);
        Es[eidx(pm, 6 + 6 * k + dim)] = FN_COS(a);
      }
    }
    // FN_FWD_SKIP_DEAD_RGB (see mlp_bf16.hip): a tile without a live sample skips the feature / view / colour layers.  One word
    // per wave in channels 56..63 of row 0 of the encoding tile (free since layer 5; the direction encoding uses 0..31).
    bool skip_tail = false;
    if (!SAVE && !BG && (flags & 1)) {
      const unsigned long long any_live = __ballot((pm < valid) && (alpha_val > 0.f));
      volatile int* slot = reinterpret_cast<volatile int*>(Es + 56);
      if (lane == 0) slot[wave] = any_live != 0ull;
      __syncthreads();
      int any = 0;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) any |= slot[w];
      skip_tail = any == 0;
    }
    if (skip_tail) {
      if (pq == 0 && pm < valid && raw) *reinterpret_cast<float4*>(raw + (p0 + pm) * 4) = make_float4(0.f, 0.f, 0.f, alpha_val);
    } else {
    // ---- feature layer (no ReLU) ------------------------------------------------------
    load_bias<2>(bv2, params + lay.FB, wn, lane);
    init_acc<2, FOLD>(acc, bv2);
    gemm<MM, 2, 0>(acc, Hs, 0, 32, wblock<MM>(packed, lay.PF[8]), 32, 0, wn * 2, wm, lane, dbg,
                   SAVE ? act + act_h(PL, PEP, 7) + p0 * 256 : nullptr, valid, wave);
    __syncthreads();
    epilogue_fwd<2, false, false, FOLD>(acc, bv2, Hs, wm, wn, lane, nullptr, 256, valid);
    __syncthreads();
    if (SAVE) {
      float* avp = act + act_vpe(PL, PEP) + p0 * 32;
      for (int i = tid; i < TM * 8; i += NTHR) {
        const int m = i >> 3, sl = i & 7;
        if (m < valid)
          store_nt(avp + m * 32 + sl * 4, *reinterpret_cast<const float4*>(Es + m * 64 + ((sl ^ (m & 15)) << 2)));
      }
    }
    // ---- view layer: [feat | vpe32] -> 128, ReLU ---------------------------------------
    {
      AccT<L16, 1> av;
      float bv1[L16 ? 2 : 1];
      load_bias<1>(bv1, params + lay.VB, wn, lane);
      init_acc<1, FOLD>(av, bv1);
      gemm<MM, 1, 0>(av, Hs, 0, 32, wblock<MM>(packed, lay.PF[9]), 36, 0, wn, wm, lane, dbg,
                     SAVE ? act + act_feat(PL, PEP) + p0 * 256 : nullptr, valid, wave);
      gemm<MM, 1, 1>(av, Es, 0, 4, wblock<MM>(packed, lay.PF[9]), 36, 32, wn, wm, lane, dbg);
      __syncthreads();
      epilogue_fwd<1, true, false, FOLD>(av, bv1, Hs, wm, wn, lane, nullptr, 128, valid);
      __syncthreads();
      if (SAVE) {   // hv: 32 slots per row, whole 512-byte rows per half wave
        float* ahv = act + act_hv(PL, PEP) + p0 * 128;
        for (int i = tid; i < TM * 32; i += NTHR) {
          const int m = i >> 5, sl = i & 31;
          if (m < valid)
            store_nt(ahv + m * 128 + ((sl ^ (m & 15)) << 2), *reinterpret_cast<const float4*>(Hs + m * 256 + sl * 4));
        }
      }
    }
    // ---- rgb head (VALU) + output -------------------------------------------------------
    {
      const float* wr = params + lay.RW;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = pq * 32 + i * 4;
        const float4 h = *reinterpret_cast<const float4*>(Hs + pm * 256 + ((((k >> 2) ^ (pm & 15))) << 2));
        const float4 w0 = *reinterpret_cast<const float4*>(wr + k);
        const float4 w1 = *reinterpret_cast<const float4*>(wr + 128 + k);
        const float4 w2 = *reinterpret_cast<const float4*>(wr + 256 + k);
        s0 = fmaf(h.x, w0.x, s0); s0 = fmaf(h.y, w0.y, s0); s0 = fmaf(h.z, w0.z, s0); s0 = fmaf(h.w, w0.w, s0);
        s1 = fmaf(h.x, w1.x, s1); s1 = fmaf(h.y, w1.y, s1); s1 = fmaf(h.z, w1.z, s1); s1 = fmaf(h.w, w1.w, s1);
        s2 = fmaf(h.x, w2.x, s2); s2 = fmaf(h.y, w2.y, s2); s2 = fmaf(h.z, w2.z, s2); s2 = fmaf(h.w, w2.w, s2);
      }
      s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64);
      s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
      s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
      if (pq == 0 && pm < valid && raw) {
        float4 o;
        o.x = s0 + params[lay.RB]; o.y = s1 + params[lay.RB + 1]; o.z = s2 + params[lay.RB + 2]; o.w = alpha_val;
        *reinterpret_cast<float4*>(raw + (p0 + pm) * 4) = o;
      }
    }
    }   // !skip_tail
    tile = b_next_tile(sched, sched_word, tid);   // closing barrier inside: H / Es are rewritten by the next tile
  }
  b_sched_exit(sched, tid);
}

template <bool SAVE, bool BG, int MM>
static int fwd_launch_t(int grid, hipStream_t st, int64_t P, int S, const float* rays11, const float* z, const float* params,
                        const float* packed_fwd, float* raw, float* act, const NetLayout& lay, unsigned* sched, const int* live_idx,
                        const int* live_cnt, int flags) {
  auto kern = mlp_fwd_kernel<SAVE, BG, MM>;
  static bool attr = false;
  if (!attr) {
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), LDS_BYTES, st, P, S, rays11, z, params, packed_fwd, raw, act, lay, sched, live_idx,
                     live_cnt, flags);
  FN_LAUNCH_CHECK();
  return 0;
}

static int fwd_launch(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                      const float* packed_fwd, float* raw, float* act, const int* live_idx, const int* live_cnt,
                      fn_stream_t stream, int flags = 0, int mm = MM_F32) {
  const NetLayout& lay = layout_of(kind);
  const int64_t P = n * S;
  const int64_t ntiles = (P + TM - 1) / TM;
  int grid = num_cus() * WG_PER_CU;
  if (ntiles < grid) grid = (int)ntiles;
  hipStream_t st = fn::S(stream);
  unsigned* sched = b_sched_pair();
  FN_CHECK_ARG(sched != nullptr, "scheduler counters (hipMalloc failed?)");
  const int fl = (kind == 0 && !live_idx && !act) ? flags : 0;
#define FN_FWD(SAVE_, BG_, MM_) \
  fwd_launch_t<SAVE_, BG_, MM_>(grid, st, P, S, rays11, z, params, packed_fwd, raw, act, lay, sched, live_idx, live_cnt, fl)
  if (mm == MM_X6) {
    if (kind == 2) return act ? FN_FWD(true, true, MM_X6) : FN_FWD(false, true, MM_X6);
    return act ? FN_FWD(true, false, MM_X6) : FN_FWD(false, false, MM_X6);
  }
  if (kind == 2) return act ? FN_FWD(true, true, MM_F32) : FN_FWD(false, true, MM_F32);
  return act ? FN_FWD(true, false, MM_F32) : FN_FWD(false, false, MM_F32);
#undef FN_FWD
}
extern "C" int fastnerf_mlp_fwd_ex(int kind, int64_t n, int S, const float* rays11, const float* z,
                                   const float* params, const float* packed_fwd, float* raw, float* act,
                                   fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n >= 0 && S >= 1, "kind in 0..2, n>=0, S>=1");
  FN_CHECK_ARG(n == 0 || (rays11 && z && params && packed_fwd && raw), "null pointer");
  if (n == 0) return 0;
  return fwd_launch(kind, n, S, rays11, z, params, packed_fwd, raw, act, nullptr, nullptr, stream);
}
// exact-fp32 twin of fastnerf_mlp_bf16_fwd_live
extern "C" int fastnerf_mlp_fwd_flags_ex(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                                         const float* packed_fwd, float* raw, int flags, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n >= 0 && S >= 1, "kind in 0..2, n>=0, S>=1");
  FN_CHECK_ARG(n == 0 || (rays11 && z && params && packed_fwd && raw), "null pointer");
  if (n == 0) return 0;
  return fwd_launch(kind, n, S, rays11, z, params, packed_fwd, raw, nullptr, nullptr, nullptr, stream, flags);
}

extern "C" int fastnerf_mlp_fwd_live_ex(int kind, int64_t n, int S, const float* rays11, const float* z,
                                        const float* params, const float* packed_fwd, float* act,
                                        const int32_t* live_idx, const int32_t* live_cnt, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(rays11 && z && params && packed_fwd && act && live_idx && live_cnt, "null pointer");
  FN_CHECK_ARG(n * (int64_t)S < ((int64_t)1 << 31), "live lists index points with int32");
  return fwd_launch(kind, n, S, rays11, z, params, packed_fwd, nullptr, act, live_idx, live_cnt, stream);
}
extern "C" int fastnerf_mlp_fwd(int64_t n, int S, const float* rays11, const float* z, const float* params,
                                const float* packed_fwd, float* raw, float* act, fn_stream_t stream) {
  return fastnerf_mlp_fwd_ex(0, n, S, rays11, z, params, packed_fwd, raw, act, stream);
}

// ---- MM_X6 ("bf16x6") entry points: the call protocol of fastnerf_mlp_{fwd,bwd}_ex / _live_ex / _flags_ex, weights from
// fastnerf_mlp_x6_pack; saved activations and gradient workspaces have the exact-fp32 kernels' layouts and sizes.
extern "C" int fastnerf_mlp_x6_fwd(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                                   const float* packed_fwd, float* raw, float* act, int flags, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n >= 0 && S >= 1, "kind in 0..2, n>=0, S>=1");
  FN_CHECK_ARG(n == 0 || (rays11 && z && params && packed_fwd && raw), "null pointer");
  if (n == 0) return 0;
  return fwd_launch(kind, n, S, rays11, z, params, packed_fwd, raw, act, nullptr, nullptr, stream, flags, MM_X6);
}
extern "C" int fastnerf_mlp_x6_fwd_live(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                                        const float* packed_fwd, float* act, const int32_t* live_idx, const int32_t* live_cnt,
                                        fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(rays11 && z && params && packed_fwd && act && live_idx && live_cnt, "null pointer");
  FN_CHECK_ARG(n * (int64_t)S < ((int64_t)1 << 31), "live lists index points with int32");
  return fwd_launch(kind, n, S, rays11, z, params, packed_fwd, nullptr, act, live_idx, live_cnt, stream, 0, MM_X6);
}
