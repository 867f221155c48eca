// mlp_bwd_dx.hip -- the dY chain of the backward pass: one persistent kernel walks the layers in reverse with transposed packed weights and
// leaves every layer's pre-activation gradient in HBM for the dW jobs (mlp_bwd_dw.hip launches it through fn_launch_dx).
#include "mlp_common.h"

// =========================================================================================
// backward: dX chain
// =========================================================================================
// epilogue: optional rank-1 term (dalpha x wa), ReLU mask from the forward's ballot words, write H
// and the pre-activation gradient buffer.
struct DxPre {  // loaded before the k-loop (see load_bias)
  unsigned mlo, mhi;
  float wan[4];   // rank-1 weights of this lane's columns: two column tiles of 32, or four of 16 (L16)
};
template <bool MASK, bool RANK1, bool L16 = false>
__device__ __forceinline__ DxPre dx_preload(const unsigned long long* __restrict__ maskw, const float* __restrict__ wa,
                                            int wn, int lane) {
  DxPre p;
  p.mlo = p.mhi = 0u;
  p.wan[0] = p.wan[1] = p.wan[2] = p.wan[3] = 0.f;
  if (MASK) {
    const unsigned long long w = maskw[lane];
    p.mlo = (unsigned)w;
    p.mhi = (unsigned)(w >> 32);
  }
  if (RANK1) {
    if constexpr (L16) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) p.wan[ct] = wa[(wn * 4 + ct) * 16 + (lane & 15)];
    } else {
      p.wan[0] = wa[(wn * 2 + 0) * 32 + (lane & 31)];
      p.wan[1] = wa[(wn * 2 + 1) * 32 + (lane & 31)];
    }
  }
  return p;
}
template <bool MASK, bool RANK1>
__device__ __forceinline__ void epilogue_dx(const f32x4m (&acc)[4][4], float* Hs, const float* Es_dalpha, const DxPre& pre,
                                            float* __restrict__ dsave, int wm, int wn, int lane, int valid) {
  asm volatile("" : "+v"(lane));
  const unsigned mlo = pre.mlo, mhi = pre.mhi;
  float* colp[4][4];
  h_cols16<4>(Hs, wm, wn, lane, colp);
  const int hq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float wan = pre.wan[ct];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[mt][ct][r];
        if (RANK1) v = fmaf(Es_dalpha[wm * 64 + mt * 16 + 4 * hq + r], wan, v);
        if (MASK) {   // this lane's sign word of the forward (epilogue_fwd, same enumeration)
          const int idx = (mt * 4 + ct) * 4 + r;
          const int keep = __builtin_amdgcn_sbfe(idx < 32 ? (int)mlo : (int)mhi, 31 - (idx & 31), 1);
          v = __int_as_float(__float_as_int(v) & keep);
        }
        H16_AT(colp, ct, mt, r) = v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool MASK, bool RANK1>
__device__ __forceinline__ void epilogue_dx(const f32x16 (&acc)[2][2], float* Hs, const float* Es_dalpha,
                                            const DxPre& pre, float* __restrict__ dsave, int wm, int wn, int lane,
                                            int valid) {
  asm volatile("" : "+v"(lane));
  const unsigned mlo = pre.mlo, mhi = pre.mhi;
  float* colp[2][8];
  h_cols<2>(Hs, wm, wn, lane, colp);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = (wn * 2 + nt) * 32 + (lane & 31);
    const float wan = pre.wan[nt];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 64 + mt * 32 + crow(r, lane);
        float v = acc[mt][nt][r];
        if (RANK1) v = fmaf(Es_dalpha[m], wan, v);
        if (MASK) {   // this lane's sign word of the forward (epilogue_fwd): value idx is bit 31 - (idx & 31) of half idx >> 5
          const int idx = (nt * 2 + mt) * 16 + r;
          const int keep = __builtin_amdgcn_sbfe(idx < 32 ? (int)mlo : (int)mhi, 31 - (idx & 31), 1);   // 0 or -1
          v = __int_as_float(__float_as_int(v) & keep);
        }
        H_AT(colp, nt, mt, r) = v;
        if (dsave != nullptr && m < valid) dsave[(unsigned)(m * 256 + n)] = v;
      }
      __builtin_amdgcn_sched_barrier(0);  // bound live ranges: one 32x32 tile at a time
    }
  }
}

template <int MM = MM_F32>
__global__ void __launch_bounds__(NTHR, 2 * NTHR / 512 * WG_PER_CU)
mlp_bwd_dx_kernel(int64_t P, const float* __restrict__ draw, const float* __restrict__ act,
                  const float* __restrict__ params, const float* __restrict__ packed_t, float* __restrict__ dact,
                  NetLayout lay, unsigned* __restrict__ sched, const int* __restrict__ live_idx,
                  const int* __restrict__ live_cnt) {
  const int64_t PL = P;   // (live-list mode: see the forward kernel)
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;
  float* Es = smem + LDS_H;  // Es[0..127] = dalpha of the tile's rows
  volatile int* sched_word = reinterpret_cast<volatile int*>(Es + 1024);   // tile scheduler word (sched.h): unused part of Es
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t ntiles = (P + TM - 1) / TM;
  stagger_start();

  for (int64_t tile = blockIdx.x; tile < ntiles;) {
    const int64_t p0 = tile * TM;
    const int valid = (int)((P - p0) < TM ? (P - p0) : TM);
    const unsigned long long* maskw =
        reinterpret_cast<const unsigned long long*>(act + act_mask(PL, lay.pe_pad)) + tile * (8 * NWAVES * 64);
    // ---- phase A: dYv = (drgb . Wr) * [hv > 0] -> H[:, 0:128] --------------------------
    {
      const int pm = tid >> 2, pq = tid & 3;
      const bool ok = pm < valid;
      const int64_t pp = ok ? p0 + pm : P - 1;
      const float4 dr = *reinterpret_cast<const float4*>(draw + (live_idx ? (int64_t)live_idx[pp] : pp) * 4);
      if (pq == 0) Es[pm] = ok ? dr.w : 0.f;
      const float* wr = params + lay.RW;
      const float* hv = act + act_hv(PL, lay.pe_pad) + pp * 128;
      float* dyv = dact + dact_yv(PL) + pp * 128;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = pq * 32 + i * 4;
        const float4 h = *reinterpret_cast<const float4*>(hv + k);
        const float4 w0 = *reinterpret_cast<const float4*>(wr + k);
        const float4 w1 = *reinterpret_cast<const float4*>(wr + 128 + k);
        const float4 w2 = *reinterpret_cast<const float4*>(wr + 256 + k);
        float4 o;
        o.x = (h.x > 0.f) ? fmaf(dr.z, w2.x, fmaf(dr.y, w1.x, dr.x * w0.x)) : 0.f;
        o.y = (h.y > 0.f) ? fmaf(dr.z, w2.y, fmaf(dr.y, w1.y, dr.x * w0.y)) : 0.f;
        o.z = (h.z > 0.f) ? fmaf(dr.z, w2.z, fmaf(dr.y, w1.z, dr.x * w0.z)) : 0.f;
        o.w = (h.w > 0.f) ? fmaf(dr.z, w2.w, fmaf(dr.y, w1.w, dr.x * w0.w)) : 0.f;
        if (!ok) o = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Hs + pm * 256 + ((((k >> 2) ^ (pm & 15))) << 2)) = o;
        if (ok) store_nt(dyv + k, o);
      }
    }
    __syncthreads();
    constexpr bool L16 = MM != MM_F32;
    // every 256 x 256 product after the first finds its first weights loaded one layer ahead (gemm_seg16, "chained weights")
    constexpr bool CHAIN = L16;
    AccT<L16, 2> acc;
    WRegsT<L16, 2> wch;
    // ---- dfeat = dYv . Wv[:, :256]  (K = 128) ------------------------------------------
    zero_acc<2>(acc);
    gemm<MM, 2, 0>(acc, Hs, 0, 16, wblock<MM>(packed_t, lay.PB[0]), 16, 0, wn * 2, wm, lane);
    __syncthreads();
    wprefetch(wch, wblock<MM>(packed_t, lay.PB[1]), 32, 0, 32, wn * 2, lane);
    epilogue_dx<false, false>(acc, Hs, Es, dx_preload<false, false, L16>(nullptr, nullptr, wn, lane), nullptr, wm, wn, lane,
                              valid);
    __syncthreads();
    // ---- dY7 = (dfeat . Wf + dalpha x wa) * [h7 > 0] ------------------------------------
    zero_acc<2>(acc);
    {
      const DxPre pre = dx_preload<true, true, L16>(maskw + (7 * NWAVES + wave) * 64, params + lay.AW, wn, lane);
      gemm<MM, 2, 0, CHAIN>(acc, Hs, 0, 32, wblock<MM>(packed_t, lay.PB[1]), 32, 0, wn * 2, wm, lane, 0,
                            dact + dact_feat(PL) + p0 * 256, valid, wave, wch);    // streams dfeat (what it reads) out
      __syncthreads();
      wprefetch(wch, wblock<MM>(packed_t, lay.PB[2]), 32, 0, 32, wn * 2, lane);
      epilogue_dx<true, true>(acc, Hs, Es, pre, nullptr, wm, wn, lane, valid);
    }
    __syncthreads();
    // ---- dY_{l-1} = (dY_l . W_l) * [h_{l-1} > 0],  l = 7..1 ------------------------------
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
      const int64_t off = lay.PB[9 - l];   // PB[2] = L7t ... PB[8] = L1t
      zero_acc<2>(acc);
      const DxPre pre = dx_preload<true, false, L16>(maskw + ((l - 1) * NWAVES + wave) * 64, nullptr, wn, lane);
      gemm<MM, 2, 0, CHAIN>(acc, Hs, 0, 32, wblock<MM>(packed_t, off), 32, 0, wn * 2, wm, lane, 0,
                            dact + dact_y(PL, l) + p0 * 256, valid, wave, wch);    // streams dY_l (what it reads) out
      __syncthreads();
      wprefetch(wch, wblock<MM>(packed_t, lay.PB[l > 1 ? 10 - l : 8]), 32, 0, 32, wn * 2, lane);   // (l == 1: nobody's; a re-read)
      epilogue_dx<true, false>(acc, Hs, Es, pre, nullptr, wm, wn, lane, valid);
      __syncthreads();
    }
    {   // dY0 has no consumer loop: copy it out row-wise
      float* d0 = dact + dact_y(PL, 0) + p0 * 256;
      for (int i = tid; i < TM * 64; i += NTHR) {
        const int m = i >> 6, sl = i & 63;
        if (m < valid)
          store_nt(d0 + m * 256 + ((sl ^ (m & 15)) << 2), *reinterpret_cast<const float4*>(Hs + m * 256 + sl * 4));
      }
    }
    tile = b_next_tile(sched, sched_word, tid);   // closing barrier inside: H is rewritten by the next tile's phase A
  }
  b_sched_exit(sched, tid);
}

template <int MM>
static int launch_dx_t(int grid, hipStream_t st, int64_t P, const float* draw, const float* act, const float* params, const float* packed_bwd,
                       float* dact, const NetLayout& L, const int* live_idx, const int* live_cnt) {
  static bool attr_done = false;
  if (!attr_done) {
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_dx_kernel<MM>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done = true;
  }
  unsigned* sched = b_sched_pair();
  FN_CHECK_ARG(sched != nullptr, "scheduler counters (hipMalloc failed?)");
  hipLaunchKernelGGL(mlp_bwd_dx_kernel<MM>, dim3(grid), dim3(NTHR), LDS_BYTES, st, P, draw, act, params, packed_bwd, dact, L, sched, live_idx, live_cnt);
  FN_LAUNCH_CHECK();
  return 0;
}
int fn_launch_dx(int mm, int grid, hipStream_t st, int64_t P, const float* draw, const float* act, const float* params, const float* packed_bwd,
                 float* dact, const NetLayout& L, const int* live_idx, const int* live_cnt) {
  return mm == MM_X6 ? launch_dx_t<MM_X6>(grid, st, P, draw, act, params, packed_bwd, dact, L, live_idx, live_cnt)
                     : launch_dx_t<MM_F32>(grid, st, P, draw, act, params, packed_bwd, dact, L, live_idx, live_cnt);
}
