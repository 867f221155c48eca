// mlp_bwd_dw.hip -- dW = dY^T X per layer (+ bias column sums, + the alpha head's rank-1 row), the head gradients, the ordered reduction of the
// per-workgroup partials, and the backward entry points (which launch the dX kernel of mlp_bwd_dx.hip first).
#include "mlp_common.h"

int fn_launch_dx(int mm, int grid, hipStream_t st, int64_t P, const float* draw, const float* act, const float* params, const float* packed_bwd,
                 float* dact, const NetLayout& L, const int* live_idx, const int* live_cnt);   // mlp_bwd_dx.hip

// =========================================================================================
// backward: dW = dY^T X  (split over workgroups by point chunk, partials reduced afterwards)
// =========================================================================================
#define DW_MT 32  // points per LDS stage

// WO x WI waves (4 or 8), each wave TO x TI MFMA tiles:  NO = WO*TO*32, KI = WI*TI*32.
// The 256x256 jobs run 8 waves x 128 accumulator registers (two waves per SIMD) so that one wave's
// staging / bias work overlaps the other's MFMAs.
template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1>     // (the fp32-MFMA dW; bf16x6: mlp_bwd_dw6_kernel below)
__global__ void __launch_bounds__(WO * WI * 64, WO * WI / 4)
mlp_bwd_dw_kernel(int64_t P, const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx,
                  const float* __restrict__ draw /*RANK1: dalpha = draw[p*4+3]*/, float* __restrict__ partial_w,
                  float* __restrict__ partial_b, float* __restrict__ partial_r, const int* __restrict__ live_idx,
                  const int* __restrict__ live_cnt) {
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);   // live-list mode: rows 0 .. *live_cnt of dY / X
  constexpr int NO = WO * TO * 32, KI = WI * TI * 32;
  constexpr int NTD = WO * WI * 64;  // threads
  constexpr int STAGE = DW_MT * (NO + KI) + DW_MT;  // floats per LDS stage (+32 dalpha)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave / WI, wi = wave % WI;
  // contiguous chunk of points for this workgroup (multiple of DW_MT)
  const int64_t ntile_all = (P + DW_MT - 1) / DW_MT;
  const int64_t per = (ntile_all + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = blockIdx.x * per;
  int64_t t1 = t0 + per;
  if (t1 > ntile_all) t1 = ntile_all;

  f32x16 acc[TO][TI];
#pragma unroll
  for (int a = 0; a < TO; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f, rsum = 0.f;

  constexpr int YV = (DW_MT * NO / 4 + NTD - 1) / NTD;  // float4 per thread for the dY stage
  constexpr int XV = (DW_MT * KI / 4 + NTD - 1) / NTD;  // float4 per thread for the X stage
  static_assert(DW_MT * NO % 4 == 0 && DW_MT * KI % 4 == 0 && NO <= NTD && KI <= NTD, "stage split");
  float4 ry[YV], rx[XV];
  float rda = 0.f;

  auto load_stage = [&](int64_t t) {
    const int64_t pbase = t * DW_MT;
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int e = (i * NTD + tid) * 4;
      const int m = e / NO, c = e % NO;
      const int64_t p = pbase + m;
      ry[i] = (e < DW_MT * NO && p < P) ? *reinterpret_cast<const float4*>(dY + p * ldy + c)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int e = (i * NTD + tid) * 4;
      const int m = e / KI, c = e % KI;
      const int64_t p = pbase + m;
      rx[i] = (e < DW_MT * KI && p < P) ? *reinterpret_cast<const float4*>(X + p * ldx + c)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (RANK1 && tid < DW_MT) {
      const int64_t p = pbase + tid;
      rda = (p < P) ? draw[(live_idx ? (int64_t)live_idx[p] : p) * 4 + 3] : 0.f;
    }
  };
  auto store_stage = [&](float* st) {
#pragma unroll
    for (int i = 0; i < YV; ++i)
      if ((i * NTD + tid) * 4 < DW_MT * NO) *reinterpret_cast<float4*>(st + (i * NTD + tid) * 4) = ry[i];
#pragma unroll
    for (int i = 0; i < XV; ++i)
      if ((i * NTD + tid) * 4 < DW_MT * KI) *reinterpret_cast<float4*>(st + DW_MT * NO + (i * NTD + tid) * 4) = rx[i];
    if (RANK1 && tid < DW_MT) st[DW_MT * (NO + KI) + tid] = rda;
  };

  if (t0 < t1) {
    load_stage(t0);
    store_stage(smem);
  }
  __syncthreads();
  for (int64_t t = t0; t < t1; ++t) {
    float* cur = smem + ((t - t0) & 1) * STAGE;
    float* nxt = smem + (((t - t0) & 1) ^ 1) * STAGE;
    if (t + 1 < t1) load_stage(t + 1);
    const float* Ys = cur;
    const float* Xs = cur + DW_MT * NO;
    // MFMA over the stage's 32 points, 2 per step; fragments of the next step are fetched from LDS
    // before the current step's MFMAs are issued (ping-pong registers): with one wave per SIMD
    // nothing else hides the ds_read latency.
    {
      float fa0[TO], fb0[TI], fa1[TO], fb1[TI];
      auto ldf = [&](float (&a)[TO], float (&b)[TI], int k2) {
        const int m = k2 * 2 + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TO; ++i) a[i] = Ys[m * NO + (wo * TO + i) * 32 + (lane & 31)];
#pragma unroll
        for (int j = 0; j < TI; ++j) b[j] = Xs[m * KI + (wi * TI + j) * 32 + (lane & 31)];
      };
      auto mm = [&](const float (&a)[TO], const float (&b)[TI]) {
#pragma unroll
        for (int i = 0; i < TO; ++i)
#pragma unroll
          for (int j = 0; j < TI; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
      };
      ldf(fa0, fb0, 0);
#pragma unroll
      for (int k2 = 0; k2 < DW_MT / 2; k2 += 2) {
        // sched_barrier: keep the LDS reads AHEAD of the MFMA block (the machine scheduler otherwise
        // sinks them next to their first use and re-exposes the latency)
        ldf(fa1, fb1, k2 + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (k2 + 2 < DW_MT / 2) ldf(fa0, fb0, k2 + 2);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (BIAS) {
      if (tid < NO) {
#pragma unroll 8
        for (int m = 0; m < DW_MT; ++m) bsum += Ys[m * NO + tid];
      }
    }
    if (RANK1) {
      if (tid < KI) {
        const float* da = cur + DW_MT * (NO + KI);
#pragma unroll 8
        for (int m = 0; m < DW_MT; ++m) rsum = fmaf(da[m], Xs[m * KI + tid], rsum);
      }
    }
    if (t + 1 < t1) store_stage(nxt);
    __syncthreads();
  }
  // write partials
  float* pw = partial_w + (int64_t)blockIdx.x * NO * KI;
#pragma unroll
  for (int i = 0; i < TO; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (wo * TO + i) * 32 + crow(r, lane);
        const int c = (wi * TI + j) * 32 + (lane & 31);
        pw[(int64_t)o * KI + c] = acc[i][j][r];
      }
  if (BIAS && tid < NO) partial_b[(int64_t)blockIdx.x * NO + tid] = bsum;
  if (RANK1 && tid < KI) partial_r[(int64_t)blockIdx.x * KI + tid] = rsum;
}

// ---------------------------------------------------------------------------------------------------------------------
// MM_X6 dW.  On this chip a SIMD issues NOTHING VALU-class while one of its waves streams v_mfma_f32_32x32x16_bf16 back to back
// (tools/micro/coexec_split.hip, profiles/r03_mfma_valu_exclusion.md): the partner wave's VALU work does not hide under the MFMAs,
// it adds to them; only LDS / memory latency overlaps.  The synchronous stage of mlp_bwd_dw_kernel (global -> registers -> fp32 LDS
// -> barrier, then every wave gathers and splits the fragments it multiplies) split every dY tile in the WI waves that share it
// and every X tile in WO waves -- three quarters of that arithmetic was redundant, and all of it was serial with the MFMAs
// (51 % matrix-pipe busy; 66 % with this kernel, profiles/r03_sq_counters.md).  Here
//   * a k-step is 16 points; its (NO + KI) / 32 operand tiles are split ONCE, each by one wave: a lane loads its 8 consecutive
//     points of a channel straight from global memory (32 lanes = one 128-byte line per point) into registers TWO k-steps ahead
//     (counted vmcnt waits: no control flow between a load and its use), splits them (split3_frag) and stores the three bf16
//     pieces as fragment-ordered 1 KiB planes into the split buffer S; the bias column sums / the rank-1 row come from the same
//     registers;
//   * S is double-buffered (2 x 48 KiB for the 256 x 256 jobs): k-step q + 1 is split after k-step q has been multiplied, ONE
//     barrier per k-step; a wave's MFMA operands are three ds_read_b128 per tile;
//   * whole k-steps address with a wave-uniform base + immediates (no per-lane address arithmetic); the k-steps that touch the
//     end of the workgroup's row range clamp their rows and zero them when they are split.
// ---------------------------------------------------------------------------------------------------------------------
// CTI2 > 0: the last CTI2 of the WI * TI input tiles come from a second tensor X2 of width 32 * CTI2 (the view layer's two inputs,
// feature and encoded direction, ride in ONE job: their common dY is read and split once).
// CTO2 > 0: the last CTO2 of the WO * TO output tiles come from a second gradient tensor dY2 of width 32 * CTO2 (layers 0 and 5 both
// multiply the positional encoding: it is read and split once); bias sums are taken over dY only.
// wg / nwg: this workgroup's chunk of the k-steps and the number of chunks (mlp_bwd_dw6_kernel: blockIdx.x / gridDim.x; the trunk
// launch below: blockIdx.x / a function of the point count)
template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1, int CTI2 = 0, int CTO2 = 0>
__device__ __forceinline__ void dw6_body(const int64_t P, const float* __restrict__ dY, const float* __restrict__ X,
                                         const float* __restrict__ draw, float* __restrict__ partial_w, float* __restrict__ partial_b,
                                         float* __restrict__ partial_r, const int* __restrict__ live_idx,
                                         const float* __restrict__ X2, const float* __restrict__ dY2, const int wg, const int nwg) {
  constexpr int NO = WO * TO * 32, KI = WI * TI * 32;
  constexpr int CTO = WO * TO, CTI = WI * TI, NTILE = CTO + CTI;
  constexpr int NW = WO * WI;
  constexpr int TPW = (NTILE + NW - 1) / NW;         // tiles a wave splits per k-step
  extern __shared__ __attribute__((aligned(16))) uint4 S6[];   // [2][tile][piece h | m | l][lane]
  float* const DA = reinterpret_cast<float*>(S6 + 2 * NTILE * 192);   // RANK1: dalpha of the 16 points of a k-step, [2][16]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave / WI, wi = wave % WI;
  const int64_t nq_all = (P + 15) / 16;
  const int64_t per = (nq_all + nwg - 1) / nwg;
  const int64_t q0 = wg * per;
  int64_t q1 = q0 + per;
  if (q1 > nq_all) q1 = nq_all;
  const int nq = (int)(q1 - q0);
  int64_t Pend = q1 * 16;                             // rows this workgroup may read
  if (Pend > P) Pend = P;

  f32x16 acc[TO][TI];
#pragma unroll
  for (int a = 0; a < TO; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float ssum[TPW];   // side sums of the tiles THIS wave splits: bias column sums (dY tiles) / rank-1 row (X tiles)
#pragma unroll
  for (int k = 0; k < TPW; ++k) ssum[k] = 0.f;

  // the tiles this wave splits: tile t = k * NW + wave; t < CTO: channels t*32.. of dY, else channels (t - CTO)*32.. of X.
  // Rows are counted from the workgroup's first row.
  constexpr int CTI1 = CTI - CTI2, KI1 = CTI1 * 32, KI2 = CTI2 * 32;   // row widths of X and X2 (KI = KI1 + KI2 partial columns)
  static_assert(!(RANK1 && CTI2), "the rank-1 row is taken over X only");
  constexpr int CTO1 = CTO - CTO2, NO1 = CTO1 * 32, NO2 = CTO2 * 32;   // row widths of dY and dY2 (NO = NO1 + NO2 partial rows)
  static_assert(!(RANK1 && CTO2) && !(CTI2 && CTO2), "one second tensor per job");
  const float* tsrc[TPW];
  bool tisy[TPW], tis2[TPW];   // tis2: the tile comes from the second tensor of its side
  // KNOWN: the wave count divides the number of dY tiles and there is no second tensor (the 256 x 256 jobs: 14 of a step's 22 launches):
  // whether slot k holds a dY or an X tile is then the same for every wave and known at compile time -- no wave-uniform branch splits the
  // main loop's body, which is what lets the split of k-step q + 1 be scheduled between the MFMAs of k-step q (X6_DW_PIPE below)
  constexpr bool KNOWN = (CTO % NW == 0) && (NTILE % NW == 0) && CTO2 == 0 && CTI2 == 0;
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    const int t = k * NW + wave;
    tisy[k] = KNOWN ? (k * NW < CTO) : (t < CTO);
    tis2[k] = (CTI2 > 0 && t >= CTO + CTI1) || (CTO2 > 0 && t >= CTO1 && t < CTO);
    tsrc[k] = tisy[k] ? (tis2[k] ? dY2 + q0 * (16 * NO2) + (t - CTO1) * 32 : dY + q0 * (16 * NO1) + t * 32)
                      : (tis2[k] ? X2 + q0 * (16 * KI2) + (t - CTO - CTI1) * 32 : X + q0 * (16 * KI1) + (t - CTO) * 32);
  }
  const int relmax = (int)(Pend - q0 * 16) - 1;   // last row of this workgroup (nq > 0: >= 0)
  const int col = lane & 31, half8 = (lane >> 5) * 8;
  const unsigned boffy = (unsigned)(half8 * NO1 + col) * 4u, boffy2 = (unsigned)(half8 * NO2 + col) * 4u,
                 boffx = (unsigned)(half8 * KI1 + col) * 4u, boffx2 = (unsigned)(half8 * KI2 + col) * 4u;   // byte offsets of row 0
  struct Raw { float v[TPW][8]; };
  auto ldb = [](const float* base, unsigned byte_off) __attribute__((always_inline)) -> float {   // uniform base + 32-bit offset
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  // whole = std::true_type: every row of the k-step exists (straight-line code, counted waits); false_type: rows beyond the range
  // are clamped to the last row here and zeroed in split_store (the last k-steps of a workgroup, and the two-ahead loads past them)
  auto load_raw = [&](Raw& r, int st, auto whole) __attribute__((always_inline)) {   // st = k-step of this workgroup
    if constexpr (decltype(whole)::value) {
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int t = k * NW + wave;
        if (NTILE % NW == 0 || t < NTILE) {
          if (tisy[k] && CTO2 > 0 && tis2[k]) {
            const float* b = tsrc[k] + (int64_t)st * (16 * NO2);
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[k][e] = ldb(b, boffy2 + (unsigned)(e * NO2 * 4));
          } else if (tisy[k]) {
            const float* b = tsrc[k] + (int64_t)st * (16 * NO1);
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[k][e] = ldb(b, boffy + (unsigned)(e * NO1 * 4));
          } else if (CTI2 > 0 && tis2[k]) {
            const float* b = tsrc[k] + (int64_t)st * (16 * KI2);
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[k][e] = ldb(b, boffx2 + (unsigned)(e * KI2 * 4));
          } else {
            const float* b = tsrc[k] + (int64_t)st * (16 * KI1);
#pragma unroll
            for (int e = 0; e < 8; ++e) r.v[k][e] = ldb(b, boffx + (unsigned)(e * KI1 * 4));
          }
        }
      }
    } else {
      unsigned row[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { const int rr = st * 16 + half8 + e; row[e] = (unsigned)(rr < relmax ? rr : relmax); }
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int t = k * NW + wave;
        if (NTILE % NW == 0 || t < NTILE) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            r.v[k][e] = ldb(tsrc[k], (row[e] * (unsigned)(tisy[k] ? (tis2[k] ? NO2 : NO1) : (tis2[k] ? KI2 : KI1)) + (unsigned)col) * 4u);
        }
      }
    }
  };
  // RANK1: d(loss)/d(sigma) = draw[p][3], one point per lane 0..15 of wave 0, staged through LDS one k-step ahead of the split that
  // multiplies it.  Two dependent loads in live mode (row -> point -> draw), each issued one pair of k-steps ahead of its use.
  const int* const li = live_idx ? live_idx : reinterpret_cast<const int*>(draw);   // (a valid address when there is no list)
  auto load_ix = [&](int st) __attribute__((always_inline)) -> int64_t {
    const int rr = st * 16 + (lane & 15);
    const int64_t p = q0 * 16 + (rr < relmax ? rr : relmax);
    const int64_t lv = li[p];
    return live_idx ? lv : p;
  };
  auto split_store = [&](const Raw& r, int buf, int st, auto whole) __attribute__((always_inline)) {
    const int nv = relmax + 1 - (st * 16 + half8);         // this lane's rows e < nv exist
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int t = k * NW + wave;
      if (NTILE % NW == 0 || t < NTILE) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = r.v[k][e];
        if constexpr (!decltype(whole)::value) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = e < nv ? v[e] : 0.f;
        }
        uint4 h, m, l;
        uint4* d = S6 + ((buf * NTILE + t) * 3) * 64 + lane;
        split3_frag(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), h, m, l);
        d[0] = h; d[64] = m; d[128] = l;
        if (BIAS && (KNOWN ? (k * NW < CTO) : (t < CTO1))) ssum[k] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        if (RANK1 && (KNOWN ? (k * NW >= CTO) : (t >= CTO))) {
          const float4 d0 = *reinterpret_cast<const float4*>(DA + (st & 1) * 16 + half8);
          const float4 d1 = *reinterpret_cast<const float4*>(DA + (st & 1) * 16 + half8 + 4);
          const float da[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) ssum[k] = fmaf(da[e], v[e], ssum[k]);
        }
      }
    }
  };
  // six-product MFMAs of the k-step held by S[buf]; X tiles two at a time, product-major over the 2 x TO accumulators of the
  // pair: consecutive MFMAs never target the same accumulator (a dependent MFMA waits for its predecessor's full latency)
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    const uint4* Sb = S6 + buf * NTILE * 192 + lane;
    uint4 a[TO][3];
#pragma unroll
    for (int i = 0; i < TO; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[i][pl] = Sb[((wo * TO + i) * 3 + pl) * 64];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // small terms first
    constexpr int JP = TI >= 2 ? 2 : 1;
#pragma unroll
    for (int j0 = 0; j0 < TI; j0 += JP) {
      uint4 b[JP][3];
#pragma unroll
      for (int jj = 0; jj < JP; ++jj)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          if (j0 + jj < TI) b[jj][pl] = Sb[((CTO + wi * TI + j0 + jj) * 3 + pl) * 64];
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int jj = 0; jj < JP; ++jj)
#pragma unroll
          for (int i = 0; i < TO; ++i)
            if (j0 + jj < TI) acc[i][j0 + jj] = mfma_bf16(a[i][PA[t]], b[jj][PB[t]], acc[i][j0 + jj]);
    }
  };
  auto publish = [&]() __attribute__((always_inline)) {   // S pieces written / fragments read: hand the buffers over
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  if (nq > 0) {
    // k-steps in pairs (S[0] / registers r0 hold the even ones); a trailing odd k-step multiplies a stage of zeros
    constexpr std::true_type WHOLE{};
    constexpr std::false_type ANY{};
    Raw r0, r1;
    load_raw(r0, 0, ANY);
    load_raw(r1, 1, ANY);
    // RANK1, wave 0, lanes 0..15: dalpha of the k-steps 2, 4, ... / 3, 5, ... on their way to DA, and the points after them
    float dr0 = 0.f, dr1 = 0.f;
    int64_t ix0 = 0, ix1 = 0;
    const bool da_lane = RANK1 && wave == 0 && lane < 16;
    if (da_lane) {
      DA[lane] = draw[load_ix(0) * 4 + 3];
      DA[16 + lane] = draw[load_ix(1) * 4 + 3];
      dr0 = draw[load_ix(2) * 4 + 3];
      dr1 = draw[load_ix(3) * 4 + 3];
      ix0 = load_ix(4);
      ix1 = load_ix(5);
    }
    if (RANK1) publish();
    split_store(r0, 0, 0, ANY);
    load_raw(r0, 2, ANY);
    publish();
    auto mix = [&](auto sync) __attribute__((always_inline)) {   // (one pipeline per half of the loop body: distinct sync ids)
      if constexpr (KNOWN && !RANK1) interleave6<0, TO * TI * 6, TPW * 4 * 11 + (BIAS ? 8 : 0), decltype(sync)::value>();
    };
    auto pair = [&](int d, auto whole) __attribute__((always_inline)) {
      const int st = 2 * d;
      if (da_lane) { DA[lane] = dr0; dr0 = draw[ix0 * 4 + 3]; ix0 = load_ix(st + 6); }
      multiply(0);
      split_store(r1, 1, st + 1, whole);
      mix(std::integral_constant<int, 1>{});
      load_raw(r1, st + 3, whole);
      publish();
      if (da_lane) { DA[16 + lane] = dr1; dr1 = draw[ix1 * 4 + 3]; ix1 = load_ix(st + 7); }
      multiply(1);
      split_store(r0, 0, st + 2, whole);
      mix(std::integral_constant<int, 2>{});
      load_raw(r0, st + 4, whole);
      publish();
    };
    const int n2 = (nq + 1) / 2;
    const int nwhole = (relmax + 1) / 16;                     // k-steps 0 .. nwhole - 1 have all their rows
    int dmain = nwhole >= 5 ? (nwhole - 3) / 2 : 0;           // pairs whose k-steps up to 2 d + 4 are whole
    if (dmain > n2) dmain = n2;
#pragma unroll 1
    for (int d = 0; d < dmain; ++d) pair(d, WHOLE);
#pragma unroll 1
    for (int d = dmain; d < n2; ++d) pair(d, ANY);
  }
  // write partials (zeros from workgroups without points: reduce_all sums every chunk's region)
  float* pw = partial_w + (int64_t)wg * NO * KI;
#pragma unroll
  for (int i = 0; i < TO; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (wo * TO + i) * 32 + crow(r, lane);
        const int c = (wi * TI + j) * 32 + (lane & 31);
        pw[(int64_t)o * KI + c] = acc[i][j][r];
      }
  if (BIAS || RANK1) {
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int t = k * NW + wave;
      if (NTILE % NW == 0 || t < NTILE) {
        const float sv = ssum[k] + __shfl_xor(ssum[k], 32, 64);
        if (lane < 32) {
          if (BIAS && t < CTO1) partial_b[(int64_t)wg * NO1 + t * 32 + lane] = sv;
          if (RANK1 && t >= CTO) partial_r[(int64_t)wg * KI + (t - CTO) * 32 + lane] = sv;
        }
      }
    }
  }
}

template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1, int CTI2 = 0, int CTO2 = 0>
__global__ void __launch_bounds__(WO * WI * 64, WO * WI / 4)
mlp_bwd_dw6_kernel(int64_t P, const float* __restrict__ dY, const float* __restrict__ X,
                   const float* __restrict__ draw, float* __restrict__ partial_w, float* __restrict__ partial_b,
                   float* __restrict__ partial_r, const int* __restrict__ live_idx, const int* __restrict__ live_cnt,
                   const float* __restrict__ X2 = nullptr, const float* __restrict__ dY2 = nullptr) {
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);
  dw6_body<WO, WI, TO, TI, BIAS, RANK1, CTI2, CTO2>(P, dY, X, draw, partial_w, partial_b, partial_r, live_idx, X2, dY2, (int)blockIdx.x,
                                                    (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The eight 256 x 256 jobs of a pass (L1 .. L7 and the feature layer with the alpha head's rank-1 row) in ONE launch (round 6): grid =
// (chunks, 8 jobs).  The number of split-K chunks per job is a function of the POINT COUNT ALONE (dw_trunk_chunks: an eighth of the chip per
// 256 k-steps of 16 points, between 1/8 of the CUs and all of them), so that
//   * the partials and their reduction shrink with the batch: a 512-ray shard (the reference's configs train at 1024 - 1920 rays,
//     lego.txt:16; an 8-way strong-scaling shard of BASELINE configs[1] is 512) writes 32 chunks per job instead of 256 -- 75 MB per
//     pass through reduce_all instead of 0.7 GB -- while 8 jobs x ncu / 8 chunks still fill every CU with whole rounds of equal work;
//   * the order in which partial sums meet depends on the point count only: the live-list backward (device-side count) chunks exactly like
//     the plain backward of the same points -- bit-identical gradients (tests/test_gpu_compact.py).  The host sizes the grid for
//     the capacity P_total; chunks beyond dw_trunk_chunks(P) exit and are not reduced.
// ---------------------------------------------------------------------------------------------------------------------
#define DW_QMAX 256   // k-steps (of 16 points) a chunk grows to before the next eighth of the chip is added
__host__ __device__ static inline int dw_trunk_chunks(int64_t P, int ncu) {
  const int unit = ncu >= 8 ? ncu / 8 : 1;
  const int64_t nq = (P + 15) / 16;
  int64_t k = (nq + (int64_t)unit * DW_QMAX - 1) / ((int64_t)unit * DW_QMAX);
  k = k < 1 ? 1 : (k > 8 ? 8 : k);
  return unit * (int)k;
}
struct DwTrunk {
  const float* dY[8];
  const float* X[8];
  float* pw[8];
  float* pb[8];
};
__global__ void __launch_bounds__(512, 2)
mlp_bwd_dw6_trunk_kernel(int64_t P, DwTrunk J, const float* __restrict__ draw, float* __restrict__ partial_r,
                         const int* __restrict__ live_idx, const int* __restrict__ live_cnt, int ncu) {
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);
  const int nact = dw_trunk_chunks(P, ncu);
  if ((int)blockIdx.x >= nact) return;
  const int job = (int)blockIdx.y;
  if (job == 7)
    dw6_body<4, 2, 2, 4, true, true>(P, J.dY[7], J.X[7], draw, J.pw[7], J.pb[7], partial_r, live_idx, nullptr, nullptr, (int)blockIdx.x, nact);
  else
    dw6_body<4, 2, 2, 4, true, false>(P, J.dY[job], J.X[job], nullptr, J.pw[job], J.pb[job], nullptr, live_idx, nullptr, nullptr,
                                      (int)blockIdx.x, nact);
}

// rgb head + alpha bias gradients (VALU reduction over points): per-workgroup partials
//   out[wg][0..383] = dWr[c][k], [384..386] = dbr[c], [387] = dba
__global__ void __launch_bounds__(128) head_grads_kernel(int64_t P, const float* __restrict__ draw,
                                                          const float* __restrict__ hv,
                                                          float* __restrict__ partial, const int* __restrict__ live_idx,
                                                          const int* __restrict__ live_cnt) {
  const int k = threadIdx.x;
  if (live_idx) P = *live_cnt;
  auto dptr = [&](int64_t q) { return draw + (live_idx ? (int64_t)live_idx[q] : q) * 4; };
  const int64_t per = (P + gridDim.x - 1) / gridDim.x;
  const int64_t pa = blockIdx.x * per;
  int64_t pb = pa + per;
  if (pb > P) pb = P;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, sb = 0.f;
  int64_t p = pa;
  constexpr int NF = 16;   // independent points in flight (8 waves per CU: latency, not bandwidth, set the pace at 4); same sum order
  for (; p + NF <= pb; p += NF) {
    float4 d[NF];
    float h[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      d[i] = *reinterpret_cast<const float4*>(dptr(p + i));
      h[i] = hv[(p + i) * 128 + k];
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      s0 = fmaf(d[i].x, h[i], s0); s1 = fmaf(d[i].y, h[i], s1); s2 = fmaf(d[i].z, h[i], s2);
      if (k < 4) sb += (k == 0) ? d[i].x : (k == 1) ? d[i].y : (k == 2) ? d[i].z : d[i].w;
    }
  }
  for (; p < pb; ++p) {
    const float4 d = *reinterpret_cast<const float4*>(dptr(p));
    const float h = hv[p * 128 + k];
    s0 = fmaf(d.x, h, s0); s1 = fmaf(d.y, h, s1); s2 = fmaf(d.z, h, s2);
    if (k < 4) sb += (k == 0) ? d.x : (k == 1) ? d.y : (k == 2) ? d.z : d.w;
  }
  float* o = partial + (int64_t)blockIdx.x * 388;
  o[k] = s0; o[128 + k] = s1; o[256 + k] = s2;
  if (k < 4) o[384 + k] = sb;
}

// ---- one launch reduces every job's per-workgroup partials into the flat gradient ------------
struct RedSeg {
  int64_t src;        // offset into the partial buffer
  int64_t wg_stride;  // floats between consecutive workgroups' partials
  int64_t dst;        // offset into the flat gradient
  int nwg, rows, cols, ld, valid_cols;
  int dyn;            // 1: the segment's chunk count is dw_trunk_chunks(point count) (the trunk launch); nwg is its capacity
};
#define MAX_SEGS 32
struct RedTable {
  RedSeg s[MAX_SEGS];
  int n;
};

__global__ void __launch_bounds__(256) reduce_all_kernel(RedTable tab, const float* __restrict__ partial,
                                                          float* __restrict__ grads, int64_t P, const int* __restrict__ live_cnt, int ncu) {
  RedSeg sg = tab.s[blockIdx.y];
  if (sg.dyn) {
    if (live_cnt) P = (int64_t)*live_cnt;
    const int n = dw_trunk_chunks(P, ncu);
    sg.nwg = n < sg.nwg ? n : sg.nwg;
  }
  const int64_t total = (int64_t)sg.rows * sg.cols;
  const float* src = partial + sg.src;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / sg.cols), c = (int)(e % sg.cols);
    if (c >= sg.valid_cols) continue;
    // RS running sums (chunk w goes to sum w mod RS), combined pairwise: a fixed order that depends on the chunk count alone, and RS independent
    // loads in flight per lane -- the kernel is a chain of dependent HBM round trips, not a bandwidth problem (round 6; it was 5 % of the 512-ray step)
    constexpr int RS = 16;      // (32 measured 6x SLOWER: 46 -> 277 us at 512 rays -- the loads of a round no longer issue back to back)
    float a[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) a[i] = 0.f;
    int w = 0;
    for (; w + RS <= sg.nwg; w += RS) {
#pragma unroll
      for (int i = 0; i < RS; ++i) a[i] += src[(int64_t)(w + i) * sg.wg_stride + e];
    }
#pragma unroll
    for (int i = 0; i < RS - 1; ++i)      // (the tail: chunk counts are multiples of ncu / 8, the head-gradient grid of 1024 -- normally empty)
      if (w + i < sg.nwg) a[i] += src[(int64_t)(w + i) * sg.wg_stride + e];
#pragma unroll
    for (int h = RS / 2; h >= 1; h >>= 1)
#pragma unroll
      for (int i = 0; i < h; ++i) a[i] += a[i + h];
    grads[sg.dst + (int64_t)r * sg.ld + c] = a[0];
  }
}

// dW jobs of one net: NO, KI, bias?, rank1?   (KI of the two pe jobs = the layout's pe_pad)
struct DwJobDesc { int NO, KI, bias, rank1; };
static DwJobDesc dw_job(int j, int pe_pad) {
  switch (j) {
    case 0: return {256, pe_pad, 1, 0};     // L0 (pe)
    case 8: return {256, pe_pad, 0, 0};     // L5 (pe part)
    case 9: return {256, 256, 1, 1};        // feature / remap (+ alpha / sigma row)
    case 10: return {128, 256, 1, 0};       // view layer (feature part)
    case 11: return {128, 32, 0, 0};        // view layer (vpe part)
    default: return {256, 256, 1, 0};       // 1..7: L1..L7 (h part)
  }
}
#define HEAD_MAX_WG 1024
static int64_t dw_job_floats(int j, int pe_pad) {
  const DwJobDesc d = dw_job(j, pe_pad);
  return (int64_t)d.NO * d.KI + (d.bias ? d.NO : 0) + (d.rank1 ? d.KI : 0);
}
// regions in the order 0, 8, 1..7, 9, 10, 11 (then the head partials, "job 12"): the pairs that the bf16x6 path runs as ONE job
// (0 + 8: both multiply the positional encoding; 10 + 11: both multiply dYv) are neighbours
static int64_t dw_job_base(int j, int ncu, int pe_pad) {
  static const int order[12] = {0, 8, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11};
  int64_t o = 0;
  for (int i = 0; i < 12; ++i) {
    if (order[i] == j) return o;
    o += dw_job_floats(order[i], pe_pad) * ncu;
  }
  return o;   // j == 12: everything
}
extern "C" int64_t fastnerf_mlp_bwd_partial_floats(void) {
  return dw_job_base(12, num_cus(), 96) + (int64_t)HEAD_MAX_WG * 388;   // sized for the widest layout
}

template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1, int MM = MM_F32, int CTI2 = 0, int CTO2 = 0>
static int launch_dw(int64_t P, const float* dY, int ldy, const float* X, int ldx, const float* draw, float* base,
                     int nwg, hipStream_t st, const int* live_idx = nullptr, const int* live_cnt = nullptr,
                     const float* X2 = nullptr, const float* dY2 = nullptr) {
  constexpr int NO = WO * TO * 32, KI = WI * TI * 32;
  float* pw = base;
  float* pb = base + (int64_t)nwg * NO * KI;
  float* pr = pb + (BIAS ? (int64_t)nwg * (NO - CTO2 * 32) : 0);
  if constexpr (MM == MM_X6) {
    if (ldy != NO - CTO2 * 32 || ldx != KI - CTI2 * 32 || (CTI2 > 0) != (X2 != nullptr) || (CTO2 > 0) != (dY2 != nullptr)) {
      fn::set_error("launch_dw: the bf16x6 dW kernel needs ld == width (and X2 / dY2 exactly when CTI2 / CTO2 > 0)");
      return -1;
    }
    constexpr int lds6 = 2 * (WO * TO + WI * TI) * 3 * 1024 + 128;
    auto kern6 = mlp_bwd_dw6_kernel<WO, WI, TO, TI, BIAS, RANK1, CTI2, CTO2>;
    static bool attr6 = false;
    if (!attr6) {
      FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern6), hipFuncAttributeMaxDynamicSharedMemorySize, lds6));
      attr6 = true;
    }
    hipLaunchKernelGGL(kern6, dim3(nwg), dim3(WO * WI * 64), lds6, st, P, dY, X, draw, pw, pb, pr, live_idx, live_cnt, X2, dY2);
    FN_LAUNCH_CHECK();
    return 0;
  } else {
    static_assert(MM == MM_F32 && CTI2 == 0 && CTO2 == 0, "two-tensor jobs exist for the bf16x6 dW kernel only");
    constexpr int STAGE = DW_MT * (NO + KI) + DW_MT;
    const size_t lds = 2 * STAGE * sizeof(float);
    auto kern = mlp_bwd_dw_kernel<WO, WI, TO, TI, BIAS, RANK1>;
    static bool attr = false;
    if (!attr) {
      FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(WO * WI * 64), lds, st, P, dY, ldy, X, ldx, draw, pw, pb, pr, live_idx, live_cnt);
    FN_LAUNCH_CHECK();
    return 0;
  }
}

static void add_seg(RedTable& T, int64_t src, int64_t wg_stride, int nwg, int rows, int cols, int64_t dst, int ld,
                    int valid_cols, int dyn = 0) {
  RedSeg& s = T.s[T.n++];
  s.src = src; s.wg_stride = wg_stride; s.nwg = nwg; s.rows = rows; s.cols = cols; s.dst = dst; s.ld = ld;
  s.valid_cols = valid_cols; s.dyn = dyn;
}

template <int MM>
static int bwd_launch_t(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                      const float* packed_bwd, float* dact, float* partial, float* grads, const int* live_idx,
                      const int* live_cnt, fn_stream_t stream) {
  constexpr int MW = MM;
  const NetLayout& L = layout_of(kind);
  const int PEP = L.pe_pad;
  hipStream_t st = fn::S(stream);
  const int64_t P = n * S;
  const int64_t ntiles = (P + TM - 1) / TM;
  const int ncu = num_cus();
  int grid = ncu * WG_PER_CU;
  if (ntiles < grid) grid = (int)ntiles;
  if (int rc_dx = fn_launch_dx(MM, grid, st, P, draw, act, params, packed_bwd, dact, L, live_idx, live_cnt)) return rc_dx;

  // ---- dW jobs: every job writes per-chunk partials into its own region ------------------
  // (grids and chunk counts are functions of the point count alone -- one workgroup per CU and a fixed head-gradient grid for the jobs with
  // their own launch, dw_trunk_chunks for the eight 256 x 256 jobs of the bf16x6 trunk launch -- so the order in which partial sums meet
  // depends on the point count only: live-list backward == plain backward of the same points, bit for bit)
  const int nwg = ncu;
  RedTable T;
  T.n = 0;
  int rc;
  const float* a_pe = act + act_pe(P, PEP);
  auto region = [&](int j) { return partial + dw_job_base(j, ncu, PEP); };
  auto segs = [&](int j, int64_t dstW, int ld, int validc, int64_t dstB, int64_t dstR, int dyn = 0) {
    const DwJobDesc d = dw_job(j, PEP);
    const int64_t b = dw_job_base(j, ncu, PEP);
    add_seg(T, b, (int64_t)d.NO * d.KI, nwg, d.NO, d.KI, dstW, ld, validc, dyn);
    int64_t o = b + (int64_t)nwg * d.NO * d.KI;
    if (d.bias) { add_seg(T, o, d.NO, nwg, 1, d.NO, dstB, d.NO, d.NO, dyn); o += (int64_t)nwg * d.NO; }
    if (d.rank1) add_seg(T, o, d.KI, nwg, 1, d.KI, dstR, d.KI, d.KI, dyn);
  };
  // L0 (+ L5's pe part in the same job under MM_X6: one read and one split of the positional encoding for both;
  //     8 waves x (64 outputs x all pe tiles), partials [512][pe] + bias [256] across the neighbouring regions of jobs 0 and 8)
  if constexpr (MW == MM_X6) {
    // (16 waves x one X tile each for the 64-channel encoding: 225 us against 243 for 8 waves x both tiles, 326 for 4 waves -- the job is HBM-latency-bound, more waves in flight win; `tools/ab_kstats.sh`, round 6)
    if (PEP == 64) rc = launch_dw<8, 2, 2, 1, true, false, MW, 0, 8>(P, dact + dact_y(P, 0), 256, a_pe, 64, nullptr, region(0), nwg, st, live_idx, live_cnt, nullptr, dact + dact_y(P, 5));
    else rc = launch_dw<8, 1, 2, 3, true, false, MW, 0, 8>(P, dact + dact_y(P, 0), 256, a_pe, 96, nullptr, region(0), nwg, st, live_idx, live_cnt, nullptr, dact + dact_y(P, 5));
    if (rc) return rc;
    const int64_t b0 = dw_job_base(0, ncu, PEP);
    add_seg(T, b0, (int64_t)512 * PEP, nwg, 256, PEP, L.LW[0], L.in_pe, L.in_pe);
    add_seg(T, b0 + (int64_t)256 * PEP, (int64_t)512 * PEP, nwg, 256, PEP, L.LW[5], 256 + L.in_pe, L.in_pe);
    add_seg(T, b0 + (int64_t)nwg * 512 * PEP, 256, nwg, 1, 256, L.LB[0], 256, 256);
  } else {
    if (PEP == 64) rc = launch_dw<4, 1, 2, 2, true, false, MW>(P, dact + dact_y(P, 0), 256, a_pe, 64, nullptr, region(0), nwg, st, live_idx, live_cnt);
    else rc = launch_dw<4, 1, 2, 3, true, false, MW>(P, dact + dact_y(P, 0), 256, a_pe, 96, nullptr, region(0), nwg, st, live_idx, live_cnt);
    if (rc) return rc;
    segs(0, L.LW[0], L.in_pe, L.in_pe, L.LB[0], 0);
  }
  // L1..L7 (h part); MM_X6: together with the feature layer in the trunk launch below
  if constexpr (MW != MM_X6) {
    for (int l = 1; l < 8; ++l) {
      if ((rc = launch_dw<4, 2, 2, 4, true, false, MW>(P, dact + dact_y(P, l), 256, act + act_h(P, PEP, l - 1), 256, nullptr, region(l), nwg, st, live_idx, live_cnt))) return rc;
      segs(l, L.LW[l] + (l == 5 ? L.in_pe : 0), l == 5 ? 256 + L.in_pe : 256, 256, L.LB[l], 0);
    }
  }
  // L5 pe part (MM_X6: done with L0 above)
  if constexpr (MW != MM_X6) {
    if (PEP == 64) rc = launch_dw<4, 1, 2, 2, false, false, MW>(P, dact + dact_y(P, 5), 256, a_pe, 64, nullptr, region(8), nwg, st, live_idx, live_cnt);
    else rc = launch_dw<4, 1, 2, 3, false, false, MW>(P, dact + dact_y(P, 5), 256, a_pe, 96, nullptr, region(8), nwg, st, live_idx, live_cnt);
    if (rc) return rc;
    segs(8, L.LW[5], 256 + L.in_pe, L.in_pe, 0, 0);
  }
  // feature / remap layer (+bias) with the alpha / sigma head as a rank-1 row
  if constexpr (MW == MM_X6) {
    // the trunk launch: jobs 0..6 = L1..L7, job 7 = the feature layer; chunk w of a job writes where workgroup w of its own launch did
    DwTrunk J;
    for (int l = 1; l <= 8; ++l) {
      const int j = l < 8 ? l : 9;
      J.dY[l - 1] = l < 8 ? dact + dact_y(P, l) : dact + dact_feat(P);
      J.X[l - 1] = act + act_h(P, PEP, l - 1);
      J.pw[l - 1] = region(j);
      J.pb[l - 1] = region(j) + (int64_t)nwg * 256 * 256;
    }
    float* pr = J.pb[7] + (int64_t)nwg * 256;
    constexpr int lds6 = 2 * 16 * 3 * 1024 + 128;
    static bool attr_t = false;
    if (!attr_t) {
      FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_dw6_trunk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds6));
      attr_t = true;
    }
    hipLaunchKernelGGL(mlp_bwd_dw6_trunk_kernel, dim3(dw_trunk_chunks(P, ncu), 8), dim3(512), lds6, st, P, J, draw, pr, live_idx, live_cnt, ncu);
    FN_LAUNCH_CHECK();
    for (int l = 1; l < 8; ++l) segs(l, L.LW[l] + (l == 5 ? L.in_pe : 0), l == 5 ? 256 + L.in_pe : 256, 256, L.LB[l], 0, 1);
    segs(9, L.FW, 256, 256, L.FB, L.AW, 1);
  } else {
    if ((rc = launch_dw<4, 2, 2, 4, true, true, MW>(P, dact + dact_feat(P), 256, act + act_h(P, PEP, 7), 256, draw, region(9), nwg, st, live_idx, live_cnt))) return rc;
    segs(9, L.FW, 256, 256, L.FB, L.AW);
  }
  // view layer
  if constexpr (MW == MM_X6) {
    // one job for both inputs of the view layer (feature [P,256] | encoded direction [P,32]): dYv is read and split once; 12 waves,
    // partials [128][288] + bias [128] across the (adjacent) regions of jobs 10 and 11
    if ((rc = launch_dw<4, 3, 1, 3, true, false, MW, 1, 0>(P, dact + dact_yv(P), 128, act + act_feat(P, PEP), 256, nullptr, region(10), nwg, st,
                                                     live_idx, live_cnt, act + act_vpe(P, PEP), nullptr))) return rc;
    const int64_t b10 = dw_job_base(10, ncu, PEP);
    add_seg(T, b10, 128 * 288, nwg, 128, 288, L.VW, 283, 283);
    add_seg(T, b10 + (int64_t)nwg * 128 * 288, 128, nwg, 1, 128, L.VB, 128, 128);
  } else {
    if ((rc = launch_dw<2, 4, 2, 2, true, false, MW>(P, dact + dact_yv(P), 128, act + act_feat(P, PEP), 256, nullptr, region(10), nwg, st, live_idx, live_cnt))) return rc;
    segs(10, L.VW, 283, 256, L.VB, 0);
    if ((rc = launch_dw<4, 1, 1, 1, false, false, MW>(P, dact + dact_yv(P), 128, act + act_vpe(P, PEP), 32, nullptr, region(11), nwg, st, live_idx, live_cnt))) return rc;
    segs(11, L.VW + 256, 283, 27, 0, 0);
  }
  // rgb head + alpha bias
  {
    const int hg = HEAD_MAX_WG;
    const int64_t hb = dw_job_base(12, ncu, PEP);
    hipLaunchKernelGGL(head_grads_kernel, dim3(hg), dim3(128), 0, st, P, draw, act + act_hv(P, PEP), partial + hb, live_idx, live_cnt);
    FN_LAUNCH_CHECK();
    add_seg(T, hb, 388, hg, 1, 388, L.RW, 388, 387);   // dWr (384) + dbr (3), contiguous in every layout
    add_seg(T, hb + 387, 388, hg, 1, 1, L.AB, 1, 1);   // dba
  }
  hipLaunchKernelGGL(reduce_all_kernel, dim3(256, T.n), dim3(256), 0, st, T, partial, grads, P, live_cnt, ncu);   // (256 x 256 threads: one element of a 256 x 256 segment per lane)
  FN_LAUNCH_CHECK();
  return 0;
}
static int bwd_launch(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                      const float* packed_bwd, float* dact, float* partial, float* grads, const int* live_idx,
                      const int* live_cnt, fn_stream_t stream, int mm = MM_F32) {
  return mm == MM_X6 ? bwd_launch_t<MM_X6>(kind, n, S, draw, act, params, packed_bwd, dact, partial, grads, live_idx, live_cnt, stream)
                     : bwd_launch_t<MM_F32>(kind, n, S, draw, act, params, packed_bwd, dact, partial, grads, live_idx, live_cnt, stream);
}
extern "C" int fastnerf_mlp_bwd_ex(int kind, int64_t n, int S, const float* draw, const float* act,
                                   const float* params, const float* packed_bwd, float* dact, float* partial,
                                   float* grads, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(draw && act && params && packed_bwd && dact && partial && grads, "null pointer");
  return bwd_launch(kind, n, S, draw, act, params, packed_bwd, dact, partial, grads, nullptr, nullptr, stream);
}
// exact-fp32 twin of fastnerf_mlp_bf16_bwd_live
extern "C" int fastnerf_mlp_bwd_live_ex(int kind, int64_t n, int S, const float* draw, const float* act,
                                        const float* params, const float* packed_bwd, float* dact, float* partial,
                                        float* grads, const int32_t* live_idx, const int32_t* live_cnt,
                                        fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(draw && act && params && packed_bwd && dact && partial && grads && live_idx && live_cnt, "null pointer");
  return bwd_launch(kind, n, S, draw, act, params, packed_bwd, dact, partial, grads, live_idx, live_cnt, stream);
}
extern "C" int fastnerf_mlp_bwd(int64_t n, int S, const float* draw, const float* act, const float* params,
                                const float* packed_bwd, float* dact, float* partial, float* grads,
                                fn_stream_t stream) {
  return fastnerf_mlp_bwd_ex(0, n, S, draw, act, params, packed_bwd, dact, partial, grads, stream);
}

// ---- MM_X6 ("bf16x6") entry points: the call protocol of fastnerf_mlp_{fwd,bwd}_ex / _live_ex / _flags_ex, weights from
// fastnerf_mlp_x6_pack; saved activations and gradient workspaces have the exact-fp32 kernels' layouts and sizes.
extern "C" int fastnerf_mlp_x6_bwd(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                                   const float* packed_bwd, float* dact, float* partial, float* grads, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(draw && act && params && packed_bwd && dact && partial && grads, "null pointer");
  return bwd_launch(kind, n, S, draw, act, params, packed_bwd, dact, partial, grads, nullptr, nullptr, stream, MM_X6);
}
extern "C" int fastnerf_mlp_x6_bwd_live(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                                        const float* packed_bwd, float* dact, float* partial, float* grads,
                                        const int32_t* live_idx, const int32_t* live_cnt, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(draw && act && params && packed_bwd && dact && partial && grads && live_idx && live_cnt, "null pointer");
  return bwd_launch(kind, n, S, draw, act, params, packed_bwd, dact, partial, grads, live_idx, live_cnt, stream, MM_X6);
}
