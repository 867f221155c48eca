// mlp_bf16.hip -- split-bf16 ("bf16x3") implementation of the fused 8x256 MLP (forward, dX chain, dW) for gfx950.
//
// Every fp32 operand x is carried as a pair of bf16 values (hi = bf16(x), lo = bf16(x - hi)); a product
// a*b is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation on v_mfma_f32_32x32x16_bf16 (three
// 32-cycle K=16 instructions instead of eight 64-cycle K=2 fp32 instructions: 5.3x the matrix rate).  The
// dropped lo*lo term and the residual of the two-term split are ~2^-17 relative per product; measured against
// the fp32-MFMA kernels of mlp_fwd / mlp_bwd_*.hip this moves raw network outputs by ~1e-6 relative and the rendered RGB by
// 5e-7 (tools/bf16x3_study.py), i.e. fp32 rounding class -- two orders below the 1e-4 parity tolerance.
//
// Same network functions as mlp_fwd.hip (nerf-ours/model.py:37-63, nerf++-ours/nerf_network.py:70-142); kinds 0/1 and the
// nerf++ background net (kind 2: 4-D inverted-sphere input, 84-channel encoding).
//
// Tiling: 64-point tiles, two 256-thread workgroups per CU, wave = 64x64 output block (2x2 MFMA tiles).
// LDS per workgroup: H as two bf16 planes [64][256] (hi, lo: 32 KiB each) + E planes [64][64] (8 KiB each).
// Output columns are interleaved between a wave's two MFMA column tiles (n = wn*64 + 2*j + nt) so that a lane
// owns two ADJACENT outputs and the epilogue writes them as one packed 32-bit LDS store per plane.
//
// Saved tensors (activations for dW, pre-activation gradients) are written straight from the accumulator
// registers in "K-fragment order": the MFMA C layout gives a lane 4 consecutive POINTS of one channel, which
// is half of the 8-point x 1-channel 16-byte element an A/B operand of the dW GEMM (K = points) wants.  Layout
// of a tensor with C channels (CT = C/32 channel tiles), in 16-byte units:
//     (((tile*CT + ct)*4 + ks)*2 + part)*64 + kb*32 + c        part: 0 hi / 1 lo;  points tile*64+ks*16+kb*8+0..7
// so the dW kernel loads every operand fragment as one contiguous 1 KiB wave access with no LDS transpose.
// 256-channel tensors are stored in the wave-permuted channel order q = (n>>6)*64 + (n&1)*32 + ((n&63)>>1)
// (then every store instruction writes 512 contiguous bytes); the final reduction un-permutes.
#include <stdlib.h>
#include "common.h"
#include "sched.h"
#include "mlp_layout.h"

using namespace fnl;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

// Bisection hook for the SLP-vectoriser corruption (tools/slp_bisect.py): what is executed at every
// k-loop exit, in front of the epilogue.  0: nothing (product); 1: 34 idle wait states (drains the matrix pipe);
// 2: s_waitcnt vmcnt(0) lgkmcnt(0) (drains every outstanding load, incl. the pre-loaded bias / mask words)
__device__ __forceinline__ void bdbg_drain() {
}
#define BTM 64
#define BNTHR 256
#define BLDS_BYTES (2 * BTM * 256 * 2 + 2 * BTM * 64 * 2)   // 81920

static int b_num_cus() {
  static int n = 0;
  if (n > 0) return n;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
  if (n <= 0) n = 256;
  return n;
}

__device__ __forceinline__ unsigned bf16_rne(float v) {   // bits of bf16(v), round to nearest even
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void split2(float v, unsigned& hi, unsigned& lo) {
  hi = bf16_rne(v);
  lo = bf16_rne(v - __uint_as_float(hi << 16));
}
__device__ __forceinline__ void unpk8(const uint4& h, const uint4& l, float (&o)[8]) {
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    o[2 * q] = __uint_as_float(hw[q] << 16) + __uint_as_float(lw[q] << 16);
    o[2 * q + 1] = __uint_as_float(hw[q] & 0xffff0000u) + __uint_as_float(lw[q] & 0xffff0000u);
  }
}

// ---- saved-tensor layouts (16-byte units) -----------------------------------------------------------
// activations: h0..h7 (8 ct each) | feat (8) | vpe (1) | hv (4) | ReLU sign bits h0..h7 | sign bits hv | pe (2 or 3 ct)
__host__ __device__ inline int64_t ba_h(int64_t nt, int l) { return (int64_t)l * nt * 4096; }
__host__ __device__ inline int64_t ba_feat(int64_t nt) { return 8 * nt * 4096; }
__host__ __device__ inline int64_t ba_vpe(int64_t nt) { return 9 * nt * 4096; }
__host__ __device__ inline int64_t ba_hv(int64_t nt) { return ba_vpe(nt) + nt * 512; }
__host__ __device__ inline int64_t ba_mask(int64_t nt) { return ba_hv(nt) + nt * 2048; }    // 1024 units / tile
__host__ __device__ inline int64_t ba_maskv(int64_t nt) { return ba_mask(nt) + nt * 1024; }  // 64 units / tile
__host__ __device__ inline int64_t ba_pe(int64_t nt) { return ba_maskv(nt) + nt * 64; }      // pe_pad/32 tiles of 512 units, last
__host__ __device__ inline int64_t ba_total(int64_t nt, int pe_pad) { return ba_pe(nt) + nt * (pe_pad / 32) * 512; }
// pre-activation gradients: dY0..dY7 (8 ct each) | dfeat (8) | dYv (4) | dalpha (fp32, 64 per tile)
__host__ __device__ inline int64_t bd_y(int64_t nt, int l) { return (int64_t)l * nt * 4096; }
__host__ __device__ inline int64_t bd_feat(int64_t nt) { return 8 * nt * 4096; }
__host__ __device__ inline int64_t bd_yv(int64_t nt) { return 9 * nt * 4096; }
__host__ __device__ inline int64_t bd_alpha(int64_t nt) { return 9 * nt * 4096 + nt * 2048; }
__host__ __device__ inline int64_t bd_total(int64_t nt) { return bd_alpha(nt) + nt * 16; }

// ---- packed bf16 weights --------------------------------------------------------------------------
// dst (uint4 units): ((nt_g*KS + ks)*2 + part)*64 + lane  ->  8 bf16 = B[k = ks*16 + (lane>>5)*8 + 0..7][n(nt_g, lane)]
//   N == 256: n = (nt_g>>1)*64 + 2*(lane&31) + (nt_g&1)      N == 128 (view layer): n = nt_g*32 + (lane&31)
//   forward  (trans 0): B[k][n] = W[n][col(k)]   (two input segments, the first padded to segA_pad)
//   backward (trans 1): B[k][n] = W[k][coloff + n] for k < segA_valid
struct BPackDesc {
  int64_t src_off, dst_off;   // floats / uint4
  int ld, N, Kp, segA_pad, segA_valid, segB_valid, trans, coloff;
};
struct BPackTable { BPackDesc d[19]; };

__global__ void __launch_bounds__(256) bpack_kernel(BPackTable tab, const float* __restrict__ params,
                                                     uint4* __restrict__ dst_fwd, uint4* __restrict__ dst_bwd) {
  const BPackDesc d = tab.d[blockIdx.y];
  const int KS = d.Kp / 16;
  const int64_t total = (int64_t)(d.N / 32) * KS * 2 * 64;
  uint4* dst = (d.trans ? dst_bwd : dst_fwd) + d.dst_off;
  const float* src = params + d.src_off;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(e & 63);
    const int part = (int)((e >> 6) & 1);
    const int64_t blk = e >> 7;
    const int nt = (int)(blk / KS), ks = (int)(blk % KS);
    const int n = (d.N == 256) ? ((nt >> 1) * 64 + 2 * (l & 31) + (nt & 1)) : (nt * 32 + (l & 31));
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kp = ks * 16 + (l >> 5) * 8 + j;
      float v = 0.f;
      if (d.trans) {
        if (kp < d.segA_valid) v = src[(int64_t)kp * d.ld + d.coloff + n];
      } else {
        int col = -1;
        if (kp < d.segA_pad) { if (kp < d.segA_valid) col = kp; }
        else { const int q = kp - d.segA_pad; if (q < d.segB_valid) col = d.segA_valid + q; }
        if (col >= 0) v = src[(int64_t)n * d.ld + col];
      }
      unsigned hi, lo;
      split2(v, hi, lo);
      w[j] = part ? lo : hi;
    }
    uint4 o;
    o.x = w[0] | (w[1] << 16); o.y = w[2] | (w[3] << 16); o.z = w[4] | (w[5] << 16); o.w = w[6] | (w[7] << 16);
    dst[e] = o;
  }
}

// packed offsets in uint4 units; forward ids 0..7 trunk, 8 feature, 9 view; backward ids 0 Vt, 1 Ft, 2..8 = L7t..L1t
struct BOff { int64_t off[10]; int64_t total; };
static BOff b_offsets(const NetLayout& L) {
  BOff o{};
  int64_t p = 0;
  for (int l = 0; l < 10; ++l) {
    o.off[l] = p;
    const int kp = (l == 0) ? L.pe_pad : (l == 5 ? L.pe_pad + 256 : (l == 9 ? 288 : 256));
    const int N = (l == 9) ? 128 : 256;
    p += (int64_t)(N / 32) * (kp / 16) * 2 * 64;
  }
  o.total = p;
  return o;
}
static BOff b_offsets_bwd() {
  BOff o{};
  int64_t p = 0;
  for (int j = 0; j < 9; ++j) {
    o.off[j] = p;
    p += (int64_t)8 * ((j == 0 ? 128 : 256) / 16) * 2 * 64;
  }
  o.off[9] = p;
  o.total = p;
  return o;
}
static const NetLayout& b_layout(int kind) {
  static const NetLayout L[3] = {make_layout(0), make_layout(1), make_layout(2)};
  return L[kind < 0 || kind > 2 ? 0 : kind];
}

extern "C" int64_t fastnerf_mlp_bf16_floats(int kind, int what, int64_t n_points) {
  if (kind < 0 || kind > 2) return -1;
  const int64_t nt = (n_points + BTM - 1) / BTM;
  switch (what) {
    case 1: return b_offsets(b_layout(kind)).total * 4;   // packed forward weights
    case 2: return b_offsets_bwd().total * 4;              // packed backward (transposed) weights
    case 3: return ba_total(nt, b_layout(kind).pe_pad) * 4;                       // saved activations for n_points
    case 4: return bd_total(nt) * 4;                       // pre-activation gradients for n_points
    default: return -1;
  }
}

extern "C" int fastnerf_mlp_bf16_pack(int kind, const float* params, float* packed_fwd, float* packed_bwd,
                                      fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && params && packed_fwd, "kind in 0..2, non-null pointers");
  const NetLayout& L = b_layout(kind);
  const BOff O = b_offsets(L), OB = b_offsets_bwd();
  BPackTable T;
  for (int l = 0; l < 8; ++l) {
    BPackDesc d{};
    d.src_off = L.LW[l]; d.dst_off = O.off[l]; d.N = 256;
    d.ld = (l == 0) ? L.in_pe : (l == 5 ? 256 + L.in_pe : 256);
    d.Kp = (l == 0) ? L.pe_pad : (l == 5 ? L.pe_pad + 256 : 256);
    if (l == 0) { d.segA_pad = L.pe_pad; d.segA_valid = L.in_pe; d.segB_valid = 0; }
    else if (l == 5) { d.segA_pad = L.pe_pad; d.segA_valid = L.in_pe; d.segB_valid = 256; }
    else { d.segA_pad = 256; d.segA_valid = 256; d.segB_valid = 0; }
    T.d[l] = d;
  }
  { BPackDesc d{}; d.src_off = L.FW; d.dst_off = O.off[8]; d.ld = 256; d.N = 256; d.Kp = 256; d.segA_pad = 256; d.segA_valid = 256; T.d[8] = d; }
  { BPackDesc d{}; d.src_off = L.VW; d.dst_off = O.off[9]; d.ld = 283; d.N = 128; d.Kp = 288; d.segA_pad = 256; d.segA_valid = 256; d.segB_valid = 27; T.d[9] = d; }
  int njobs = 10;
  if (packed_bwd) {
    // Vt: dfeat[m][i] = sum_o dYv[m][o] Wv[o][i]
    { BPackDesc d{}; d.trans = 1; d.src_off = L.VW; d.dst_off = OB.off[0]; d.ld = 283; d.N = 256; d.Kp = 128; d.segA_valid = 128; T.d[10] = d; }
    { BPackDesc d{}; d.trans = 1; d.src_off = L.FW; d.dst_off = OB.off[1]; d.ld = 256; d.N = 256; d.Kp = 256; d.segA_valid = 256; T.d[11] = d; }
    for (int l = 7; l >= 1; --l) {
      BPackDesc d{};
      d.trans = 1; d.src_off = L.LW[l]; d.dst_off = OB.off[9 - l]; d.N = 256; d.Kp = 256; d.segA_valid = 256;
      d.ld = (l == 5) ? 256 + L.in_pe : 256;
      d.coloff = (l == 5) ? L.in_pe : 0;
      T.d[12 + (7 - l)] = d;
    }
    njobs = 19;
  }
  hipLaunchKernelGGL(bpack_kernel, dim3(32, njobs), dim3(256), 0, fn::S(stream), T, params,
                     reinterpret_cast<uint4*>(packed_fwd), reinterpret_cast<uint4*>(packed_bwd));
  FN_LAUNCH_CHECK();
  return 0;
}

// ---- LDS addressing (byte offsets inside a plane) ---------------------------------------------------
__device__ __forceinline__ int hoff(int m, int slot) { return m * 512 + ((slot ^ (m & 15)) << 4); }        // 32 slots/row
__device__ __forceinline__ int eoff(int m, int slot) { return m * 128 + ((slot ^ ((m >> 1) & 7)) << 4); }  // 8 slots/row

// nerf++ background: channels 64..95 of the 4-D encoding, two bf16 planes [64][32] that borrow the top 8 KiB of Hhi
#define X2_HI_OFF 24576
#define X2_LO_OFF 28672
__device__ __forceinline__ int x2off(int m, int slot) { return m * 64 + ((slot ^ ((m >> 2) & 3)) << 4); }   // 4 slots/row

__device__ __forceinline__ f32x16 bmfma(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Weight fragments of the first two k-steps of the NEXT k-loop, loaded before the epilogue in front of it: the
// epilogue's global stores (saved tensors) are then YOUNGER than these loads, so the k-loop's first vmcnt waits do
// not have to drain them (vmcnt retires in order; a k-loop that starts behind 16 KiB of stores per wave measured
// +30 %: 11.5 k vs 8.9 k ticks, tools/trace_fwd.py).
template <int NT>
struct BPre { uint4 h0[NT], l0[NT]; };   // k-step 0 (a second prefetched k-step costs 16 VGPRs: scratch spills in dX)
template <int NT>
__device__ __forceinline__ void bprefetch(BPre<NT>& p, const uint4* __restrict__ Bp, int KS, int b_ks0, int nt0, int lane) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const uint4* q = Bp + ((int64_t)(nt0 + nt) * KS + b_ks0) * 128 + lane;
    p.h0[nt] = q[0]; p.l0[nt] = q[64];
  }
  __builtin_amdgcn_sched_barrier(0);
}

// accumulate nks (even) k-steps of 16.  A planes in LDS (H layout or E layout); B packed in global.
// PRE: the fragments of k-steps 0 and 1 are already in *pre (needs nks >= 4).
// AMODE: 0 = H planes, 1 = E planes, 2 = X2 block (pass Ahi = Hhi + X2_HI_OFF, Alo = Hhi + X2_LO_OFF)
template <int NT, int AMODE, bool PRE = false, int PF = 2>
__device__ __forceinline__ void bgemm(f32x16 (&acc)[2][NT], const char* Ahi, const char* Alo, int a_ks0, int nks,
                                      const uint4* __restrict__ Bp, int KS, int b_ks0, int nt0, int lane,
                                      const BPre<NT>* pre = nullptr) {
  asm volatile("" : "+v"(lane));
  const int lrow = lane & 31, kb = lane >> 5;
  // slot ^ swz = (2K + kb) ^ swz = 2K ^ (kb ^ swz): the lane part (ysw, arow) is loop invariant, 2K is uniform, so a
  // k-step's LDS offset costs one v_xor (SGPR operand) + one v_lshl_add; the row tile is an immediate offset
  const int ysw = kb ^ (AMODE == 1 ? ((lrow >> 1) & 7) : (AMODE == 2 ? ((lrow >> 2) & 3) : (lrow & 15)));
  const int arow = lrow * (AMODE == 1 ? 128 : (AMODE == 2 ? 64 : 512));
  auto aoff = [&](int mt, int ks) {
    return arow + (((2 * (a_ks0 + ks)) ^ ysw) << 4) + mt * (32 * (AMODE == 1 ? 128 : (AMODE == 2 ? 64 : 512)));
  };
  // weight fragments through buffer loads: resource = this layer's packed block (uniform), VGPR offset = lane * 16
  // (loop invariant), the (column tile, k-step, plane) part is a scalar offset -> no vector address arithmetic
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Bp), 0, 0x7fffffff, 0x00020000);
  const int bvofs = lane * 16;
  int bsofs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsofs[nt] = ((nt0 + nt) * KS + b_ks0) * 2048;
  auto ldA = [&](uint4 (&ah)[2], uint4 (&al)[2], int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int o = aoff(mt, ks);
      ah[mt] = *reinterpret_cast<const uint4*>(Ahi + o);
      al[mt] = *reinterpret_cast<const uint4*>(Alo + o);
    }
  };
  auto ldB = [&](uint4 (&bh)[NT], uint4 (&bl)[NT], int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
      bh[nt] = __builtin_bit_cast(uint4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(brsrc, bvofs, bsofs[nt] + ks * 2048, 0));
      bl[nt] = __builtin_bit_cast(uint4, (u32x4b)__builtin_amdgcn_raw_buffer_load_b128(brsrc, bvofs, bsofs[nt] + ks * 2048 + 1024, 0));
    }
  };
  auto mm = [&](const uint4 (&ah)[2], const uint4 (&al)[2], const uint4 (&bh)[NT], const uint4 (&bl)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bmfma(ah[mt], bh[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bmfma(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bmfma(al[mt], bh[nt], acc[mt][nt]);
  };
  uint4 ah0[2], al0[2], ah1[2], al1[2], bh0[NT], bl0[NT], bh1[NT], bl1[NT];
  __builtin_amdgcn_s_setprio(1);
  // weights two k-steps ahead (four register sets), activations one ahead; nks % 4 == 0 except the 2-step segments
  if (PF == 2 && nks >= 4) {
    uint4 bh2[NT], bl2[NT], bh3[NT], bl3[NT];
    if (PRE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { bh0[nt] = pre->h0[nt]; bl0[nt] = pre->l0[nt]; }
      ldB(bh1, bl1, 1);
    } else {
      ldB(bh0, bl0, 0); ldB(bh1, bl1, 1);
    }
    ldA(ah0, al0, 0);
#pragma unroll 1
    for (int ks = 0; ks < nks; ks += 4) {
      ldB(bh2, bl2, ks + 2); ldA(ah1, al1, ks + 1);
      mm(ah0, al0, bh0, bl0);
      ldB(bh3, bl3, ks + 3); ldA(ah0, al0, ks + 2);
      mm(ah1, al1, bh1, bl1);
      if (ks + 4 < nks) ldB(bh0, bl0, ks + 4);
      ldA(ah1, al1, ks + 3);
      mm(ah0, al0, bh2, bl2);
      if (ks + 4 < nks) { ldB(bh1, bl1, ks + 5); ldA(ah0, al0, ks + 4); }
      mm(ah1, al1, bh3, bl3);
    }
    __builtin_amdgcn_s_setprio(0);
    bdbg_drain();
    return;
  }
  ldB(bh0, bl0, 0); ldA(ah0, al0, 0);
#pragma unroll 1
  for (int ks = 0; ks < nks; ks += 2) {
    ldB(bh1, bl1, ks + 1); ldA(ah1, al1, ks + 1);
    mm(ah0, al0, bh0, bl0);
    if (ks + 2 < nks) { ldB(bh0, bl0, ks + 2); ldA(ah0, al0, ks + 2); }
    mm(ah1, al1, bh1, bl1);
  }
  __builtin_amdgcn_s_setprio(0);
  bdbg_drain();
}

template <int NT>
__device__ __forceinline__ void bzero(f32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}
// (bzero costs nothing: the compiler folds the zero into the first MFMA of every accumulator as an inline constant;
// starting the accumulators at the bias instead measured slower -- 64 extra v_mov per layer.)
// C layout of the 32x32 MFMAs: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int bcrow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ void gstore8(uint2* p, const uint2& v) {   // 8-byte piece of a saved K-fragment element
  __builtin_nontemporal_store(((u64)v.y << 32) | v.x, reinterpret_cast<u64*>(p));
}

__device__ __forceinline__ void gstore16(uint4* p, const uint4& v) {
  typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
  const u32x4s t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4s*>(p));
}

struct EpiArgs {
  const float* bias;          // BIAS
  const float* dalpha4;       // RANK1: LDS float4 rows, .w = dalpha
  const float* wa;            // RANK1: alpha / sigma weights
  const void* mask_in;        // MASK: the forward's ReLU sign bits of this tile and layer (one word per thread)
  void* mask_out;             // MOUT: where this tile's sign bits go
  uint2* gsave;               // GSAVE: K-fragment tensor, already offset to the tile
  // this lane's bias / rank-1 weights / sign words, loaded by bepi*_preload BEFORE the k-loop so that their
  // latency is not paid at the head of the epilogue
  float r_b0, r_b1, r_wa0, r_wa1;
  uint2 r_min2;
};
template <bool BIAS, bool MASK, bool RANK1>
__device__ __forceinline__ void bepi256_preload(EpiArgs& ea, int wn, int lane) {
  const int n0 = wn * 64 + 2 * (lane & 31);
  if (BIAS) { ea.r_b0 = ea.bias[n0]; ea.r_b1 = ea.bias[n0 + 1]; }
  if (RANK1) { ea.r_wa0 = ea.wa[n0]; ea.r_wa1 = ea.wa[n0 + 1]; }
  if (MASK) ea.r_min2 = reinterpret_cast<const uint2*>(ea.mask_in)[wn * 64 + lane];
}
template <bool BIAS, bool MASK>
__device__ __forceinline__ void bepi128_preload(EpiArgs& ea, int wn, int lane) {
  if (BIAS) ea.r_b0 = ea.bias[wn * 32 + (lane & 31)];
  if (MASK) ea.r_min2.x = reinterpret_cast<const unsigned*>(ea.mask_in)[wn * 64 + lane];
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {   // bf16(a) | bf16(b) << 16, one v_cvt_pk_bf16_f32
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = cvt_pk(a, b);
  lo = cvt_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split1(float a, unsigned& hi, unsigned& lo) {   // low 16 bits of hi / lo
  hi = cvt_pk(a, 0.f);
  lo = cvt_pk(a - __uint_as_float(hi << 16), 0.f);
}
// ReLU sign bits: every thread shifts one bit per value into a private word, most significant = first value.
// v is the value AFTER v_max_f32(v, +0), which never returns -0, so "positive" == "bits != 0".
__device__ __forceinline__ unsigned mask_push(unsigned m, float v) {
  return __builtin_amdgcn_alignbit(m, __float_as_uint(v) + 0x7fffffffu, 31);   // (m << 1) | (v > 0)
}
// k-th pushed value of a full 32-value word sits at bit 31 - k: one signed bit-field extract + one AND
__device__ __forceinline__ float mask_get(unsigned m, int k, float v) {
  const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)m, 31 - k, 1);
  return __uint_as_float(__float_as_uint(v) & keep);
}

// Epilogue of the 256-wide layers.  Lane (j = lane&31, half = lane>>5) owns columns n0 = wn*64 + 2j (+1) of the
// rows mt*32 + bcrow(r, lane).  v = acc (+bias) (+dalpha*wa) (ReLU | forward-mask), split to (hi, lo), then
//   - LDS: the column pair is one packed 32-bit store per plane and row,
//   - global: 4 consecutive rows of one column = 8 bytes of a K-fragment element (see the file header),
//   - sign bits (forward, ReLU layers): 64 per thread and layer, order mt, r, nt.
template <bool BIAS, bool RELU, bool MASK, bool RANK1, bool MOUT, bool GSAVE>
__device__ __forceinline__ void bepi256(const f32x16 (&acc)[2][2], const EpiArgs& ea, char* Hhi, char* Hlo, int wn,
                                        int lane) {
  asm volatile("" : "+v"(lane));
  const int j = lane & 31, half = lane >> 5;
  const int n0 = wn * 64 + 2 * j;
  float b0 = 0.f, b1 = 0.f, wa0 = 0.f, wa1 = 0.f;
  if (BIAS) { b0 = ea.r_b0; b1 = ea.r_b1; }
  if (RANK1) { wa0 = ea.r_wa0; wa1 = ea.r_wa1; }
  uint2 min2 = make_uint2(0u, 0u), mout2 = make_uint2(0u, 0u);
  if (MASK) min2 = ea.r_min2;
  // LDS offset of row m = mt*32 + bcrow(r, lane), columns n0, n0+1:  hoff(m, n0 >> 3) + (n0 & 7) * 2.  With
  // m & 15 = ((r & 3) | ((r >> 2) & 1) << 3) ^ (half << 2) the lane part folds into ONE register,
  //   o = (lanebase ^ (c_r << 4)) + (mt*32 + (r&3) + 8*(r>>2)) * 512,   c_r = (r & 3) | ((r >> 2) & 1) << 3,
  // i.e. 8 distinct v_xor results per call and an immediate offset per store -- no per-lane offset table has to stay
  // live across the k-loops (16 VGPRs that used to push the dX and background-forward kernels into scratch spills).
  const int lanebase = (((n0 >> 3) ^ (half << 2)) << 4) + (n0 & 7) * 2 + half * (4 * 512);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    unsigned H[16], L[16];
    unsigned mi = mt ? min2.y : min2.x, mo = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + bcrow(r, lane);
      float v0 = acc[mt][0][r] + b0, v1 = acc[mt][1][r] + b1;
      if (RANK1) {
        const float da = ea.dalpha4[m * 4 + 3];
        v0 = fmaf(da, wa0, v0);
        v1 = fmaf(da, wa1, v1);
      }
      if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      if (MOUT) { mo = mask_push(mo, v0); mo = mask_push(mo, v1); }
      if (MASK) { v0 = mask_get(mi, 2 * r, v0); v1 = mask_get(mi, 2 * r + 1, v1); }
      split_pair(v0, v1, H[r], L[r]);
      // (rows m and m + 32 share m & 15, so the second row tile is the first one's offset + 32 rows)
      const int o = (lanebase ^ (((r & 3) | (((r >> 2) & 1) << 3)) << 4)) + (mt * 32 + (r & 3) + 8 * (r >> 2)) * 512;
      *reinterpret_cast<unsigned*>(Hhi + o) = H[r];
      *reinterpret_cast<unsigned*>(Hlo + o) = L[r];
    }
    if (mt) mout2.y = mo; else mout2.x = mo;
    if (GSAVE) {
      // 16-byte stores: exchange halves so that lanes 0..31 hold the whole 8-point element of the even k-group
      // and lanes 32..63 that of the odd one (v_permlane32_swap: X.hi <-> Y.lo); one wave store = 1 KiB contiguous
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int ct = wn * 2 + nt, ks = mt * 2 + k2;
          const unsigned sel = nt ? 0x07060302u : 0x05040100u;
          const int ge = 2 * k2, go = 2 * k2 + 1;
          unsigned xe[2][2], xo[2][2];   // [part][word]
          xe[0][0] = __builtin_amdgcn_perm(H[4 * ge + 1], H[4 * ge], sel); xe[0][1] = __builtin_amdgcn_perm(H[4 * ge + 3], H[4 * ge + 2], sel);
          xe[1][0] = __builtin_amdgcn_perm(L[4 * ge + 1], L[4 * ge], sel); xe[1][1] = __builtin_amdgcn_perm(L[4 * ge + 3], L[4 * ge + 2], sel);
          xo[0][0] = __builtin_amdgcn_perm(H[4 * go + 1], H[4 * go], sel); xo[0][1] = __builtin_amdgcn_perm(H[4 * go + 3], H[4 * go + 2], sel);
          xo[1][0] = __builtin_amdgcn_perm(L[4 * go + 1], L[4 * go], sel); xo[1][1] = __builtin_amdgcn_perm(L[4 * go + 3], L[4 * go + 2], sel);
#pragma unroll
          for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
              const auto r = __builtin_amdgcn_permlane32_swap(xe[part][w], xo[part][w], false, false);
              xe[part][w] = r[0]; xo[part][w] = r[1];
            }
            uint4* p4 = reinterpret_cast<uint4*>(ea.gsave) + (((ct * 4 + ks) * 2 + part) * 64 + half * 32 + j);
            gstore16(p4, make_uint4(xe[part][0], xe[part][1], xo[part][0], xo[part][1]));
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MOUT) reinterpret_cast<uint2*>(ea.mask_out)[wn * 64 + lane] = mout2;
}

// Epilogue of the 128-wide view layer / dYv: wave wn owns columns wn*32 + j (natural order, 4 channel tiles);
// a lane converts two consecutive rows at a time (= consecutive points of the K-fragment element).
template <bool BIAS, bool RELU, bool MASK, bool MOUT, bool GSAVE>
__device__ __forceinline__ void bepi128(const f32x16 (&acc)[2][1], const EpiArgs& ea, char* Hhi, char* Hlo, int wn,
                                        int lane) {
  asm volatile("" : "+v"(lane));
  const int j = lane & 31, half = lane >> 5;
  const int n = wn * 32 + j;
  const float bv = BIAS ? ea.r_b0 : 0.f;
  unsigned mi = 0u, mo = 0u;
  if (MASK) mi = ea.r_min2.x;
  const int slot = n >> 3, inslot = (n & 7) * 2;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    unsigned H[8], L[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int m = mt * 32 + bcrow(r, lane);   // rows m, m+1
      float v0 = acc[mt][0][r] + bv, v1 = acc[mt][0][r + 1] + bv;
      if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      if (MOUT) { mo = mask_push(mo, v0); mo = mask_push(mo, v1); }
      if (MASK) { v0 = mask_get(mi, mt * 16 + r, v0); v1 = mask_get(mi, mt * 16 + r + 1, v1); }
      split_pair(v0, v1, H[r >> 1], L[r >> 1]);
      const int o0 = hoff(m, slot) + inslot, o1 = hoff(m + 1, slot) + inslot;
      *reinterpret_cast<unsigned short*>(Hhi + o0) = (unsigned short)H[r >> 1];
      *reinterpret_cast<unsigned short*>(Hhi + o1) = (unsigned short)(H[r >> 1] >> 16);
      *reinterpret_cast<unsigned short*>(Hlo + o0) = (unsigned short)L[r >> 1];
      *reinterpret_cast<unsigned short*>(Hlo + o1) = (unsigned short)(L[r >> 1] >> 16);
    }
    if (GSAVE) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ks = mt * 2 + (g >> 1), kb = g & 1;
        uint2* p = ea.gsave + ((((wn * 4 + ks) * 2) * 64 + kb * 32 + j) * 2 + half);
        gstore8(p, make_uint2(H[2 * g], H[2 * g + 1]));
        gstore8(p + 128, make_uint2(L[2 * g], L[2 * g + 1]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MOUT) reinterpret_cast<unsigned*>(ea.mask_out)[wn * 64 + lane] = mo;
}

// =========================================================================================
// forward
// =========================================================================================
// inverted-sphere background point (x', y', z', 1/r) of nerf++ (ddp_model.py:16-45); same arithmetic as mlp_fwd.hip
__device__ __forceinline__ void b_bg_point(const float* __restrict__ o, const float* __restrict__ d, float depth, float x[4]) {
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

// BG == false: points o + d*z, 63-channel encoding in the E planes.
// BG == true : nerf++ background net: inverted-sphere points (4-D), samples consumed far -> near
//              (ddp_model.py:118-124), 84 channels = 64 in E + 20 (padded to 32) in the X2 block that borrows the
//              top 8 KiB of Hhi while H is free (layer 0) or after layer 5 has consumed h4 (re-encoded from registers).
template <bool SAVE, bool BG>
__global__ void __launch_bounds__(BNTHR, 2)
mlp_fwd_bf16_kernel(int64_t P, int S, const float* __restrict__ rays, const float* __restrict__ zv,
                    const float* __restrict__ params, const uint4* __restrict__ pk, float* __restrict__ raw,
                    uint4* __restrict__ act, NetLayout lay, BOff boff, unsigned* __restrict__ sched,
                    const int* __restrict__ live_idx, const int* __restrict__ live_cnt, int flags) {
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  char* Hhi = bsm;
  char* Hlo = bsm + BTM * 512;
  char* Ehi = bsm + 2 * BTM * 512;
  char* Elo = Ehi + BTM * 128;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Live-list mode (training with exact zero-gradient point compaction): tile-local point j of the launch is point
  // live_idx[j] of the ray batch and the number of points is a DEVICE value (no host round trip); the saved tensors are
  // laid out for the capacity P the caller sized them for (nt_lay), only the first ceil(*live_cnt / 64) tiles exist.
  const int64_t nt_lay = (P + BTM - 1) / BTM;
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);
  const int64_t ntiles = (P + BTM - 1) / BTM;
  u64* maskw_all = SAVE ? reinterpret_cast<u64*>(act + ba_mask(nt_lay)) : nullptr;            // [tile][8][256] u64
  unsigned* maskv_all = SAVE ? reinterpret_cast<unsigned*>(act + ba_maskv(nt_lay)) : nullptr;   // [tile][256] u32

  // scheduler word: the last 8 bytes of the lo plane = columns >= 128 of row 63, which hold stale feature values at the end
  // of a tile (the rgb head reads columns < 128) and are next written by the following tile's layer-0 epilogue, one
  // barrier after every thread has read the word
  volatile int* sched_word = reinterpret_cast<volatile int*>(bsm + 2 * BTM * 512 - 8);
  [[maybe_unused]] int titer = 0;   // (phase trace builds stamp the 5th tile of a workgroup)
  for (int64_t tile = blockIdx.x; tile < ntiles; ++titer) {
    const int64_t p0 = tile * BTM;
    const int valid = (int)((P - p0) < BTM ? (P - p0) : BTM);
    const int pm = tid >> 2, pq = tid & 3;
    int64_t pp = p0 + pm;
    if (pp >= P) pp = P - 1;
    if (live_idx) pp = live_idx[pp];
    const int64_t ray = pp / S;
    const float* rr = rays + ray * 11;
    // store one PE channel at row pm: E planes (c < 64) or the X2 block (c >= 64), and (SAVE, stage) the K-fragment
    // staging copy that aliases the head of H
    auto est = [&](int c, float v, bool stage = true) {
      unsigned h, l;
      split1(v, h, l);
      if (c < 64) {
        const int o = eoff(pm, c >> 3) + (c & 7) * 2;
        *reinterpret_cast<unsigned short*>(Ehi + o) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(Elo + o) = (unsigned short)l;
      } else {
        const int o = x2off(pm, (c - 64) >> 3) + (c & 7) * 2;
        *reinterpret_cast<unsigned short*>(Hhi + X2_HI_OFF + o) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(Hhi + X2_LO_OFF + o) = (unsigned short)l;
      }
      if (SAVE && stage) {
        const int t = (((((c >> 5) * 4 + (pm >> 4)) * 2) * 64) + ((pm >> 3) & 1) * 32 + (c & 31)) * 16 + (pm & 7) * 2;
        *reinterpret_cast<unsigned short*>(bsm + t) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(bsm + t + 1024) = (unsigned short)l;
      }
    };
    float xq = 0.f;   // BG: coordinate pq of this row's 4-D point, kept for the layer-5 re-encode
    auto write_x2 = [&](bool stage) {   // channels 64..95 of the 4-D encoding (dimension pq of row pm)
      est(64 + pq, cosf(fmul(xq, 128.0f)), stage);
      est(68 + pq, sinf(fmul(xq, 256.0f)), stage);
      est(72 + pq, cosf(fmul(xq, 256.0f)), stage);
      est(76 + pq, sinf(fmul(xq, 512.0f)), stage);
      est(80 + pq, cosf(fmul(xq, 512.0f)), stage);
      est(84 + pq, 0.f, stage); est(88 + pq, 0.f, stage); est(92 + pq, 0.f, stage);
    };
    if (!BG) {
      const float zz = zv[pp];
      float x[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = fadd(rr[c], fmul(rr[3 + c], zz));
      if (pq == 0) { est(0, x[0]); est(1, x[1]); est(2, x[2]); est(63, 0.f); }
      for (int jj = pq; jj < 30; jj += 4) {
        const int k = jj / 3, dim = jj - 3 * k;
        const float a = fmul(x[dim], (float)(1 << k));
        float sa_, ca_;
        sincosf(a, &sa_, &ca_);   // one range reduction for the pair
        est(3 + 6 * k + dim, sa_);
        est(6 + 6 * k + dim, ca_);
      }
    } else {
      const int sidx = (int)(pp - ray * S);
      const float zz = zv[ray * S + (S - 1 - sidx)];   // flipped sample order
      float x4[4];
      b_bg_point(rr, rr + 3, zz, x4);
      xq = x4[pq];
      est(pq, xq);
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float a = fmul(xq, (float)(1 << k));
        float sa_, ca_;
        sincosf(a, &sa_, &ca_);
        est(4 + 8 * k + pq, sa_);
        est(8 + 8 * k + pq, ca_);
      }
      est(60 + pq, sinf(fmul(xq, 128.0f)));
      write_x2(true);
    }
    __syncthreads();
    if (SAVE) {   // PE tile in K-fragment order: 16 (24) KiB staged at the head of H
      constexpr int PE_U4 = BG ? 1536 : 1024;
      uint4* dst = act + ba_pe(nt_lay) + tile * PE_U4;
#pragma unroll
      for (int i = 0; i < PE_U4 / 256; ++i) dst[i * 256 + tid] = *reinterpret_cast<const uint4*>(bsm + (i * 256 + tid) * 16);
    }
    f32x16 acc[2][2];
    EpiArgs ea{};
    constexpr bool PRE = false;           // (next layer's first weight fragments ahead of the epilogue: measured -1 % in the forward -- spills; dX keeps it)
    constexpr int KPF = BG ? 1 : 2;       // background net: weight fragments one k-step ahead (two would spill: 256 VGPRs + scratch)
    BPre<2> pre;
    // L0
    ea.bias = params + lay.LB[0];
    bepi256_preload<true, false, false>(ea, wn, lane);
    bzero<2>(acc);
    if (!BG) {
      bgemm<2, 1, false, KPF>(acc, Ehi, Elo, 0, 4, pk + boff.off[0], 4, 0, wn * 2, lane);
    } else {
      bgemm<2, 1, false, KPF>(acc, Ehi, Elo, 0, 4, pk + boff.off[0], 6, 0, wn * 2, lane);
      bgemm<2, 2, false, KPF>(acc, Hhi + X2_HI_OFF, Hhi + X2_LO_OFF, 0, 2, pk + boff.off[0], 6, 4, wn * 2, lane);
    }
    if (SAVE || BG) __syncthreads();   // staging copy / X2 block read out before H is written
    if (SAVE) {
      ea.gsave = reinterpret_cast<uint2*>(act + ba_h(nt_lay, 0) + tile * 4096);
      ea.mask_out = maskw_all + (tile * 8 + 0) * 256;
    }
    if (PRE) bprefetch<2>(pre, pk + boff.off[1], 16, 0, wn * 2, lane);
    bepi256<true, true, false, false, SAVE, SAVE>(acc, ea, Hhi, Hlo, wn, lane);
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
      ea.bias = params + lay.LB[l];
      bepi256_preload<true, false, false>(ea, wn, lane);
      bzero<2>(acc);
      const uint4* B = pk + boff.off[l];
      if (l == 5) {
        if (!BG) {
          bgemm<2, 1, PRE, KPF>(acc, Ehi, Elo, 0, 4, B, 20, 0, wn * 2, lane, &pre);
          bgemm<2, 0, false, KPF>(acc, Hhi, Hlo, 0, 16, B, 20, 4, wn * 2, lane);
        } else {
          bgemm<2, 0, false, KPF>(acc, Hhi, Hlo, 0, 16, B, 22, 6, wn * 2, lane);
          __syncthreads();   // h4 consumed: the top of Hhi becomes the X2 block again
          write_x2(false);
          __syncthreads();
          bgemm<2, 1, false, KPF>(acc, Ehi, Elo, 0, 4, B, 22, 0, wn * 2, lane);
          bgemm<2, 2, false, KPF>(acc, Hhi + X2_HI_OFF, Hhi + X2_LO_OFF, 0, 2, B, 22, 4, wn * 2, lane);
        }
      } else {
        bgemm<2, 0, PRE, KPF>(acc, Hhi, Hlo, 0, 16, B, 16, 0, wn * 2, lane, &pre);
      }
      __syncthreads();
      if (SAVE) {
        ea.gsave = reinterpret_cast<uint2*>(act + ba_h(nt_lay, l) + tile * 4096);
        ea.mask_out = maskw_all + (tile * 8 + l) * 256;
      }
      if (PRE) {   // layer l + 1 (layer 5 starts with its skip-input segment), after layer 7 the feature layer
        const int ln = l + 1;
        bprefetch<2>(pre, pk + boff.off[ln], ln == 5 ? 20 : 16, 0, wn * 2, lane);
      }
      bepi256<true, true, false, false, SAVE, SAVE>(acc, ea, Hhi, Hlo, wn, lane);
      __syncthreads();
    }
    // alpha head + view-direction encoding
    float alpha_val;
    {
      const float* wa = params + lay.AW;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = pq * 64 + i * 8;
        const int o = hoff(pm, k >> 3);
        float h8[8];
        unpk8(*reinterpret_cast<const uint4*>(Hhi + o), *reinterpret_cast<const uint4*>(Hlo + o), h8);
#pragma unroll
        for (int q = 0; q < 8; ++q) s = fmaf(h8[q], wa[k + q], s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      alpha_val = s + params[lay.AB];
      const float v[3] = {rr[8], rr[9], rr[10]};
      unsigned short* gv = SAVE ? reinterpret_cast<unsigned short*>(act + ba_vpe(nt_lay) + tile * 512) : nullptr;
      auto estv = [&](int c, float val) {
        unsigned h, l;
        split1(val, h, l);
        const int o = eoff(pm, c >> 3) + (c & 7) * 2;
        *reinterpret_cast<unsigned short*>(Ehi + o) = (unsigned short)h;
        *reinterpret_cast<unsigned short*>(Elo + o) = (unsigned short)l;
        if (SAVE) {
          const int t = ((((pm >> 4) * 2) * 64) + ((pm >> 3) & 1) * 32 + c) * 8 + (pm & 7);   // 2-byte units
          gv[t] = (unsigned short)h;
          gv[t + 512] = (unsigned short)l;
        }
      };
      if (pq == 0) {
        estv(0, v[0]); estv(1, v[1]); estv(2, v[2]);
#pragma unroll
        for (int c = 27; c < 32; ++c) estv(c, 0.f);
      }
      for (int jj = pq; jj < 12; jj += 4) {
        const int k = jj / 3, dim = jj - 3 * k;
        const float a = fmul(v[dim], (float)(1 << k));
        float sa_, ca_;
        sincosf(a, &sa_, &ca_);
        estv(3 + 6 * k + dim, sa_);
        estv(6 + 6 * k + dim, ca_);
      }
    }
    // FN_FWD_SKIP_DEAD_RGB (inference launches of the NeRF net only): when EVERY sample of the tile has sigma <= 0, every one
    // of them gets alpha = 0 and weight = 0 exactly in the compositing (no sigma noise in this mode: the caller's promise), so
    // their colour logits can reach no output and no gradient -- the feature layer, the view layer and the colour head (17 % of
    // the tile's MACs) are skipped and the logits are written as zeros.  Rays that miss the scene are whole tiles of this kind.
    bool skip_tail = false;
    if (!SAVE && !BG && (flags & 1)) {
      // (no __syncthreads_and: it brings a static LDS word, and 80 KiB + 4 bytes per workgroup means ONE workgroup per CU.)
      // One word per wave in 16 bytes of the encoding plane that are free here: channels 56..63 of row 0 -- the point encoding
      // was consumed by layer 5 and the direction encoding written above occupies channels 0..31.
      const unsigned long long any_live = __ballot((pm < valid) && (alpha_val > 0.f));
      volatile int* slot = reinterpret_cast<volatile int*>(Ehi + eoff(0, 7));
      if (lane == 0) slot[wn] = any_live != 0ull;
      __syncthreads();
      skip_tail = (slot[0] | slot[1] | slot[2] | slot[3]) == 0;
    }
    if (skip_tail) {
      if (pq == 0 && pm < valid && raw) *reinterpret_cast<float4*>(raw + (p0 + pm) * 4) = make_float4(0.f, 0.f, 0.f, alpha_val);
    } else {
    // feature layer (no ReLU)
    ea.bias = params + lay.FB;
    bepi256_preload<true, false, false>(ea, wn, lane);
    bzero<2>(acc);
    bgemm<2, 0, PRE, KPF>(acc, Hhi, Hlo, 0, 16, pk + boff.off[8], 16, 0, wn * 2, lane, &pre);
    __syncthreads();
    if (SAVE) ea.gsave = reinterpret_cast<uint2*>(act + ba_feat(nt_lay) + tile * 4096);
    BPre<1> prev;
    if (PRE) bprefetch<1>(prev, pk + boff.off[9], 18, 0, wn, lane);
    bepi256<true, false, false, false, false, SAVE>(acc, ea, Hhi, Hlo, wn, lane);
    __syncthreads();
    // view layer: [feat256 | vpe32] -> 128, ReLU
    {
      f32x16 av[2][1];
      ea.bias = params + lay.VB;
      bepi128_preload<true, false>(ea, wn, lane);
      bzero<1>(av);
      bgemm<1, 0, PRE, KPF>(av, Hhi, Hlo, 0, 16, pk + boff.off[9], 18, 0, wn, lane, &prev);
      bgemm<1, 1, false, KPF>(av, Ehi, Elo, 0, 2, pk + boff.off[9], 18, 16, wn, lane);
      __syncthreads();
      if (SAVE) {
        ea.gsave = reinterpret_cast<uint2*>(act + ba_hv(nt_lay) + tile * 2048);
        ea.mask_out = maskv_all + tile * 256;
      }
      bepi128<true, true, false, SAVE, SAVE>(av, ea, Hhi, Hlo, wn, lane);
      __syncthreads();
    }
    // rgb head
    {
      const float* wr = params + lay.RW;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = pq * 32 + i * 8;
        const int o = hoff(pm, k >> 3);
        float h8[8];
        unpk8(*reinterpret_cast<const uint4*>(Hhi + o), *reinterpret_cast<const uint4*>(Hlo + o), h8);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          s0 = fmaf(h8[q], wr[k + q], s0);
          s1 = fmaf(h8[q], wr[128 + k + q], s1);
          s2 = fmaf(h8[q], wr[256 + k + q], s2);
        }
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
    tile = b_next_tile(sched, sched_word, tid);   // (contains the tile's closing barrier)
  }
  b_sched_exit(sched, tid);
}

static int b_fwd_launch(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                        const float* packed_fwd, float* raw, float* act, const int* live_idx, const int* live_cnt,
                        fn_stream_t stream, int flags = 0) {
  const NetLayout& lay = b_layout(kind);
  const BOff O = b_offsets(lay);
  const int64_t P = n * S;
  const int64_t ntiles = (P + BTM - 1) / BTM;
  int grid = b_num_cus() * 2;
  if (ntiles < grid) grid = (int)ntiles;
  static bool attr_done = false;
  if (!attr_done) {
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_bf16_kernel<false, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BLDS_BYTES));
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_bf16_kernel<true, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BLDS_BYTES));
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_bf16_kernel<false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BLDS_BYTES));
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_bf16_kernel<true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BLDS_BYTES));
    attr_done = true;
  }
  const uint4* pk = reinterpret_cast<const uint4*>(packed_fwd);
  uint4* a4 = reinterpret_cast<uint4*>(act);
  const dim3 g(grid), b(BNTHR);
  hipStream_t st = fn::S(stream);
  unsigned* sched = b_sched_pair();
  FN_CHECK_ARG(sched != nullptr, "scheduler counters (hipMalloc failed?)");
  if (kind == 2) {
    if (act) hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, true>), g, b, BLDS_BYTES, st, P, S, rays11, z, params, pk, raw, a4, lay, O, sched, live_idx, live_cnt, 0);
    else hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, true>), g, b, BLDS_BYTES, st, P, S, rays11, z, params, pk, raw, a4, lay, O, sched, live_idx, live_cnt, 0);
  } else {
    if (act) hipLaunchKernelGGL((mlp_fwd_bf16_kernel<true, false>), g, b, BLDS_BYTES, st, P, S, rays11, z, params, pk, raw, a4, lay, O, sched, live_idx, live_cnt, 0);
    else hipLaunchKernelGGL((mlp_fwd_bf16_kernel<false, false>), g, b, BLDS_BYTES, st, P, S, rays11, z, params, pk, raw, a4, lay, O, sched, live_idx, live_cnt,
                            (kind == 0 && !live_idx) ? flags : 0);
  }
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_mlp_bf16_fwd(int kind, int64_t n, int S, const float* rays11, const float* z,
                                     const float* params, const float* packed_fwd, float* raw, float* act,
                                     fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n >= 0 && S >= 1, "kind in 0..2, n>=0, S>=1");
  FN_CHECK_ARG(n == 0 || (rays11 && z && params && packed_fwd && raw), "null pointer");
  if (n == 0) return 0;
  return b_fwd_launch(kind, n, S, rays11, z, params, packed_fwd, raw, act, nullptr, nullptr, stream);
}

// Inference forward with options (see fastnerf.h: FN_FWD_SKIP_DEAD_RGB)
extern "C" int fastnerf_mlp_bf16_fwd_flags(int kind, int64_t n, int S, const float* rays11, const float* z,
                                           const float* params, const float* packed_fwd, float* raw, int flags,
                                           fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n >= 0 && S >= 1, "kind in 0..2, n>=0, S>=1");
  FN_CHECK_ARG(n == 0 || (rays11 && z && params && packed_fwd && raw), "null pointer");
  if (n == 0) return 0;
  return b_fwd_launch(kind, n, S, rays11, z, params, packed_fwd, raw, nullptr, nullptr, nullptr, stream, flags);
}

// Training forward over a live-point list (see fastnerf.h): activations of the points live_idx[0 .. *live_cnt) are saved
// in list order; nothing else is written.  act is sized for all n*S points.
extern "C" int fastnerf_mlp_bf16_fwd_live(int kind, int64_t n, int S, const float* rays11, const float* z,
                                          const float* params, const float* packed_fwd, float* act,
                                          const int32_t* live_idx, const int32_t* live_cnt, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(rays11 && z && params && packed_fwd && act && live_idx && live_cnt, "null pointer");
  FN_CHECK_ARG(n * (int64_t)S < ((int64_t)1 << 31), "live lists index points with int32");
  return b_fwd_launch(kind, n, S, rays11, z, params, packed_fwd, nullptr, act, live_idx, live_cnt, stream);
}

// =========================================================================================
// backward: dX chain (pre-activation gradients of every layer, K-fragment order)
// =========================================================================================
__global__ void __launch_bounds__(BNTHR, 2)
mlp_bwd_dx_bf16_kernel(int64_t P, const float* __restrict__ draw, const uint4* __restrict__ act,
                       const float* __restrict__ params, const uint4* __restrict__ pkt, uint4* __restrict__ dact,
                       NetLayout lay, BOff boff, unsigned* __restrict__ sched, const int* __restrict__ live_idx,
                       const int* __restrict__ live_cnt) {
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  char* Hhi = bsm;
  char* Hlo = bsm + BTM * 512;
  float* Dr = reinterpret_cast<float*>(bsm + 2 * BTM * 512);   // [64] float4: drgb, dalpha of the tile's rows
  volatile int* sched_word = reinterpret_cast<volatile int*>(bsm + 2 * BTM * 512 + BTM * 16);   // behind Dr: nobody else's
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t nt_lay = (P + BTM - 1) / BTM;   // (live-list mode: see the forward kernel)
  if (live_idx) P = (int64_t)__builtin_amdgcn_readfirstlane(*live_cnt);
  const int64_t ntiles = (P + BTM - 1) / BTM;
  const u64* maskw_all = reinterpret_cast<const u64*>(act + ba_mask(nt_lay));
  const unsigned* maskv_all = reinterpret_cast<const unsigned*>(act + ba_maskv(nt_lay));

  for (int64_t tile = blockIdx.x; tile < ntiles;) {
    const int64_t p0 = tile * BTM;
    if (tid < BTM) {
      const int64_t p = p0 + tid;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < P) d = *reinterpret_cast<const float4*>(draw + (live_idx ? (int64_t)live_idx[p] : p) * 4);
      *reinterpret_cast<float4*>(Dr + tid * 4) = d;
      reinterpret_cast<float*>(dact + bd_alpha(nt_lay))[p0 + tid] = d.w;   // compact copy for the dW rank-1 row
    }
    __syncthreads();
    EpiArgs ea{};
    constexpr int DXPF = 2;
    constexpr bool PRE = true;            // the next product's first weight fragments are loaded ahead of the epilogue (+1 %)
    BPre<2> pre;
    // ---- dYv = (drgb . Wr) * [hv > 0] -> H[:, 0:128] ---------------------------------------------------
    {
      f32x16 av[2][1];
      const int n = wn * 32 + (lane & 31);
      const float* wr = params + lay.RW;
      const float w0 = wr[n], w1 = wr[128 + n], w2 = wr[256 + n];
      ea.mask_in = maskv_all + tile * 256;
      bepi128_preload<false, true>(ea, wn, lane);
      if (PRE) bprefetch<2>(pre, pkt + boff.off[0], 8, 0, wn * 2, lane);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float4 d = *reinterpret_cast<const float4*>(Dr + (mt * 32 + bcrow(r, lane)) * 4);
          av[mt][0][r] = fmaf(d.z, w2, fmaf(d.y, w1, d.x * w0));
        }
      ea.gsave = reinterpret_cast<uint2*>(dact + bd_yv(nt_lay) + tile * 2048);
      bepi128<false, false, true, false, true>(av, ea, Hhi, Hlo, wn, lane);
    }
    __syncthreads();
    f32x16 acc[2][2];
    // ---- dfeat = dYv . Wv[:, :256]  (K = 128) ----------------------------------------------------------
    bzero<2>(acc);
    bgemm<2, 0, PRE, DXPF>(acc, Hhi, Hlo, 0, 8, pkt + boff.off[0], 8, 0, wn * 2, lane, &pre);
    __syncthreads();
    ea.gsave = reinterpret_cast<uint2*>(dact + bd_feat(nt_lay) + tile * 4096);
    if (PRE) bprefetch<2>(pre, pkt + boff.off[1], 16, 0, wn * 2, lane);
    bepi256<false, false, false, false, false, true>(acc, ea, Hhi, Hlo, wn, lane);
    __syncthreads();
    // ---- dY7 = (dfeat . Wf + dalpha x wa) * [h7 > 0] ---------------------------------------------------
    ea.mask_in = maskw_all + (tile * 8 + 7) * 256;
    ea.dalpha4 = Dr;
    ea.wa = params + lay.AW;
    bepi256_preload<false, true, true>(ea, wn, lane);
    bzero<2>(acc);
    bgemm<2, 0, PRE, DXPF>(acc, Hhi, Hlo, 0, 16, pkt + boff.off[1], 16, 0, wn * 2, lane, &pre);
    __syncthreads();
    ea.gsave = reinterpret_cast<uint2*>(dact + bd_y(nt_lay, 7) + tile * 4096);
    if (PRE) bprefetch<2>(pre, pkt + boff.off[2], 16, 0, wn * 2, lane);
    bepi256<false, false, true, true, false, true>(acc, ea, Hhi, Hlo, wn, lane);
    __syncthreads();
    // ---- dY_{l-1} = (dY_l . W_l[:, h part]) * [h_{l-1} > 0],  l = 7..1 ---------------------------------
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
      ea.mask_in = maskw_all + (tile * 8 + (l - 1)) * 256;
      bepi256_preload<false, true, false>(ea, wn, lane);
      bzero<2>(acc);
      bgemm<2, 0, PRE, DXPF>(acc, Hhi, Hlo, 0, 16, pkt + boff.off[9 - l], 16, 0, wn * 2, lane, &pre);
      __syncthreads();
      ea.gsave = reinterpret_cast<uint2*>(dact + bd_y(nt_lay, l - 1) + tile * 4096);
      if (PRE && l > 1) bprefetch<2>(pre, pkt + boff.off[10 - l], 16, 0, wn * 2, lane);
      bepi256<false, false, true, false, false, true>(acc, ea, Hhi, Hlo, wn, lane);
      if (l > 1) __syncthreads();
    }
    tile = b_next_tile(sched, sched_word, tid);   // (contains the tile's closing barrier)
  }
  b_sched_exit(sched, tid);
}

// =========================================================================================
// backward: dW = dY^T X, split over workgroups by point range; every workgroup writes one fp32 partial (position
// order) and one reduction launch sums them.
// =========================================================================================
// one k-step (16 points) of a dW job from an LDS stage: fragments, 3-term MFMAs, bias / rank-1 side sums
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 bmfma4(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void unpk8v(const u32x4& h, const u32x4& l, float (&o)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    o[2 * q] = __uint_as_float(h[q] << 16) + __uint_as_float(l[q] << 16);
    o[2 * q + 1] = __uint_as_float(h[q] & 0xffff0000u) + __uint_as_float(l[q] & 0xffff0000u);
  }
}
// One k-step of the dW tile product in three pieces, so that a caller can run the LDS fragment reads of step q+1 under the
// MFMAs of step q (DW_PIPE) or all three back to back with the DMA issue between issue and wait.
// The fragment reads are inline asm: for a plain LDS load hipcc inserts s_waitcnt vmcnt(0) while an LDS-DMA is
// pending (it cannot see that the DMA targets another stage), which would drain the prefetch every k-step.
// The hand-over in the caller (counted vmcnt + s_barrier) is what orders these reads behind the DMA of THIS stage.
template <int TO, int TI>
struct DwFrag {
  u32x4 ah[TO], al[TO], xh[TI], xl[TI];
};
template <int WO, int WI, int TO, int TI>
__device__ __forceinline__ void dw_issue_reads(DwFrag<TO, TI>& f, const uint4* st, int wo, int wi, int lane) {
  constexpr int CTO = WO * TO;
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(const char*)st + lane * 16;
#pragma unroll
  for (int i = 0; i < TO; ++i) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.ah[i]) : "v"(lbase + ((wo * TO + i) * 128) * 16));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.al[i]) : "v"(lbase + ((wo * TO + i) * 128 + 64) * 16));
  }
#pragma unroll
  for (int j = 0; j < TI; ++j) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.xh[j]) : "v"(lbase + ((CTO + wi * TI + j) * 128) * 16));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.xl[j]) : "v"(lbase + ((CTO + wi * TI + j) * 128 + 64) * 16));
  }
}
// wait for the reads; tying the registers to the wait keeps every consumer behind it
template <int TO, int TI>
__device__ __forceinline__ void dw_wait_reads(DwFrag<TO, TI>& f) {
#pragma unroll
  for (int i = 0; i < TO; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.ah[i]), "+v"(f.al[i]));
#pragma unroll
  for (int j = 0; j < TI; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.xh[j]), "+v"(f.xl[j]));
}
template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1>
__device__ __forceinline__ void dw_mfma(f32x16 (&acc)[TO][TI], float (&bsum)[TO], float (&rsum)[TI], const DwFrag<TO, TI>& f, int wo,
                                        int wi, const float4& d0, const float4& d1) {
#pragma unroll
  for (int i = 0; i < TO; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) acc[i][j] = bmfma4(f.ah[i], f.xh[j], acc[i][j]);
#pragma unroll
  for (int i = 0; i < TO; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) acc[i][j] = bmfma4(f.ah[i], f.xl[j], acc[i][j]);
#pragma unroll
  for (int i = 0; i < TO; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) acc[i][j] = bmfma4(f.al[i], f.xh[j], acc[i][j]);
  // column sums of dY (bias gradient): ~23 VALU per tile and k-step that nothing hides inside a single wave per SIMD.  The WI
  // waves of a row group hold the same dY tiles: without a rank-1 row to balance against they split the tiles between them
  // (tile i belongs to wave i % WI) instead of leaving all of them to one wave and the others at the barrier.
  constexpr bool SPLITB = BIAS && !RANK1 && WI > 1 && TO % WI == 0;
  if (SPLITB) {
#pragma unroll
    for (int i = 0; i < TO; ++i)
      if (i % WI == wi) {
        float v[8];
        unpk8v(f.ah[i], f.al[i], v);
        bsum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
  } else if (BIAS && wi == wo % WI) {
#pragma unroll
    for (int i = 0; i < TO; ++i) {
      float v[8];
      unpk8v(f.ah[i], f.al[i], v);
      bsum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  if (RANK1 && wo == (wi + 1) % WO) {
    const float da[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
    for (int j = 0; j < TI; ++j) {
      float v[8];
      unpk8v(f.xh[j], f.xl[j], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) rsum[j] = fmaf(da[e], v[e], rsum[j]);
    }
  }
}
// `between` runs after the fragment reads have been ISSUED and before they are waited for: the caller puts the DMA issue of
// a later stage there, so its ~80 issue cycles per piece cover the LDS read latency instead of preceding it.
template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1, class F>
__device__ __forceinline__ void dw_stage_compute(f32x16 (&acc)[TO][TI], float (&bsum)[TO], float (&rsum)[TI], const uint4* st, int wo,
                                                 int wi, int lane, const float4& d0, const float4& d1, F&& between) {
  DwFrag<TO, TI> f;
  dw_issue_reads<WO, WI, TO, TI>(f, st, wo, wi, lane);
  between();
  dw_wait_reads<TO, TI>(f);
  dw_mfma<WO, WI, TO, TI, BIAS, RANK1>(acc, bsum, rsum, f, wo, wi, d0, d1);
}

// Every operand byte is fetched from HBM exactly once per workgroup (a first version that loaded fragments straight
// from global let the two waves that share an operand tile both miss in L2: PMC showed 1.7x the algorithmic fetch
// bytes).  Per k-step of 16 points the (CTO + CTI) operand tiles (2 KiB each: hi + lo plane) are copied by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction) into a ring of three stages, two k-steps ahead of the MFMAs;
// the hand-over is a raw s_barrier behind a counted vmcnt wait so that the newest stage stays in flight across it.
// dW is HBM bound (matrix pipe ~35% busy), so the DMA issue cost is irrelevant here.
template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1>
__global__ void __launch_bounds__(WO * WI * 64, 1)   // (HIP: second argument = minimum waves per SIMD; the LDS ring allows one workgroup per CU)
mlp_bwd_dw_lds_bf16_kernel(int64_t P, int64_t ntiles, const uint4* __restrict__ dY, const uint4* __restrict__ X,
                           const float* __restrict__ dalpha, float* __restrict__ partial_w, float* __restrict__ partial_b,
                           float* __restrict__ partial_r, const int* __restrict__ live_cnt) {
  if (live_cnt) ntiles = ((int64_t)__builtin_amdgcn_readfirstlane(*live_cnt) + BTM - 1) / BTM;   // live-list mode
  constexpr int NO = WO * TO * 32, KI = WI * TI * 32;
  constexpr int CTO = WO * TO, CTI = WI * TI;
  constexpr int NW = WO * WI;
  constexpr int STAGE_U4 = (CTO + CTI) * 128;            // uint4 per k-step stage
  constexpr int NPIECE = 2 * (CTO + CTI);                // 1 KiB pieces per stage
  constexpr int PPW = (NPIECE + NW - 1) / NW;            // pieces per wave
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave / WI, wi = wave % WI;
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t t0 = blockIdx.x * per;
  int64_t t1 = t0 + per;
  if (t1 > ntiles) t1 = ntiles;
  // a workgroup without tiles (short live lists: 390 tiles leave 61 of 256 idle) writes nothing: breduce_kernel sums the
  // partials of the first ceil(ntiles / per) workgroups only
  if (t0 >= ntiles) return;

  f32x16 acc[TO][TI];
#pragma unroll
  for (int a = 0; a < TO; ++a)
#pragma unroll
    for (int b = 0; b < TI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum[TO], rsum[TI];
#pragma unroll
  for (int a = 0; a < TO; ++a) bsum[a] = 0.f;
#pragma unroll
  for (int b = 0; b < TI; ++b) rsum[b] = 0.f;

  const int64_t nq = (t1 > t0) ? (t1 - t0) * 4 : 0;   // k-steps of 16 points
  // piece i of this wave = piece p = i*NW + wave of the stage: operand tile p>>1, plane p&1
  auto dma_stage = [&](int64_t q, int buf) __attribute__((always_inline)) {
    const int64_t tile = t0 + (q >> 2);
    const int ks = (int)(q & 3);
    char* dst = dsm + (size_t)buf * (STAGE_U4 * 16);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = i * NW + wave;
      if (NPIECE % NW == 0 || p < NPIECE) {
        const int ct = p >> 1, half = p & 1;
        const uint4* src = (ct < CTO) ? dY + ((tile * CTO + ct) * 4 + ks) * 128 + half * 64 + lane
                                      : X + ((tile * CTI + (ct - CTO)) * 4 + ks) * 128 + half * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 2);
      }
    }
  };
  if (nq > 0) {   // nq is a multiple of 4 (>= DW_NST - 1)
    const bool full_share = (NPIECE % NW == 0) || (PPW - 1) * NW + wave < NPIECE;   // this wave issues PPW pieces / stage
    // wait until at most n of this wave's newest stages are still in flight
    auto wait_stages = [&](int n) __attribute__((always_inline)) {
      if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (n == 1) { if (full_share) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW - 1) : "memory"); }
      else if (n == 2) { if (full_share) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PPW - 1)) : "memory"); }
      else { if (full_share) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (PPW - 1)) : "memory"); }
    };
    static_assert(3 <= 5 && 3 * PPW < 64, "wait_stages counts up to three stages in flight (vmcnt is 6 bits)");
    // `newer` stages were issued after the one that has to be complete: let min(newer, cap) of them stay in flight
    auto wait_newer = [&](int64_t newer, int cap) __attribute__((always_inline)) {
      if (newer >= cap) wait_stages(cap);
      else if (newer == 2) wait_stages(2);
      else if (newer == 1) wait_stages(1);
      else wait_stages(0);
    };
#pragma unroll
    for (int i = 0; i < 3 - 1; ++i) dma_stage(i, i);
    wait_stages(3 - 2);   // stage 0 landed (own pieces)
    __builtin_amdgcn_s_barrier();
    int buf = 0;
#pragma unroll 1
    for (int64_t q = 0; q < nq; ++q) {
      float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
      if (RANK1 && wo == (wi + 1) % WO) {   // before the DMA issue: its wait must not drain the newest stage
        const float4* dp = reinterpret_cast<const float4*>(dalpha + (t0 + (q >> 2)) * 64 + (q & 3) * 16 + (lane >> 5) * 8);
        d0 = dp[0]; d1 = dp[1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const int bn = (buf >= 1) ? buf - 1 : 3 - 1;   // (buf + DW_NST - 1) % DW_NST: the stage consumed at step q-1
      const int64_t qn = q + 3 - 1;
      dw_stage_compute<WO, WI, TO, TI, BIAS, RANK1>(acc, bsum, rsum, reinterpret_cast<const uint4*>(dsm) + buf * STAGE_U4, wo, wi,
                                                    lane, d0, d1, [&]() __attribute__((always_inline)) { if (qn < nq) dma_stage(qn, bn); });
      // stage q+1 must have landed; the newer ones may stay in flight
      const int64_t newer = (nq - 1 - (q + 1));   // stages issued after q+1 (clamped below)
      wait_newer(newer, 3 - 2);
      __builtin_amdgcn_s_barrier();
      buf = (buf == 3 - 1) ? 0 : buf + 1;
    }
  }
  float* pw = partial_w + (int64_t)blockIdx.x * NO * KI;
#pragma unroll
  for (int i = 0; i < TO; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (wo * TO + i) * 32 + bcrow(r, lane);
        const int c = (wi * TI + j) * 32 + (lane & 31);
        pw[(int64_t)o * KI + c] = acc[i][j][r];
      }
  constexpr bool SPLITB = BIAS && !RANK1 && WI > 1 && TO % WI == 0;   // (see dw_stage_compute)
  if (BIAS && (SPLITB || wi == wo % WI)) {
#pragma unroll
    for (int i = 0; i < TO; ++i) {
      if (SPLITB && i % WI != wi) continue;
      const float s = bsum[i] + __shfl_xor(bsum[i], 32, 64);
      if (lane < 32) partial_b[(int64_t)blockIdx.x * NO + (wo * TO + i) * 32 + lane] = s;
    }
  }
  if (RANK1 && wo == (wi + 1) % WO) {
#pragma unroll
    for (int j = 0; j < TI; ++j) {
      const float s = rsum[j] + __shfl_xor(rsum[j], 32, 64);
      if (lane < 32) partial_r[(int64_t)blockIdx.x * KI + (wi * TI + j) * 32 + lane] = s;
    }
  }
}

// rgb head + alpha bias gradients: out[wg][0..383] = dWr[c][k], [384..386] = dbr[c], [387] = dba
__global__ void __launch_bounds__(128) head_grads_bf16_kernel(int64_t P, int64_t ntiles, const float* __restrict__ draw,
                                                               const uint4* __restrict__ hv,
                                                               float* __restrict__ partial, const int* __restrict__ live_idx,
                                                               const int* __restrict__ live_cnt) {
  const int k = threadIdx.x;
  if (live_idx) { P = *live_cnt; ntiles = (P + BTM - 1) / BTM; }
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int64_t ta = blockIdx.x * per;
  int64_t tb = ta + per;
  if (tb > ntiles) tb = ntiles;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, sb = 0.f;
  for (int64_t t = ta; t < tb; ++t) {
#pragma unroll 2
    for (int c = 0; c < 8; ++c) {   // c = ks*2 + kb: 8 points each
      const uint4* p = hv + ((t * 4 + (k >> 5)) * 4 + (c >> 1)) * 128 + (c & 1) * 32 + (k & 31);
      float h[8];
      unpk8(p[0], p[64], h);
      const int64_t pb = t * 64 + c * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pb + e < P) d = *reinterpret_cast<const float4*>(draw + (live_idx ? (int64_t)live_idx[pb + e] : pb + e) * 4);
        s0 = fmaf(d.x, h[e], s0); s1 = fmaf(d.y, h[e], s1); s2 = fmaf(d.z, h[e], s2);
        if (k < 4) sb += (k == 0) ? d.x : (k == 1) ? d.y : (k == 2) ? d.z : d.w;
      }
    }
  }
  float* o = partial + (int64_t)blockIdx.x * 388;
  o[k] = s0; o[128 + k] = s1; o[256 + k] = s2;
  if (k < 4) o[384 + k] = sb;
}

// ---- one launch reduces every job's per-workgroup partials into the flat gradient ------------
// perm bit 0: rows are in wave-permuted channel order, bit 1: columns are
struct BRedSeg {
  int64_t src, wg_stride, dst;
  int nwg, rows, cols, ld, valid_cols, perm;
  int dw;   // partials of a dW job: only the workgroups that had tiles wrote theirs
};
#define BMAX_SEGS 32
struct BRedTable {
  BRedSeg s[BMAX_SEGS];
  int n;
};
__host__ __device__ inline int b_unperm(int q) { return (q & ~63) + 2 * (q & 31) + ((q >> 5) & 1); }

__global__ void __launch_bounds__(256) breduce_kernel(BRedTable tab, const float* __restrict__ partial,
                                                       float* __restrict__ grads, int64_t ntiles, const int* __restrict__ live_cnt) {
  BRedSeg sg = tab.s[blockIdx.y];
  if (sg.dw) {   // the dW workgroups that had tiles: the first ceil(ntiles / per), per = ceil(ntiles / nwg) (mlp_bwd_dw_lds_bf16_kernel)
    if (live_cnt) ntiles = ((int64_t)*live_cnt + BTM - 1) / BTM;
    const int64_t per = (ntiles + sg.nwg - 1) / sg.nwg;
    sg.nwg = per > 0 ? (int)((ntiles + per - 1) / per) : 0;
  }
  const int64_t total = (int64_t)sg.rows * sg.cols;
  const float* src = partial + sg.src;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / sg.cols), c = (int)(e % sg.cols);
    if (sg.perm & 1) r = b_unperm(r);
    if (sg.perm & 2) c = b_unperm(c);
    if (c >= sg.valid_cols) continue;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = 0;
    for (; w + 8 <= sg.nwg; w += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] += src[(int64_t)(w + i) * sg.wg_stride + e];
    }
    // tail: partial w still goes to running sum w % 8, so that leaving out workgroups that wrote nothing (exact zeros) does not
    // change a single bit of the result
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (w + i < sg.nwg) a[i] += src[(int64_t)(w + i) * sg.wg_stride + e];
    grads[sg.dst + (int64_t)r * sg.ld + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
}

struct BJob { int NO, KI, bias, rank1; };
static BJob b_job(int j, int pe_pad) {
  switch (j) {
    case 0: return {256, pe_pad, 1, 0};     // L0 (pe)
    case 8: return {256, pe_pad, 0, 0};     // L5 (pe part)
    case 9: return {256, 256, 1, 1};    // feature / remap (+ alpha / sigma row)
    case 10: return {128, 256, 1, 0};   // view layer (feature part)
    case 11: return {128, 32, 0, 0};    // view layer (vpe part)
    default: return {256, 256, 1, 0};   // 1..7: L1..L7 (h part)
  }
}
#define BHEAD_MAX_WG 1024
static int64_t b_job_floats(int j, int pe_pad) {
  const BJob d = b_job(j, pe_pad);
  return (int64_t)d.NO * d.KI + (d.bias ? d.NO : 0) + (d.rank1 ? d.KI : 0);
}
static int64_t b_job_base(int j, int ncu, int pe_pad) {
  int64_t o = 0;
  for (int i = 0; i < j; ++i) o += b_job_floats(i, pe_pad) * ncu;
  return o;
}
extern "C" int64_t fastnerf_mlp_bf16_partial_floats(void) {
  return b_job_base(12, b_num_cus(), 96) + (int64_t)BHEAD_MAX_WG * 388;   // sized for the widest layout
}

template <int WO, int WI, int TO, int TI, bool BIAS, bool RANK1>
static int b_launch_dw(int64_t P, int64_t ntiles, const uint4* dY, int CTo, const uint4* X, int CTi, const float* dalpha,
                       float* base, int nwg, hipStream_t st, const int* live_cnt = nullptr) {
  constexpr int NO = WO * TO * 32, KI = WI * TI * 32;
  float* pw = base;
  float* pb = base + (int64_t)nwg * NO * KI;
  float* pr = pb + (BIAS ? (int64_t)nwg * NO : 0);
  FN_CHECK_ARG(CTo == WO * TO && CTi == WI * TI, "dW job shape");
  constexpr int lds = 3 * (WO * TO + WI * TI) * 128 * 16;
  auto kern = mlp_bwd_dw_lds_bf16_kernel<WO, WI, TO, TI, BIAS, RANK1>;
  static bool attr = false;
  if (!attr) {
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(WO * WI * 64), lds, st, P, ntiles, dY, X, dalpha, pw, pb, pr, live_cnt);
  FN_LAUNCH_CHECK();
  return 0;
}

static void b_add_seg(BRedTable& T, int64_t src, int64_t wg_stride, int nwg, int rows, int cols, int64_t dst, int ld,
                      int valid_cols, int perm, int dw = 0) {
  BRedSeg& s = T.s[T.n++];
  s.src = src; s.wg_stride = wg_stride; s.nwg = nwg; s.rows = rows; s.cols = cols; s.dst = dst; s.ld = ld;
  s.valid_cols = valid_cols; s.perm = perm; s.dw = dw;
}

static int b_bwd_launch(int kind, int64_t n, int S, const float* draw, const float* act_f, const float* params,
                        const float* packed_bwd, float* dact_f, float* partial, float* grads, const int* live_idx,
                        const int* live_cnt, fn_stream_t stream) {
  const NetLayout& L = b_layout(kind);
  const BOff OB = b_offsets_bwd();
  hipStream_t st = fn::S(stream);
  const int64_t P = n * S;
  const int64_t nt = (P + BTM - 1) / BTM;
  const int ncu = b_num_cus();
  const uint4* act = reinterpret_cast<const uint4*>(act_f);
  uint4* dact = reinterpret_cast<uint4*>(dact_f);
  int grid = ncu * 2;
  if (nt < grid) grid = (int)nt;
  static bool attr_done = false;
  if (!attr_done) {
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_dx_bf16_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, BLDS_BYTES));
    attr_done = true;
  }
  unsigned* sched = b_sched_pair();
  FN_CHECK_ARG(sched != nullptr, "scheduler counters (hipMalloc failed?)");
  hipLaunchKernelGGL(mlp_bwd_dx_bf16_kernel, dim3(grid), dim3(BNTHR), BLDS_BYTES, st, P, draw, act, params,
                     reinterpret_cast<const uint4*>(packed_bwd), dact, L, OB, sched, live_idx, live_cnt);
  FN_LAUNCH_CHECK();

  int nwg = ncu;
  // (always one workgroup per CU, also for batches of fewer tiles -- idle workgroups write zero partials: the order in
  // which the partial sums meet is then a function of the tile count alone, which makes the live-list backward bit-identical
  // to the plain backward of the same points)
  BRedTable T;
  T.n = 0;
  int rc;
  const int PEP = L.pe_pad;
  auto region = [&](int j) { return partial + b_job_base(j, ncu, PEP); };
  // permW: bit0 rows permuted, bit1 cols permuted
  auto segs = [&](int j, int64_t dstW, int ld, int validc, int permW, int64_t dstB, int64_t dstR) {
    const BJob d = b_job(j, PEP);
    const int64_t b = b_job_base(j, ncu, PEP);
    b_add_seg(T, b, (int64_t)d.NO * d.KI, nwg, d.NO, d.KI, dstW, ld, validc, permW, 1);
    int64_t o = b + (int64_t)nwg * d.NO * d.KI;
    if (d.bias) { b_add_seg(T, o, d.NO, nwg, 1, d.NO, dstB, d.NO, d.NO, (permW & 1) ? 2 : 0, 1); o += (int64_t)nwg * d.NO; }
    if (d.rank1) b_add_seg(T, o, d.KI, nwg, 1, d.KI, dstR, d.KI, d.KI, (permW & 2), 1);
  };
  const uint4* a_pe = act + ba_pe(nt);
  // L0
  if (PEP == 64) rc = b_launch_dw<4, 2, 2, 1, true, false>(P, nt, dact + bd_y(nt, 0), 8, a_pe, 2, nullptr, region(0), nwg, st, live_cnt);
  else rc = b_launch_dw<4, 1, 2, 3, true, false>(P, nt, dact + bd_y(nt, 0), 8, a_pe, 3, nullptr, region(0), nwg, st, live_cnt);
  if (rc) return rc;
  segs(0, L.LW[0], L.in_pe, L.in_pe, 1, L.LB[0], 0);
  // L1..L7 (h part)
  for (int l = 1; l < 8; ++l) {
    if ((rc = b_launch_dw<4, 2, 2, 4, true, false>(P, nt, dact + bd_y(nt, l), 8, act + ba_h(nt, l - 1), 8, nullptr, region(l), nwg, st, live_cnt))) return rc;
    segs(l, L.LW[l] + (l == 5 ? L.in_pe : 0), l == 5 ? 256 + L.in_pe : 256, 256, 3, L.LB[l], 0);
  }
  // L5 pe part
  if (PEP == 64) rc = b_launch_dw<4, 2, 2, 1, false, false>(P, nt, dact + bd_y(nt, 5), 8, a_pe, 2, nullptr, region(8), nwg, st, live_cnt);
  else rc = b_launch_dw<4, 1, 2, 3, false, false>(P, nt, dact + bd_y(nt, 5), 8, a_pe, 3, nullptr, region(8), nwg, st, live_cnt);
  if (rc) return rc;
  segs(8, L.LW[5], 256 + L.in_pe, L.in_pe, 1, 0, 0);
  // feature / remap layer (+bias) with the alpha / sigma head as a rank-1 row
  if ((rc = b_launch_dw<4, 2, 2, 4, true, true>(P, nt, dact + bd_feat(nt), 8, act + ba_h(nt, 7), 8,
                                                 reinterpret_cast<const float*>(dact + bd_alpha(nt)), region(9), nwg, st, live_cnt))) return rc;
  segs(9, L.FW, 256, 256, 3, L.FB, L.AW);
  // view layer
  if ((rc = b_launch_dw<4, 2, 1, 4, true, false>(P, nt, dact + bd_yv(nt), 4, act + ba_feat(nt), 8, nullptr, region(10), nwg, st, live_cnt))) return rc;
  segs(10, L.VW, 283, 256, 2, L.VB, 0);
  if ((rc = b_launch_dw<4, 1, 1, 1, false, false>(P, nt, dact + bd_yv(nt), 4, act + ba_vpe(nt), 1, nullptr, region(11), nwg, st, live_cnt))) return rc;
  segs(11, L.VW + 256, 283, 27, 0, 0, 0);
  // rgb head + alpha bias
  {
    const int hg = BHEAD_MAX_WG;   // fixed, for the same reason as nwg
    const int64_t hb = b_job_base(12, ncu, PEP);
    hipLaunchKernelGGL(head_grads_bf16_kernel, dim3(hg), dim3(128), 0, st, P, nt, draw, act + ba_hv(nt), partial + hb, live_idx, live_cnt);
    FN_LAUNCH_CHECK();
    b_add_seg(T, hb, 388, hg, 1, 388, L.RW, 388, 387, 0);   // dWr (384) + dbr (3), contiguous in every layout
    b_add_seg(T, hb + 387, 388, hg, 1, 1, L.AB, 1, 1, 0);   // dba
  }
  hipLaunchKernelGGL(breduce_kernel, dim3(64, T.n), dim3(256), 0, st, T, partial, grads, nt, live_cnt);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_mlp_bf16_bwd(int kind, int64_t n, int S, const float* draw, const float* act_f,
                                     const float* params, const float* packed_bwd, float* dact_f, float* partial,
                                     float* grads, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(draw && act_f && params && packed_bwd && dact_f && partial && grads, "null pointer");
  return b_bwd_launch(kind, n, S, draw, act_f, params, packed_bwd, dact_f, partial, grads, nullptr, nullptr, stream);
}

// Backward over a live-point list: act holds the activations fastnerf_mlp_bf16_fwd_live saved for live_idx[0 .. *live_cnt),
// draw is the full [n*S, 4] upstream gradient (read through the list).  Points not in the list contribute nothing.
extern "C" int fastnerf_mlp_bf16_bwd_live(int kind, int64_t n, int S, const float* draw, const float* act_f,
                                          const float* params, const float* packed_bwd, float* dact_f, float* partial,
                                          float* grads, const int32_t* live_idx, const int32_t* live_cnt,
                                          fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && n > 0 && S >= 1, "kind in 0..2, n>0, S>=1");
  FN_CHECK_ARG(draw && act_f && params && packed_bwd && dact_f && partial && grads && live_idx && live_cnt, "null pointer");
  return b_bwd_launch(kind, n, S, draw, act_f, params, packed_bwd, dact_f, partial, grads, live_idx, live_cnt, stream);
}
