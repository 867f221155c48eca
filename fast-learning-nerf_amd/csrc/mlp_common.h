// mlp_common.h -- what the translation units of the fp32-width MLP kernels share (mlp_pack.hip, mlp_fwd.hip, mlp_bwd_dx.hip, mlp_bwd_dw.hip):
// math modes, tile constants, the operand split, the tile GEMMs (fp32 MFMA, bf16x6 on the 16 x 16 x 32 shape), epilogue helpers.
// Everything here is a template or a forceinline device function: each kernel's code is generated in the unit that instantiates it.
//
// (file header of the former single unit mlp.hip) the 8x256 NeRF MLP (model.py:8-63) as fused fp32-MFMA kernels for gfx950.
//
//   mlp_fwd     : one persistent workgroup per CU walks tiles of 128 points.  A tile's
//                 activations [128 x 256] fp32 live in LDS (128 KiB, XOR-swizzled so that
//                 ds_read_b128 A-fragments are bank-conflict free) next to its positional
//                 encoding [128 x 64] (32 KiB) -- together exactly the CU's 160 KiB.  Points
//                 are generated on the fly (pts = o + d*z), encoded with accurate sinf/cosf,
//                 and pushed through all layers without touching HBM; only raw [P,4] (and, in
//                 training, the activations backward needs) are written.  8 waves = 2(M) x 4(N),
//                 each wave owns a 64x64 output block = 2x2 tiles of v_mfma_f32_32x32x2_f32
//                 (exact fp32, 157 TF peak).  Weights are streamed L2 -> VGPR in a pre-packed
//                 "fragment order" so that every B load is one contiguous 1 KiB wave access.
//   mlp_bwd_dx  : same structure with transposed weights, chaining dY back through the layers.
//   mlp_bwd_dw  : per layer dW = dY^T X over all points; 4 waves x 256 accumulator registers
//                 hold the whole 256x256 dW of a workgroup's point chunk (split-K over
//                 workgroups, deterministic second-pass reduction).
//
// K is permuted identically for A and B (lanes 0-31 take k0..k0+3, lanes 32-63 k0+4..k0+7 of
// every 8-wide k-step) which is legal because a dot product does not care about summation
// order beyond rounding; parity with the reference is therefore "fp32 rounding class"
// (<=1e-6 relative), not bitwise.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "mlp_layout.h"
#include "sched.h"

using namespace fnl;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------------
// Math modes of this file's kernels (template parameter MM):
//   MM_F32  v_mfma_f32_32x32x2_f32: the products and sums of an fp32 FMA chain (157 TFLOP/s peak).
//   MM_X6   "bf16x6": every fp32 operand x is decomposed EXACTLY into three bf16 pieces x = h + m + l (round-to-nearest at
//           every level: 8 + 8 + 8 significand bits, |m| <= 2^-8 |x|, |l| <= 2^-17 |x|) and a product a*b is evaluated as
//             a_h b_h + (a_h b_m + a_m b_h) + (a_m b_m + a_h b_l + a_l b_h)
//           on the bf16 matrix instructions with fp32 accumulation (forward / dX: v_mfma_f32_16x16x32_bf16, dW: v_mfma_f32_32x32x16_bf16):
//           six products at 16x the fp32 instruction's rate each (2.67x the matrix rate).  The three dropped terms are <= 2^-24 |a b| together, i.e. the product is as
//           accurate as fp32's own rounding of it -- fp32 width, unlike the two-piece split of mlp_bf16.hip (16 bits).
//           Weights are packed once per update as three bf16 planes in fragment order; activations / gradients stay fp32 in
//           LDS and in HBM (same layouts, same bytes as MM_F32) and are split in registers when a fragment is read.
// ---------------------------------------------------------------------------------------------------------------------
#define MM_F32 0
#define MM_X6 1
// (round 4) the MM_X6 forward / dX kernels multiply on v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.  The chip is
// power limited under these kernels and most of an MFMA's register traffic is its accumulator (C in + D out: 128 of ~160 bytes per lane
// for the 32x32x16 shape); the 16x16x32 shape updates a quarter of the accumulator with twice the K: half the accumulator traffic per flop,
// twice the operand traffic.  Bare streams on random data: 2240 against 1990 TFLOP/s (tools/micro/mfma_shapes.hip); the planes prototype's
// hidden layers 0.449 against 0.502 ms (tools/micro/x6_planes_proto3.hip).  profiles/r04_power_limit.md, section 5.

// The ONE build switch of this file.  -DFASTNERF_ABLATION=<bits> makes a TIMING-ONLY library with WRONG RESULTS by design (what does a phase of
// the bf16x6 kernels cost?  profiles/r03_x6_ablations.md, r04_power_limit.md, r04_hbm_side.md): 1 = no split arithmetic (garbage pieces),
// 2 = no saved-tensor store leaves the CU, 4 = positional encoding without sines / cosines, 8 / 16 = only the first one / two weight pieces are
// loaded.  build.py refuses the flag for libfastnerf.so itself: an ablation build only exists as variants/<name>.so (FASTNERF_VARIANT).
#ifndef FASTNERF_ABLATION
#define FASTNERF_ABLATION 0
#endif
constexpr bool ABL_NOSPLIT = (FASTNERF_ABLATION & 1) != 0, ABL_NOSTORE = (FASTNERF_ABLATION & 2) != 0, ABL_NOPE = (FASTNERF_ABLATION & 4) != 0;
constexpr int ABL_WPIECES = (FASTNERF_ABLATION & 8) ? 1 : ((FASTNERF_ABLATION & 16) ? 2 : 3);

#define TM 64          // points per tile (fwd / dx); the code is written for TM = 64 * k, the product is built and tested at 64 only
#define NTHR (TM * 4)  // threads per workgroup: (TM/64) x 4 waves, each a 64x64 output block
#define NWAVES (NTHR / 64)
#define WG_PER_CU (128 / TM)  // two 80 KiB workgroups share a CU's 160 KiB LDS when TM == 64
#define LDS_H (TM * 256)
#define LDS_E (TM * 64)
#define LDS_BYTES ((LDS_H + LDS_E) * 4)

static int g_num_cus = 0;
static inline int num_cus() {
  if (g_num_cus > 0) return g_num_cus;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cus = p.multiProcessorCount;
  if (g_num_cus <= 0) g_num_cus = 256;
  return g_num_cus;
}


static inline const NetLayout& layout_of(int kind) {
  static const NetLayout L[3] = {make_layout(0), make_layout(1), make_layout(2)};
  return L[kind < 0 || kind > 2 ? 0 : kind];
}

// ---------------------------------------------------------------------------------------------------------------------
// operand splits (used by the weight packing and by the kernels)
// ---------------------------------------------------------------------------------------------------------------------
// MM_X6 packing (pack6_kernel): three bf16 planes in the fragment order of v_mfma_f32_16x16x32_bf16 (column tiles of 16, k-steps of 32); uint4 units:
//   dst[((tile*KS32 + ks)*3 + plane)*64 + l] = 8 bf16 = piece `plane` of W'[...][k = ks*32 + (l>>4)*8 + 0..7]
//   fwd (n = tile*16 + (l&15)): W'[n][k] with the same k -> source column mapping as pack_kernel;  bwd: W[k][col0 + tile*16 + (l&15)]
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // bf16(a) | bf16(b) << 16 (round to nearest even)
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
// x0, x1 -> packed pieces (h, m, l); exact: x = h + m + l.  Subtract form: v_cvt_pk_bf16_f32 for a piece, shift / mask to unpack its two
// halves, v_sub_f32 for the residual (every intermediate exactly representable): 11 plain VALU instructions per pair of values as written
// (X6_VP; 9 - 10 as issued, most pairs of subtractions become one v_pk_add_f32).  The 7-instruction v_dot2c_f32_bf16 form of round 3 is gone:
// ONE v_dot2c beside an MFMA stream costs 20 cycles (tools/micro/gap_probe.hip), plain VALU half a cycle each.
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  if constexpr (ABL_NOSPLIT) { h = __float_as_uint(x0); m = __float_as_uint(x1); l = h ^ m; return; }
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  l = cvt_pk_bf16(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
// =========================================================================================
// tile GEMM pieces shared by fwd and bwd_dx
// =========================================================================================
// LDS swizzle: 16-byte slot index XOR (row & 15); conflict-free for ds_read_b128 A-fragments
// (rows = lanes) and for the ds_write_b32 epilogue (32 consecutive columns of one row).
__device__ __forceinline__ int hidx(int m, int k) { return m * 256 + ((((k >> 2) ^ (m & 15)) << 2) | (k & 3)); }
__device__ __forceinline__ int eidx(int m, int k) { return m * 64 + ((((k >> 2) ^ (m & 15)) << 2) | (k & 3)); }

// streaming (non-temporal) 16-byte store: saved activations are written once and read much later,
// they must not displace the 2.4 MB of weights every CU re-reads from its XCD's 4 MB L2
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float* p, const float4& v) {
  if constexpr (ABL_NOSTORE) { if (v.x == 1.2345e-30f) *p = v.y; return; }   // (the upper bound of ANY scheme that writes fewer saved bytes)
  f32x4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4v*>(p));
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulate `nks` k-steps (8 wide) of A (LDS, rows wm*64.., E or H layout, first k-step
// a_ks0) times packed B (global, this layer's k-steps b_ks0.., KS k-steps per n-tile).
// Side job (save_dst != nullptr, H layout only, nks == 32): while the MFMAs of this k-loop run, the
// tile's activations -- which this very loop reads from LDS -- are also streamed to HBM as whole
// 1 KiB rows (one ds_read_b128 + one global_store_dwordx4 per lane per two k-steps), instead of 64
// dword stores per wave in the epilogue that produced them.
// AMODE: 0 = H layout (256 floats/row), 1 = E layout (64 floats/row), 2 = X2 (32 floats/row, the
// extra PE channels 64..95 of the 4-D background encoding)
template <int NT, int AMODE>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[2][NT], const float* __restrict__ As, int a_ks0, int nks,
                                         const float4* __restrict__ Bp, int KS, int b_ks0, int nt0, int wm, int lane,
                                         int dbg = 0, float* __restrict__ save_dst = nullptr, int save_valid = 0,
                                         int wave = 0) {
  asm volatile("" : "+v"(lane));  // keep per-call address math inside the call (no cross-layer hoisting)
  const int lrow = lane & 31, lhalf = lane >> 5;
  const float* arow[2];
  int axor[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = wm * 64 + mt * 32 + lrow;
    arow[mt] = As + m * (AMODE == 0 ? 256 : (AMODE == 1 ? 64 : 32));
    axor[mt] = (AMODE == 2) ? ((m >> 1) & 7) : (m & 15);
  }
  const float4* bptr[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bptr[nt] = Bp + ((int64_t)(nt0 + nt) * KS + b_ks0) * 64 + lane;

  // rotating register sets (no copies: a copy of a just-issued load would force vmcnt(0)); the
  // weight fragments are fetched PF k-steps ahead of their MFMAs
  auto load_a = [&](float4 (&a)[2], int ks) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      a[mt] = *reinterpret_cast<const float4*>(arow[mt] + ((((a_ks0 + ks) * 2 + lhalf) ^ axor[mt]) << 2));
  };
  auto load_b = [&](float4 (&b)[NT], int ks) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = bptr[nt][ks * 64];
  };
  auto mfma16 = [&](const float4 (&a)[2], const float4 (&b)[NT]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float av = (t == 0) ? a[mt].x : (t == 1) ? a[mt].y : (t == 2) ? a[mt].z : a[mt].w;
          const float bv = (t == 0) ? b[nt].x : (t == 1) ? b[nt].y : (t == 2) ? b[nt].z : b[nt].w;
          acc[mt][nt] = mfma(av, bv, acc[mt][nt]);
        }
      }
    }
  };
  auto side_copy = [&]() {
    if (AMODE == 0 && save_dst != nullptr) {
      // every load of this loop already issued: stream the tile's rows out in one burst.  (vmcnt
      // retires in order, so a store issued earlier in the loop would sit in front of later weight
      // loads and stall their waits for a full HBM write round trip.)
#pragma unroll 4
      for (int i = 0; i < TM / NWAVES; ++i) {
        const int m = i * NWAVES + wave;
        if (m < save_valid) {
          const float4 v = *reinterpret_cast<const float4*>(As + m * 256 + lane * 4);
          store_nt(save_dst + (unsigned)(m * 256 + ((lane ^ (m & 15)) << 2)), v);
        }
      }
    }
  };
  float4 a0[2], b0[NT], a1[2], b1[NT];
  load_b(b0, 0); load_a(a0, 0);
#pragma unroll 1
  for (int ks = 0; ks < nks; ks += 2) {  // nks is even for every layer
    load_b(b1, ks + 1); load_a(a1, ks + 1);
    mfma16(a0, b0);
    if (ks + 2 < nks) { load_b(b0, ks + 2); load_a(a0, ks + 2); } else side_copy();
    mfma16(a1, b1);
  }
}

// ---- MM_X6: the same tile product on v_mfma_f32_32x32x16_bf16 (file header) -------------------------------------------
// k-steps of 16; a lane (row = lane & 31, kb = lane >> 5) takes k = ks*16 + kb*8 + 0..7: two 16-byte LDS reads of fp32
// activations, split in registers into the three bf16 pieces; the weight pieces arrive pre-split (pack6_kernel), three
// 1 KiB wave loads per column tile and k-step.  Small terms first; every product of a k-step shares the accumulator.
__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split3_frag(const float4& lo4, const float4& hi4, uint4& h, uint4& m, uint4& l) {
  split3_pair(lo4.x, lo4.y, h.x, m.x, l.x);
  split3_pair(lo4.z, lo4.w, h.y, m.y, l.y);
  split3_pair(hi4.x, hi4.y, h.z, m.z, l.z);
  split3_pair(hi4.z, hi4.w, h.w, m.w, l.w);
}
// ---- the pinned MFMA / VALU interleave of the software-pipelined k-loops ---------------------------------------------------------
// On a SIMD the VALU instructions of one wave do not issue while the OTHER wave streams MFMAs back to back, but a wave's own
// independent instructions do issue in the shadow of its own MFMA (about five plain VALU per v_mfma_f32_32x32x16_bf16, half a cycle each:
// profiles/r04_power_limit.md section 1).  The k-loops therefore split the NEXT unit's raw fragments between the MFMAs of the current one;
// sched_group_barrier pins the pattern.
template <int I, int NM, int NVALU, int SYNC = 0>   // NM MFMAs with NVALU VALU instructions spread evenly between them
__device__ __forceinline__ void interleave6() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, SYNC);
    constexpr int nv = ((I + 1) * NVALU) / NM - (I * NVALU) / NM;
    if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x002, nv, SYNC);
    interleave6<I + 1, NM, NVALU, SYNC>();
  }
}
// ---- MM_X6 on v_mfma_f32_16x16x32_bf16 ---------------------------------------------------------------------------------------------
// A wave's 64 x 64 outputs are 4 x 4 tiles of 16 x 16 (64 accumulator registers, as before); a k-step is 32 wide.  Lane (r16 = lane & 15,
// kc = lane >> 4) holds 8 consecutive k of row / column r16 for both operands.  The weight pieces of a k-step (4 column tiles x 3
// pieces = 48 registers) are held for the whole k-step and double buffered; the activation pieces are streamed ROW TILE by row tile:
// "unit" u = (k-step, row tile) issues its 24 MFMAs (6 products x 4 column tiles) on the pieces of row tile u while the raw fragment of
// unit u + 1 is split between them (4 pairs = 44 plain VALU per 24 MFMAs) and the LDS reads of unit u + 2 are issued -- the software
// pipeline of gemm_seg6p at half the k-granularity, with 24 instead of 48 piece registers.
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <bool L16, int NT> struct AccSel { typedef f32x16 type[2][NT]; };
template <int NT> struct AccSel<true, NT> { typedef f32x4m type[4][2 * NT]; };
template <bool L16, int NT> using AccT = typename AccSel<L16, NT>::type;

__device__ __forceinline__ f32x4m mfma16(const uint4& a, const uint4& b, f32x4m c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the arithmetic of a 16 x 16 x 32 tile product: weight / activation pieces, MFMAs, VALU instructions of one pair's split (pin count of the interleave)
constexpr int X6_NPL = 3, X6_NPROD = 6, X6_VP = 11;
struct Pieces16 { unsigned v[3][4]; };   // [piece h | m | l][pair of k] of ONE row tile
__device__ __forceinline__ uint4 piece_frag16(const Pieces16& p, int pl) { return make_uint4(p.v[pl][0], p.v[pl][1], p.v[pl][2], p.v[pl][3]); }
__device__ __forceinline__ void split_pair16(const float4 (&ar)[2], Pieces16& pn, int q) {
  const float4& s4 = ar[q >> 1];
  const float x0 = (q & 1) ? s4.z : s4.x, x1 = (q & 1) ? s4.w : s4.y;
  split3_pair(x0, x1, pn.v[0][q], pn.v[1][q], pn.v[2][q]);
}
// one unit: 6 * CT MFMAs of row tile MT on the pieces pc and the k-step's weight pieces b; between them the split of `ar` (the next unit's
// raw fragment) into pn and, behind its last pair, `refill()` (the LDS reads that reload ar for the unit after that)
#define X6_PA {2, 0, 1, 1, 0, 0}   // piece of the activations / of the weights in product t (small terms first)
#define X6_PB {0, 2, 1, 0, 1, 0}
template <int CT, int MT, int SYNC, typename ACC, typename RF>
__device__ __forceinline__ void unit16(ACC& acc, const Pieces16& pc, const uint4 (&b)[CT][3], float4 (&ar)[2], Pieces16& pn, RF&& refill) {
  constexpr int PA[6] = X6_PA, PB[6] = X6_PB;
  constexpr int NPROD = X6_NPROD, NM = NPROD * CT;
#pragma unroll
  for (int t = 0; t < NPROD; ++t)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int i = t * CT + ct;
      acc[MT][ct] = mfma16(piece_frag16(pc, PA[t]), b[ct][PB[t]], acc[MT][ct]);
#pragma unroll
      for (int pair = (i * 4) / NM; pair < ((i + 1) * 4) / NM; ++pair) {
        split_pair16(ar, pn, pair);
        if (pair == 3) refill();
      }
    }
  interleave6<0, NM, 4 * X6_VP, SYNC>();
}
// Chained weights (dX): the weight pieces of a segment's first k-step are loaded by the CALLER one layer ahead -- between the barrier that ends
// the previous layer's k-loop and its epilogue -- into registers that are dead there (they are the k-loop's own double buffer): the
// segment starts without waiting for L2 (tools/x6_timing.py: ~2 000 cycles per layer before the first MFMA otherwise).
struct NoChain {};
template <int CT> struct WRegs { uint4 b0[CT][3]; };   // k-step 0 (k-step 1 is not needed for ~3 000 cycles: the segment loads it itself)
template <bool ON, int NT> struct WRegsSel { typedef NoChain type; };
template <int NT> struct WRegsSel<true, NT> { typedef WRegs<2 * NT> type; };
template <bool L16, int NT> using WRegsT = typename WRegsSel<L16, NT>::type;
// arguments as gemm<>'s: KS, b_ks0, nks in the 8-wide k units of the call sites, nt0 = the wave's first 32-column tile
template <int CT>
__device__ __forceinline__ void wprefetch(WRegs<CT>& w, const void* Bw, int KS, int b_ks0, int /*nks*/, int nt0, int lane) {
  const uint4* Bp = reinterpret_cast<const uint4*>(Bw);
  unsigned blane = (unsigned)lane * 16u;
  asm volatile("" : "+v"(blane));
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const char* p = reinterpret_cast<const char*>(Bp + ((int64_t)(nt0 * 2 + ct) * (KS / 4) + b_ks0 / 4) * 192);
#pragma unroll
    for (int pl = 0; pl < X6_NPL; ++pl) w.b0[ct][pl] = *reinterpret_cast<const uint4*>((p + (pl * 64) * 16) + blane);
  }
}
__device__ __forceinline__ void wprefetch(NoChain&, const void*, int, int, int, int, int) {}

template <int NT, int AMODE, bool PRE, typename ACC, typename W>
__device__ __forceinline__ void gemm_seg16(ACC& acc, const float* __restrict__ As, int a_ks0, int nks, const uint4* __restrict__ Bp, int KS,
                                           int b_ks0, int nt0, int wm, int lane, float* __restrict__ save_dst, int save_valid, int wave,
                                           W& wext) {
  asm volatile("" : "+v"(lane));
  constexpr int CT = 2 * NT;
  const int r16 = lane & 15, kc = lane >> 4;
  constexpr int RS = AMODE == 0 ? 256 : (AMODE == 1 ? 64 : 32);
  const int m0 = wm * 64 + r16;
  // row tile mt: row m0 + 16 mt; the swizzle term of H / E rows (m & 15) does not depend on mt
  const float* arow0 = As + m0 * RS;
  auto load_raw = [&](float4 (&a)[2], int mt, int ks) {
    const int m = m0 + 16 * mt;
    const int ax = (AMODE == 2) ? ((m >> 1) & 7) : (m & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      a[j] = *reinterpret_cast<const float4*>(arow0 + mt * (16 * RS) + ((((a_ks0 + ks) * 8 + kc * 2 + j) ^ ax) << 2));
  };
  const char* bptr[CT];   // (scalar-base weight loads: see gemm_seg6)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bptr[ct] = reinterpret_cast<const char*>(Bp + ((int64_t)(nt0 * 2 + ct) * KS + b_ks0) * 192);
  unsigned blane = (unsigned)lane * 16u;
  auto load_b = [&](uint4 (&b)[CT][3], int ks) {
    asm volatile("" : "+v"(blane));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pl = 0; pl < X6_NPL; ++pl) {
        if (pl >= ABL_WPIECES) { b[ct][pl] = b[ct][0]; continue; }
        b[ct][pl] = *reinterpret_cast<const uint4*>((bptr[ct] + (ks * 192 + pl * 64) * 16) + blane);
      }
  };
  const int klast = nks - 1;
  float4 r0[2], r1[2];          // raw fragments of units u + 1 (being split) and u + 2 (in flight)
  Pieces16 pa, pb;              // pieces of the current and the next unit
  constexpr bool CHAINED = !std::is_same<W, NoChain>::value;
  static_assert(CHAINED || !PRE, "preloaded weights come through a WRegs");
  WRegs<CT> wloc_;
  WRegs<CT>& wr_ = [&]() -> WRegs<CT>& { if constexpr (CHAINED) return wext; else return wloc_; }();
  uint4 (&b0)[CT][3] = wr_.b0;
  uint4 b1[CT][3];
  if constexpr (!PRE) load_b(b0, 0);
  load_raw(r0, 0, 0);
  load_raw(r1, 1, 0);
  load_b(b1, klast > 0 ? 1 : 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair16(r0, pa, q);     // unit 0 is split up front; its registers then take unit 2
  load_raw(r0, 2, 0);
  __builtin_amdgcn_s_setprio(1);
  // unit u = 4 ks + mt:   pieces  pa (u even) / pb (u odd);   splits raw r1 (u even) / r0 (u odd) = unit u + 1;   refills it with unit u + 3
  // (clamped at the segment's last k-step: re-reads, never used)
  auto kstep = [&](const uint4 (&b)[CT][3], int ks, auto par) __attribute__((always_inline)) {
    constexpr int S0 = 1 + 4 * decltype(par)::value;   // one sched_group_barrier pipeline per unit of the loop body
    const int kn = ks + 1 < klast ? ks + 1 : klast;   // k-step of units u + 3 / u + 4 once they wrap
    unit16<CT, 0, S0 + 0>(acc, pa, b, r1, pb, [&]() { load_raw(r1, 3, ks); });
    unit16<CT, 1, S0 + 1>(acc, pb, b, r0, pa, [&]() { load_raw(r0, 0, kn); });
    unit16<CT, 2, S0 + 2>(acc, pa, b, r1, pb, [&]() { load_raw(r1, 1, kn); });
    unit16<CT, 3, S0 + 3>(acc, pb, b, r0, pa, [&]() { load_raw(r0, 2, kn); });
  };
  constexpr std::integral_constant<int, 0> EVEN{};
  constexpr std::integral_constant<int, 1> ODD{};
  const bool saving = AMODE == 0 && save_dst != nullptr;
#pragma unroll 1
  for (int ks = 0; ks + 2 <= nks - (saving ? 1 : 0); ks += 2) {
    kstep(b0, ks, EVEN);
    load_b(b0, ks + 2 < klast ? ks + 2 : klast);
    kstep(b1, ks + 1, ODD);
    load_b(b1, ks + 3 < klast ? ks + 3 : klast);
  }
  if (!saving) {
    if (nks & 1) kstep(b0, klast, EVEN);          // (single-step segments: the 32 extra encoding channels)
  } else {
    // nks is even (8) for every saved segment: k-step nks - 2 as above, then the LAST k-step with the tile's rows streamed out between
    // its MFMAs (every load of the segment has been issued; rows beyond the valid count are clamped: rewritten with the same bytes)
    kstep(b0, nks - 2, EVEN);
    constexpr int PA[6] = X6_PA, PB[6] = X6_PB;
    constexpr int NPROD = X6_NPROD, ROWS = TM / NWAVES, NM = NPROD * CT;
    const int last_row = save_valid - 1;
    auto last_unit = [&](auto mtc, const Pieces16& pc, float4 (&ar)[2], Pieces16& pn) __attribute__((always_inline)) {
      constexpr int MT = decltype(mtc)::value;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int i = t * CT + ct;
          acc[MT][ct] = mfma16(piece_frag16(pc, PA[t]), b1[ct][PB[t]], acc[MT][ct]);
          if (MT < 3) {
#pragma unroll
            for (int pair = (i * 4) / NM; pair < ((i + 1) * 4) / NM; ++pair) split_pair16(ar, pn, pair);
          }
#pragma unroll
          for (int r = (i * (ROWS / 4)) / NM; r < ((i + 1) * (ROWS / 4)) / NM; ++r) {
            int m = (MT * (ROWS / 4) + r) * NWAVES + wave;
            m = m < last_row ? m : last_row;
            const float4 v = *reinterpret_cast<const float4*>(As + m * 256 + lane * 4);
            store_nt(save_dst + (unsigned)(m * 256 + ((lane ^ (m & 15)) << 2)), v);
          }
        }
    };
    last_unit(std::integral_constant<int, 0>{}, pa, r1, pb);
    load_raw(r1, 3, klast);
    last_unit(std::integral_constant<int, 1>{}, pb, r0, pa);
    last_unit(std::integral_constant<int, 2>{}, pa, r1, pb);
    last_unit(std::integral_constant<int, 3>{}, pb, r0, pa);
  }
  __builtin_amdgcn_s_setprio(0);
}

// one call site for both math modes: k-steps in the 8-wide units of gemm_seg, Bw = the layer's block in this mode's packing
template <int MM, int NT, int AMODE, bool PRE, typename W>
__device__ __forceinline__ void gemm(f32x4m (&acc)[4][2 * NT], const float* __restrict__ As, int a_ks0, int nks, const void* Bw,
                                     int KS, int b_ks0, int nt0, int wm, int lane, int dbg,
                                     float* __restrict__ save_dst, int save_valid, int wave, W& w) {
  static_assert(MM == MM_X6, "the 16 x 16 accumulator layout belongs to the bf16x6 kernels");
  gemm_seg16<NT, AMODE, PRE>(acc, As, a_ks0 / 4, nks / 4, reinterpret_cast<const uint4*>(Bw), KS / 4, b_ks0 / 4, nt0, wm, lane, save_dst,
                             save_valid, wave, w);
}
template <int MM, int NT, int AMODE>
__device__ __forceinline__ void gemm(f32x4m (&acc)[4][2 * NT], const float* __restrict__ As, int a_ks0, int nks, const void* Bw,
                                     int KS, int b_ks0, int nt0, int wm, int lane, int dbg = 0,
                                     float* __restrict__ save_dst = nullptr, int save_valid = 0, int wave = 0) {
  NoChain nc;
  gemm<MM, NT, AMODE, false>(acc, As, a_ks0, nks, Bw, KS, b_ks0, nt0, wm, lane, dbg, save_dst, save_valid, wave, nc);
}
template <int MM, int NT, int AMODE>
__device__ __forceinline__ void gemm(f32x16 (&acc)[2][NT], const float* __restrict__ As, int a_ks0, int nks, const void* Bw,
                                     int KS, int b_ks0, int nt0, int wm, int lane, int dbg = 0,
                                     float* __restrict__ save_dst = nullptr, int save_valid = 0, int wave = 0) {
  static_assert(MM == MM_F32, "the 32 x 32 accumulator layout belongs to the fp32-MFMA kernels");
  gemm_seg<NT, AMODE>(acc, As, a_ks0, nks, reinterpret_cast<const float4*>(Bw), KS, b_ks0, nt0, wm, lane, dbg, save_dst, save_valid, wave);
}
template <int MM, int NT, int AMODE, bool PRE, typename W>   // (the 32 x 32 layouts take no preloaded weights: W is NoChain there)
__device__ __forceinline__ void gemm(f32x16 (&acc)[2][NT], const float* __restrict__ As, int a_ks0, int nks, const void* Bw,
                                     int KS, int b_ks0, int nt0, int wm, int lane, int dbg,
                                     float* __restrict__ save_dst, int save_valid, int wave, W&) {
  static_assert(std::is_same<W, NoChain>::value, "preloaded weights belong to the 16 x 16 path");
  gemm<MM, NT, AMODE>(acc, As, a_ks0, nks, Bw, KS, b_ks0, nt0, wm, lane, dbg, save_dst, save_valid, wave);
}
// the block of a layer whose fp32 packing starts `off` floats into the packed buffer
template <int MM>
__device__ __forceinline__ const void* wblock(const float* packed, int64_t off) {
  if constexpr (MM != MM_F32) return reinterpret_cast<const uint4*>(packed) + off * 3 / 8;
  else return reinterpret_cast<const float4*>(packed) + off / 4;
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4m (&acc)[4][2 * NT]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2 * NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][ct][r] = 0.f;
}
// Bias folding: a layer's bias is the accumulators' initial value (the column is the same for a lane's four rows of every row tile) instead of 64
// additions in the epilogue -- whose VALU instructions starve beside the partner wave's MFMA stream (tools/x6_timing.py: 14 cycles each)
template <int NT>
__device__ __forceinline__ void bias_acc(f32x4m (&acc)[4][2 * NT], const float (&bv)[2 * NT]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2 * NT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][ct][r] = bv[ct];
}
template <int NT, bool FOLD, typename ACC, int NB>
__device__ __forceinline__ void init_acc(ACC& acc, const float (&bv)[NB]) {
  if constexpr (FOLD) bias_acc<NT>(acc, bv);
  else zero_acc<NT>(acc);
}
// C layout of v_mfma_f32_16x16x32_bf16: col = lane & 15, row = 4 * (lane >> 4) + r.  Element (mt, ct, r) of a wave's 64 x 64 block is
// H[wm*64 + mt*16 + 4*hq + r][(wn*CT + ct)*16 + (lane & 15)], hq = lane >> 4; the row's swizzle term m & 15 = (hq << 2) | r separates as in
// h_cols: four column pointers per column tile (one per r), everything else an immediate.
template <int CT>
__device__ __forceinline__ void h_cols16(float* Hs, int wm, int wn, int lane, float* (&colp)[CT][4]) {
  const int hq = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int n = (wn * CT + ct) * 16 + (lane & 15);
    float* const rowp = Hs + (wm * 64 + 4 * hq) * 256 + (n & 3);
    const int q = (n >> 2) ^ (hq << 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) colp[ct][r] = rowp + ((q ^ r) << 2);
  }
}
#define H16_AT(colp, ct, mt, r) ((colp)[ct][r][((mt) * 16 + (r)) * 256])

// C layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Epilogue addressing of an accumulator tile into H.  hidx(m, n) with m = wm*64 + mt*32 + crow(r, lane): the swizzle term m & 15 is
// (r & 3) | (lane >> 5) << 2 | ((r >> 2) & 1) << 3 -- disjoint bit fields, so the XOR separates: eight column pointers per column
// tile (j = (r & 3) + 4 * ((r >> 2) & 1)), computed once per epilogue, and everything else of the address is an immediate
// (< 60 KiB): no per-value address arithmetic (it was 3 of the 5 VALU instructions per value, and VALU time is not hidden under
// the partner wave's MFMAs: profiles/r03_mfma_valu_exclusion.md).
template <int NT>
__device__ __forceinline__ void h_cols(float* Hs, int wm, int wn, int lane, float* (&colp)[NT][8]) {
  const int h = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (wn * NT + nt) * 32 + (lane & 31);
    float* const rowp = Hs + (wm * 64 + 4 * h) * 256 + (n & 3);
    const int q = (n >> 2) ^ (4 * h);
#pragma unroll
    for (int j = 0; j < 8; ++j) colp[nt][j] = rowp + ((q ^ (j & 3) ^ ((j >> 2) << 3)) << 2);
  }
}
// the element (mt, r) of a column tile: &H[hidx(wm*64 + mt*32 + crow(r, lane), n)]
#define H_AT(colp, nt, mt, r) ((colp)[nt][((r) & 3) + 4 * (((r) >> 2) & 1)][((mt) * 32 + ((r) & 3) + 8 * ((r) >> 2)) * 256])

// forward epilogue: + bias, optional ReLU, write H (LDS) and optionally the saved activation.
// When `mask_out` is given (training, ReLU layers) every lane records the sign pattern of ITS 64 accumulator values in one 64-bit word
// -- value i = (nt*2+mt)*16 + r is bit 31 - (i & 31) of half i >> 5: `v_cmp_lt_f32 vcc, 0, v ; v_addc_co_u32 w, vcc, w, w, vcc` shifts the
// word left and takes the compare as the new bit 0 (two instructions per value, no scalar round trip) -- and the wave stores its 64 words
// with one coalesced 512-byte access.  mlp_bwd_dx (same wave -> tile mapping, same lane) reads its word back and masks a gradient with
// `v_bfe_i32` + `v_and_b32`, instead of re-reading 1 KB/point/layer of activations.  (Round 3 kept wave BALLOTS, one per value, moved
// into lane i with `s_nop 3` + two `v_writelane` and fetched in dX with two `v_readlane` + select + shift: 6 / 6 instructions per value
// where this takes 5 / 2.)
template <int NT, int NB>
__device__ __forceinline__ void load_bias(float (&bv)[NB], const float* __restrict__ bias, int wn, int lane) {
  static_assert(NB == NT || NB == 2 * NT, "NT columns tiles of 32 or 2 NT of 16");
  if constexpr (NB == NT) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = bias[(wn * NT + nt) * 32 + (lane & 31)];
  } else {
#pragma unroll
    for (int ct = 0; ct < NB; ++ct) bv[ct] = bias[(wn * NB + ct) * 16 + (lane & 15)];
  }
}

// bias values are loaded by the caller BEFORE the k-loop (load_bias) so that no global load waits
// behind the activation stores issued at the end of the loop
template <int NT, bool RELU, bool MASKS = false, bool FOLDED = false>
__device__ __forceinline__ void epilogue_fwd(const f32x16 (&acc)[2][NT], const float (&bias_v)[NT], float* Hs,
                                             int wm, int wn, int lane, float* __restrict__ save, int ldsave,
                                             int valid, unsigned long long* __restrict__ mask_out = nullptr) {
  asm volatile("" : "+v"(lane));
  unsigned wlo = 0u, whi = 0u;   // this lane's sign word (see above)
  constexpr bool want_mask = RELU && NT == 2 && MASKS;
  float* colp[NT][8];
  h_cols<NT>(Hs, wm, wn, lane, colp);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (wn * NT + nt) * 32 + (lane & 31);
    const float bv = bias_v[nt];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 64 + mt * 32 + crow(r, lane);
        float v = acc[mt][nt][r] + bv;
        if (want_mask) {
          // sign bit into the word, then ReLU: three instructions, vcc lives only inside the block
          if ((nt * 2 + mt) * 16 + r < 32)
            asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_max_f32 %1, 0, %1" : "+v"(wlo), "+v"(v) : : "vcc");
          else
            asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_max_f32 %1, 0, %1" : "+v"(whi), "+v"(v) : : "vcc");
        } else if (RELU) {
          v = v > 0.f ? v : 0.f;   // (one compare + select; fmaxf is two v_max: it canonicalises first)
        }
        H_AT(colp, nt, mt, r) = v;
        if (save != nullptr && m < valid) save[(unsigned)(m * ldsave + n)] = v;
      }
      __builtin_amdgcn_sched_barrier(0);  // bound live ranges: one 32x32 tile at a time
    }
  }
  if (want_mask) mask_out[lane] = ((unsigned long long)whi << 32) | (unsigned long long)wlo;
}


// the same epilogue on the 16 x 16 accumulator layout: value index i = (mt * CT + ct) * 4 + r in the lane's sign word
template <int NT, bool RELU, bool MASKS = false, bool FOLDED = false>
__device__ __forceinline__ void epilogue_fwd(const f32x4m (&acc)[4][2 * NT], const float (&bias_v)[2 * NT], float* Hs, int wm, int wn, int lane,
                                             float* __restrict__ save, int ldsave, int valid,
                                             unsigned long long* __restrict__ mask_out = nullptr) {
  asm volatile("" : "+v"(lane));
  constexpr int CT = 2 * NT;
  unsigned wlo = 0u, whi = 0u;
  constexpr bool want_mask = RELU && NT == 2 && MASKS;
  float* colp[CT][4];
  h_cols16<CT>(Hs, wm, wn, lane, colp);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float bv = bias_v[ct];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = FOLDED ? acc[mt][ct][r] : acc[mt][ct][r] + bv;
        if (want_mask) {
          if ((mt * CT + ct) * 4 + r < 32)
            asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_max_f32 %1, 0, %1" : "+v"(wlo), "+v"(v) : : "vcc");
          else
            asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_max_f32 %1, 0, %1" : "+v"(whi), "+v"(v) : : "vcc");
        } else if (RELU) {
          v = v > 0.f ? v : 0.f;
        }
        H16_AT(colp, ct, mt, r) = v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // bound live ranges: one row tile at a time
  }
  if (want_mask) mask_out[lane] = ((unsigned long long)whi << 32) | (unsigned long long)wlo;
}


// Two workgroups share every CU (two waves per SIMD share one MFMA pipe).  Launched together on
// identical work they run in lockstep -- both in their k-loops (pipe shared) and then both in
// their epilogues (pipe idle).  A one-off pseudo-random start delay (0..15 x 1024 cycles, larger
// than an epilogue) de-phases them so that one workgroup's epilogue / barrier / PE phase overlaps
// the other's MFMAs (measured: profiles/r01_summary.md).
__device__ __forceinline__ void stagger_start() {
  static_assert(TM == 64, "the stagger de-phases the TWO workgroups of a CU");
  const unsigned h = ((unsigned)blockIdx.x * 2654435761u) >> 28;  // 0..15
  for (unsigned i = 0; i < h; ++i) __builtin_amdgcn_s_sleep(16);    // 16 x 64 cycles
}

