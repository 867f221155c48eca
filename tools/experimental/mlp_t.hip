// mlp_t.hip -- "transposed chain" split-bf16 MLP forward for gfx950: activations stay in registers.
//
// EXPERIMENTAL, NOT BUILT (not in build.py, no C-ABI entry).  Inference forward only; bit-for-bit the same math
// as mlp_bf16.hip (max |d raw| 1.8e-6 against it on 786k points).  Measured on MI355X, 4096x192 points:
//   mlp_bf16.hip forward 2.55 ms | this kernel 2.73 ms | this kernel with only its MFMAs left 1.73 ms
//   ablations of this kernel: LDS-DMA of the weight slabs ~0.6 ms (global_load_lds costs ~100 issue cycles per
//   KiB with one wave per SIMD), epilogue ~0.35 ms, slab barrier ~0.1 ms.
// Kept as the starting point for round 2: the register-resident chain removes the activation
// LDS traffic and halves the weight bytes per point, but with one wave per SIMD every non-MFMA instruction must
// fit the ~28-cycle window behind an MFMA, and the LDS-DMA does not.  Replacing the DMA by a register-staged copy
// (4 staging uint4 per thread, loads / ds_writes spread over the k-steps) was tried: hipcc then spills inside the
// slab loop (560 B/lane scratch, 7.1 ms) because it does not keep the 128 input-fragment registers in AccVGPRs;
// this kernel needs an explicit VGPR/AGPR partition (inline-asm MFMAs with "a" operands or hand-written asm).
// tools/t_fwd.py builds and times it stand-alone.
//
// Orientation: every layer is computed as out^T[n][m] = sum_k W[n][k] * act^T[k][m]  (n = output channel,
// m = point).  The MFMA A operand is a weight fragment, the B operand an activation fragment (lane = point
// m = lane&31, 8 consecutive input channels k = ks*16 + (lane>>5)*8 + 0..7).  The 32x32 accumulator tile gives a
// lane 16 output CHANNELS of its own point, which -- after bias/ReLU, the (hi, lo) bf16 split and one
// v_permlane32_swap per register pair -- IS the B fragment of the next layer.  So a wave carries its 32 points
// through the whole network without LDS traffic, barriers or address arithmetic for activations.
//
// The LDS is used for the weights instead: a workgroup (4 waves x 32 points) stages one 32-output-channel slab
// (all K, hi+lo planes, <= 40 KiB) at a time, double buffered; every wave reads the slab as A fragments.
// Global->LDS weight traffic is 2.4 MB per 128 points (mlp_bf16.hip: per 64), LDS reads are contiguous 1 KiB
// fragments, and the only barrier is the slab hand-over.
//
// Math is the same 3-term split as mlp_bf16.hip (hi*hi + hi*lo + lo*hi, fp32 accumulate).  The alpha/sigma head
// and the rgb head run as extra 32-row slabs on the matrix cores (rows 1.. / 3.. are zero).
#include <stdlib.h>
#include "common.h"
#include "mlp_layout.h"

using namespace fnl;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef T_PF
#define T_PF 3
#endif
#define TW 128                 // points per workgroup
#define TTHR 256
#define T_SLOT_U4 2560         // uint4 per LDS slot (40 KiB: 32 rows x 320 k x (hi, lo))
#define T_BIAS_FLOATS 2440      // [L0..L7 | F | V | ba | br0..2 | pad]
#define T_OFF_BIAS (2 * T_SLOT_U4 * 16)
#define T_OFF_PE (T_OFF_BIAS + T_BIAS_FLOATS * 4)               // per-lane stash of the PE fragments: 8 uint4 per thread
#define T_LDS_BYTES (T_OFF_PE + TTHR * 8 * 16)

static int t_num_cus() {
  static int n = 0;
  if (n > 0) return n;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
  if (n <= 0) n = 256;
  return n;
}

__device__ __forceinline__ unsigned t_bf16_rne(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ unsigned t_cvt_pk(float a, float b) {   // bf16(a) | bf16(b) << 16 (v_cvt_pk_bf16_f32)
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void t_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = t_cvt_pk(a, b);
  lo = t_cvt_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ f32x16 t_mfma(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- packed weights: slabs of 32 output rows, A-fragment order ---------------------------------------
// slab (uint4 units): ((ks*2 + part)*64 + lane) -> 8 bf16 = W'[row0 + (lane&31)][ks*16 + (lane>>5)*8 + 0..7]
// W' columns = [segment A padded to segA_pad | segment B]; rows >= rows_valid are zero (head slabs).
struct TPackDesc {
  int64_t src_off, dst_off;      // floats / uint4
  int ld, nslab, KS, rows_valid;
  int segA_pad, segA_valid, segA_col0, segB_valid, segB_col0;
};
struct TPackTable { TPackDesc d[12]; };

__global__ void __launch_bounds__(256) tpack_kernel(TPackTable tab, const float* __restrict__ params,
                                                     uint4* __restrict__ dst) {
  const TPackDesc d = tab.d[blockIdx.y];
  const int64_t total = (int64_t)d.nslab * d.KS * 2 * 64;
  const float* src = params + d.src_off;
  uint4* out = dst + d.dst_off;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(e & 63);
    const int part = (int)((e >> 6) & 1);
    const int64_t blk = e >> 7;
    const int slab = (int)(blk / d.KS), ks = (int)(blk % d.KS);
    const int n = slab * 32 + (l & 31);
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kp = ks * 16 + (l >> 5) * 8 + j;
      int col = -1;
      if (kp < d.segA_pad) { if (kp < d.segA_valid) col = d.segA_col0 + kp; }
      else { const int q = kp - d.segA_pad; if (q < d.segB_valid) col = d.segB_col0 + q; }
      const float v = (col >= 0 && n < d.rows_valid) ? src[(int64_t)n * d.ld + col] : 0.f;
      const unsigned hi = t_bf16_rne(v);
      const unsigned lo = t_bf16_rne(v - __uint_as_float(hi << 16));
      w[j] = part ? lo : hi;
    }
    uint4 o;
    o.x = w[0] | (w[1] << 16); o.y = w[2] | (w[3] << 16); o.z = w[4] | (w[5] << 16); o.w = w[6] | (w[7] << 16);
    out[e] = o;
  }
}

__global__ void __launch_bounds__(256) tbias_kernel(const float* __restrict__ params, float* __restrict__ dst, NetLayout lay) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // 2432 entries
  if (i < 2048) dst[i] = params[lay.LB[i >> 8] + (i & 255)];
  else if (i < 2304) dst[i] = params[lay.FB + (i - 2048)];
  else if (i < 2432) dst[i] = params[lay.VB + (i - 2304)];
  else if (i == 2432) dst[i] = params[lay.AB];
  else if (i < 2436) dst[i] = params[lay.RB + (i - 2433)];
  else if (i < 2440) dst[i] = 0.f;
}

// forward slab groups: 0..7 trunk, 8 feature, 9 alpha head, 10 view, 11 rgb head
struct TOff { int64_t off[12]; int64_t bias; int64_t total; };   // bias: fp32 table, T_BIAS_FLOATS
static TOff t_offsets() {
  TOff o{};
  const int nslab[12] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 1, 4, 1};
  const int KS[12] = {4, 16, 16, 16, 16, 20, 16, 16, 16, 16, 18, 8};
  int64_t p = 0;
  for (int i = 0; i < 12; ++i) { o.off[i] = p; p += (int64_t)nslab[i] * KS[i] * 128; }
  o.bias = p;
  p += T_BIAS_FLOATS / 4;
  o.total = p;
  return o;
}
static const NetLayout& t_layout(int kind) {
  static const NetLayout L[3] = {make_layout(0), make_layout(1), make_layout(2)};
  return L[kind < 0 || kind > 2 ? 0 : kind];
}

extern "C" int64_t fastnerf_mlp_t_floats(int kind, int what) {
  if (kind < 0 || kind > 1) return -1;
  if (what == 1) return t_offsets().total * 4;
  return -1;
}

extern "C" int fastnerf_mlp_t_pack(int kind, const float* params, float* packed_fwd, fn_stream_t stream) {
  FN_CHECK_ARG((kind == 0 || kind == 1) && params && packed_fwd, "kind in {0,1}, non-null pointers");
  const NetLayout& L = t_layout(kind);
  const TOff O = t_offsets();
  TPackTable T;
  for (int l = 0; l < 8; ++l) {
    TPackDesc d{};
    d.src_off = L.LW[l]; d.dst_off = O.off[l]; d.nslab = 8; d.rows_valid = 256;
    d.ld = (l == 0) ? L.in_pe : (l == 5 ? 256 + L.in_pe : 256);
    if (l == 0) { d.KS = 4; d.segA_pad = 64; d.segA_valid = L.in_pe; }
    else if (l == 5) { d.KS = 20; d.segA_pad = 64; d.segA_valid = L.in_pe; d.segB_valid = 256; d.segB_col0 = L.in_pe; }
    else { d.KS = 16; d.segA_pad = 0; d.segB_valid = 256; }
    T.d[l] = d;
  }
  { TPackDesc d{}; d.src_off = L.FW; d.dst_off = O.off[8]; d.ld = 256; d.nslab = 8; d.KS = 16; d.rows_valid = 256; d.segB_valid = 256; T.d[8] = d; }
  { TPackDesc d{}; d.src_off = L.AW; d.dst_off = O.off[9]; d.ld = 256; d.nslab = 1; d.KS = 16; d.rows_valid = 1; d.segB_valid = 256; T.d[9] = d; }
  { TPackDesc d{}; d.src_off = L.VW; d.dst_off = O.off[10]; d.ld = 283; d.nslab = 4; d.KS = 18; d.rows_valid = 128;
    d.segA_pad = 32; d.segA_valid = 27; d.segA_col0 = 256; d.segB_valid = 256; d.segB_col0 = 0; T.d[10] = d; }
  { TPackDesc d{}; d.src_off = L.RW; d.dst_off = O.off[11]; d.ld = 128; d.nslab = 1; d.KS = 8; d.rows_valid = 3; d.segB_valid = 128; T.d[11] = d; }
  hipLaunchKernelGGL(tpack_kernel, dim3(32, 12), dim3(256), 0, fn::S(stream), T, params, reinterpret_cast<uint4*>(packed_fwd));
  FN_LAUNCH_CHECK();
  hipLaunchKernelGGL(tbias_kernel, dim3(10), dim3(256), 0, fn::S(stream), params, packed_fwd + O.bias * 4, L);
  FN_LAUNCH_CHECK();
  return 0;
}

// ---- slab hand-over: LDS-DMA (global_load_lds_dwordx4), wave w copies the w-th KiB of every 4 KiB ------------
__device__ __forceinline__ void slab_dma(char* slot, const uint4* __restrict__ src, int n_u4, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 10; ++i)
    if (i * TTHR < n_u4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * TTHR + wave * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(slot + (i * TTHR + wave * 64) * 16), 16, 0, 0);
}

__device__ __forceinline__ void dma_piece(char* slot, const uint4* __restrict__ src, int n_u4, int i, int wave, int lane) {
  if (i * TTHR < n_u4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * TTHR + wave * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(slot + (i * TTHR + wave * 64) * 16), 16, 0, 0);
}

// 3-term MFMA over KS k-steps of one 32-row slab: A fragments from LDS (fragment index a0 onwards), B fragments
// through getB(ks, part).  Two accumulators (even / odd k-steps) keep consecutive MFMAs independent.  fill(ks) is
// emitted after the MFMAs of every k-step: independent work (the previous tile's epilogue, the next slab's DMA)
// that the wave issues while the matrix pipe is busy; sched_barriers keep the interleaving as written.
template <int KS, typename GetB, typename Fill>
__device__ __forceinline__ void tile_core(f32x16& acc, f32x16& acc2, const char* slot, int a0, int lane, GetB getB, Fill fill) {
  // One wave per SIMD: whatever is not an MFMA has to be issued in the ~28-cycle window behind each MFMA (the next
  // MFMA cannot issue before the matrix pipe frees up, everything placed in between is free).  So every k-step is
  //   mfma | ds_read A.hi(ks+T_PF), fill(ks,0) | mfma | ds_read A.lo(ks+T_PF), fill(ks,1) | mfma | fill(ks,2)
  // with sched_barriers pinning the order.  A fragments are read T_PF k-steps ahead (LDS latency ~130 cycles).
  const char* base = slot + (a0 * 128 + lane) * 16;
  uint4 ah[T_PF + 1], al[T_PF + 1];
#pragma unroll
  for (int i = 0; i < T_PF; ++i)
    if (i < KS) {
      ah[i] = *reinterpret_cast<const uint4*>(base + i * 2048);
      al[i] = *reinterpret_cast<const uint4*>(base + i * 2048 + 1024);
    }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const uint4& a_h = ah[ks % (T_PF + 1)];
    const uint4& a_l = al[ks % (T_PF + 1)];
    f32x16& c0 = (ks & 1) ? acc2 : acc;
    f32x16& c1 = (ks & 1) ? acc : acc2;
    __builtin_amdgcn_sched_barrier(0);
    c0 = t_mfma(a_h, getB(ks, 0), c0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + T_PF < KS) ah[(ks + T_PF) % (T_PF + 1)] = *reinterpret_cast<const uint4*>(base + (ks + T_PF) * 2048);
    fill(ks, 0);
    __builtin_amdgcn_sched_barrier(0);
    c1 = t_mfma(a_h, getB(ks, 1), c1);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + T_PF < KS) al[(ks + T_PF) % (T_PF + 1)] = *reinterpret_cast<const uint4*>(base + (ks + T_PF) * 2048 + 1024);
    fill(ks, 1);
    __builtin_amdgcn_sched_barrier(0);
    c0 = t_mfma(a_l, getB(ks, 0), c0);
    __builtin_amdgcn_sched_barrier(0);
    fill(ks, 2);
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void acc_zero(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ __forceinline__ void acc_add(f32x16& a, const f32x16& b) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] += b[r];
}

// C layout of the 32x32 MFMAs: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// accumulator tile (32 output channels of this lane's point) -> two B fragments (k-steps 2t, 2t+1) of the next
// layer, in 6 slices so that it can be spread over the k-steps of the following tile:
//   slice g = 0..3: rows 4g..4g+3: acc + acc2, max(floor), (hi, lo) split           (~20 VALU)
//   slice 4 / 5   : half exchange (v_permlane32_swap) and assembly of fragment 0 / 1 (~4 swaps each)
struct Epi { unsigned H[4][2], L[4][2]; };
// finished fragments are only ever read as MFMA B operands: keep them in AccVGPRs, the arch VGPRs are needed for
// the A-operand prefetch, the weight staging and the fragments under construction
__device__ __forceinline__ void park1(uint4& h, uint4& l) {
  asm volatile("" : "+a"(h.x), "+a"(h.y), "+a"(h.z), "+a"(h.w));
  asm volatile("" : "+a"(l.x), "+a"(l.y), "+a"(l.z), "+a"(l.w));
}
// 12 slices of ~6 VALU each: slices 0..7: rows 2s, 2s+1 (max, hi/lo split); 8..11: half exchange + assembly
#define EPI_SLICES 12
__device__ __forceinline__ void epi_slice(Epi& e, const f32x16& acc, float floor_v, int sl, uint4& h0, uint4& l0, uint4& h1,
                                          uint4& l1) {
  auto sw = [](unsigned& x, unsigned& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0]; y = r[1];
  };
  if (sl < 8) {
    const int g = sl >> 1, q = sl & 1;
    const float v0 = fmaxf(acc[4 * g + 2 * q], floor_v), v1 = fmaxf(acc[4 * g + 2 * q + 1], floor_v);
    t_split_pair(v0, v1, e.H[g][q], e.L[g][q]);
  } else if (sl == 8) {
    // lanes 0..31 hold rows 8g..8g+3, lanes 32..63 rows 8g+4..8g+7: after the exchange kb=0 lanes own channels
    // 0..7 and kb=1 lanes 8..15 of the 16-channel k-step
    sw(e.H[0][0], e.H[1][0]); sw(e.H[0][1], e.H[1][1]);
  } else if (sl == 9) {
    sw(e.L[0][0], e.L[1][0]); sw(e.L[0][1], e.L[1][1]);
    h0 = make_uint4(e.H[0][0], e.H[0][1], e.H[1][0], e.H[1][1]); l0 = make_uint4(e.L[0][0], e.L[0][1], e.L[1][0], e.L[1][1]);
    park1(h0, l0);
  } else if (sl == 10) {
    sw(e.H[2][0], e.H[3][0]); sw(e.H[2][1], e.H[3][1]);
  } else if (sl == 11) {
    sw(e.L[2][0], e.L[3][0]); sw(e.L[2][1], e.L[3][1]);
    h1 = make_uint4(e.H[2][0], e.H[2][1], e.H[3][0], e.H[3][1]); l1 = make_uint4(e.L[2][0], e.L[2][1], e.L[3][0], e.L[3][1]);
    park1(h1, l1);
  }
}
__device__ __forceinline__ void epi_all(const f32x16& acc, const f32x16& acc2, float floor_v, uint4& h0, uint4& l0, uint4& h1,
                                        uint4& l1) {
  Epi e;
  f32x16 sum;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum[r] = acc[r] + acc2[r];
#pragma unroll
  for (int sl = 0; sl < EPI_SLICES; ++sl) epi_slice(e, sum, floor_v, sl, h0, l0, h1, l1);
}

__device__ __forceinline__ void acc_init(f32x16& acc, const float* bias /*LDS*/, int lane) {   // bias of 32 rows
  const float4* b = reinterpret_cast<const float4*>(bias + 4 * (lane >> 5));
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 q = b[2 * g];
    acc[4 * g] = q.x; acc[4 * g + 1] = q.y; acc[4 * g + 2] = q.z; acc[4 * g + 3] = q.w;
  }
}

// value of encoding channel c0 (lanes with kb == 0) / c1 (kb == 1) of a 3-vector; channels >= NCH are zero padding
template <int NCH>
__device__ __forceinline__ float pe_channel(const float (&x)[3], int c0, int c1, int kb) {
  // both candidates are compile-time; evaluate ONE sincos on the selected argument
  auto arg = [&](int c) -> float {
    if (c < 3) return x[c];
    if (c >= NCH) return 0.f;
    const int q = c - 3, k = q / 6, d = (q % 6) % 3;
    return fmul(x[d], (float)(1 << k));
  };
  auto mode = [&](int c) -> int { return (c < 3) ? 0 : (c >= NCH ? 3 : (((c - 3) % 6) >= 3 ? 2 : 1)); };   // 0 id, 1 sin, 2 cos, 3 zero
  const float a = kb ? arg(c1) : arg(c0);
  const int md = kb ? mode(c1) : mode(c0);
  float sn, cs;
  sincosf(a, &sn, &cs);
  return md == 0 ? a : (md == 1 ? sn : (md == 2 ? cs : 0.f));
}

struct TPipe {        // weight slab pipeline state (uniform across the workgroup)
  char* smem;
  int cur;            // slot holding the slab being consumed
  int wave, lane;
};
// Called once per slab by every wave: start the DMA of the slab after the current one into the other slot (all
// waves finished reading it at the previous hand-over), ...
__device__ __forceinline__ void pipe_prefetch(TPipe& p, const uint4* __restrict__ next, int n_u4) {
  slab_dma(p.smem + (p.cur ^ 1) * (T_SLOT_U4 * 16), next, n_u4, p.wave, p.lane);
}
__device__ __forceinline__ void pipe_piece(TPipe& p, const uint4* __restrict__ next, int n_u4, int i) {
  dma_piece(p.smem + (p.cur ^ 1) * (T_SLOT_U4 * 16), next, n_u4, i, p.wave, p.lane);
}
// ... and after consuming the current slab wait for it (__syncthreads drains vmcnt) and swap.
__device__ __forceinline__ void pipe_advance(TPipe& p) {
  __syncthreads();
  p.cur ^= 1;
}

// Park a layer's input fragments in AccVGPRs: they are only read as MFMA B operands from here on, and the arch
// VGPRs are needed for the output fragments under construction and the A-operand prefetch.
template <int N>
__device__ __forceinline__ void park(uint4 (&h)[N], uint4 (&l)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    asm volatile("" : "+a"(h[i].x), "+a"(h[i].y), "+a"(h[i].z), "+a"(h[i].w));
    asm volatile("" : "+a"(l[i].x), "+a"(l[i].y), "+a"(l[i].z), "+a"(l[i].w));
  }
}

// One 256-wide layer: 8 slabs, input fragments I (16 k-steps) [+ PE fragments first when has_pe], output
// fragments O.  Wl = this layer's slabs, slab_u4 = uint4 per slab, nextW / next_u4 = first slab after this layer.
template <int PE_KS>
__device__ __forceinline__ void layer256(TPipe& p, const uint4* __restrict__ Wl, int slab_u4, const uint4* __restrict__ nextW,
                                         int next_u4, const float* bias /*LDS*/, float floor_v, bool has_pe,
                                         const uint4* pe_stash /*LDS, this thread's 8 uint4*/, uint4 (&Ih)[16],
                                         uint4 (&Il)[16], uint4 (&Oh)[16], uint4 (&Ol)[16], int lane) {
  park<16>(Ih, Il);
  f32x16 acc, acc2, prev;   // prev = finished sum of the previous tile, consumed by the epilogue slices
  Epi e;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const uint4* nsrc = (t < 7) ? Wl + (int64_t)(t + 1) * slab_u4 : nextW;
    const int nu4 = (t < 7) ? slab_u4 : next_u4;
    acc_init(acc, bias + 32 * t, lane);
    acc_zero(acc2);
    const char* slot = p.smem + p.cur * (T_SLOT_U4 * 16);
    // previous tile's epilogue slices ride on k-steps 1,3,5,7,9,11; the next slab's DMA pieces on k-steps 0..9
    auto fill = [&](int ks, int w) {
      if (w == 2) pipe_piece(p, nsrc, nu4, ks);                       // next slab's DMA pieces: k-steps 0..9
      if (t > 0 && w < 2 && ks >= 2 && 2 * (ks - 2) + w < EPI_SLICES)   // previous tile's epilogue: k-steps 2..7
        epi_slice(e, prev, floor_v, 2 * (ks - 2) + w, Oh[2 * t - 2], Ol[2 * t - 2], Oh[2 * t - 1], Ol[2 * t - 1]);
    };
    auto getI = [&](int ks, int part) -> const uint4& { return part ? Il[ks] : Ih[ks]; };
    if (PE_KS > 0 && has_pe) {
      uint4 Ph[PE_KS > 0 ? PE_KS : 1], Pl[PE_KS > 0 ? PE_KS : 1];
#pragma unroll
      for (int i = 0; i < PE_KS; ++i) { Ph[i] = pe_stash[i * TTHR]; Pl[i] = pe_stash[(4 + i) * TTHR]; }
      auto getP = [&](int ks, int part) -> const uint4& { return part ? Pl[ks] : Ph[ks]; };
      tile_core<PE_KS>(acc, acc2, slot, 0, lane, getP, [&](int, int) {});
      tile_core<16>(acc, acc2, slot, PE_KS, lane, getI, fill);
    } else {
      tile_core<16>(acc, acc2, slot, 0, lane, getI, fill);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[r] = acc[r] + acc2[r];
    pipe_advance(p);
  }
#pragma unroll
  for (int sl = 0; sl < EPI_SLICES; ++sl) epi_slice(e, prev, floor_v, sl, Oh[14], Ol[14], Oh[15], Ol[15]);
}

__global__ void __launch_bounds__(TTHR, 1)
mlp_fwd_t_kernel(int64_t P, int S, const float* __restrict__ rays, const float* __restrict__ zv,
                 const uint4* __restrict__ pk, float* __restrict__ raw, TOff off) {
  extern __shared__ __attribute__((aligned(16))) char tsm[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, kb = lane >> 5;
  const int64_t ntiles = (P + TW - 1) / TW;
  const float NEG = -3.0e38f;
  float* btab = reinterpret_cast<float*>(tsm + T_OFF_BIAS);            // [L0..L7 | F | V | ba | br]
  uint4* stash = reinterpret_cast<uint4*>(tsm + T_OFF_PE) + tid;        // entries i*TTHR: hi 0..3, lo 4..7

  TPipe pipe;
  pipe.smem = tsm;
  pipe.cur = 0;
  pipe.wave = wave;
  pipe.lane = lane;
  {
    const float* bsrc = reinterpret_cast<const float*>(pk + off.bias);
    for (int i = tid; i < T_BIAS_FLOATS; i += TTHR) btab[i] = bsrc[i];
    slab_dma(tsm, pk + off.off[0], 512, wave, lane);   // first slab of the first tile
  }
  __syncthreads();

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t pp = tile * TW + wave * 32 + m;
    const bool valid = pp < P;
    if (!valid) pp = P - 1;
    const int64_t ray = pp / S;
    const float* rr = rays + ray * 11;
    float vd[3];
    uint4 Ah[16], Al[16], Bh[16], Bl[16];
    {
      // ---- positional encoding of this lane's point, straight into B fragments (channels 16ks + 8kb + j) ----
      uint4 Ph[4], Pl[4];
      const float zz = zv[pp];
      float x[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { x[c] = fadd(rr[c], fmul(rr[3 + c], zz)); vd[c] = rr[8 + c]; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pe_channel<63>(x, 16 * ks + j, 16 * ks + 8 + j, kb);
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t_split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
        Ph[ks] = make_uint4(h[0], h[1], h[2], h[3]);
        Pl[ks] = make_uint4(l[0], l[1], l[2], l[3]);
        stash[ks * TTHR] = Ph[ks];          // kept for the skip connection into L5
        stash[(4 + ks) * TTHR] = Pl[ks];
      }
      // ---- L0: PE (4 k-steps) -> B ----------------------------------------------------------------
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < 7) pipe_prefetch(pipe, pk + off.off[0] + (t + 1) * 512, 512);
        else pipe_prefetch(pipe, pk + off.off[1], 2048);
        f32x16 acc, acc2;
        acc_init(acc, btab + 32 * t, lane);
        acc_zero(acc2);
        tile_core<4>(acc, acc2, pipe.smem + pipe.cur * (T_SLOT_U4 * 16), 0, lane,
                     [&](int ks, int part) -> const uint4& { return part ? Pl[ks] : Ph[ks]; }, [&](int, int) {});
        epi_all(acc, acc2, 0.f, Bh[2 * t], Bl[2 * t], Bh[2 * t + 1], Bl[2 * t + 1]);
        pipe_advance(pipe);
      }
    }
    // ---- L1..L7 and the feature layer, two per iteration: B -> A, A -> B ------------------------------
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      const int la = 1 + 2 * it, lb = 2 + 2 * it;   // lb == 8: feature layer
      layer256<4>(pipe, pk + off.off[la], (la == 5) ? 2560 : 2048, pk + off.off[lb], 2048, btab + 256 * la, 0.f, la == 5,
                  stash, Bh, Bl, Ah, Al, lane);
      const uint4* nxt = (lb == 8) ? pk + off.off[9] : pk + off.off[lb + 1];
      const int nxt_u4 = (lb + 1 == 5) ? 2560 : 2048;
      layer256<0>(pipe, pk + off.off[lb], 2048, nxt, nxt_u4, btab + 256 * lb, (lb == 8) ? NEG : 0.f, false, stash, Ah, Al,
                  Bh, Bl, lane);
    }
    // here: A = h7, B = feature, current slab = alpha / sigma head (row 0)
    float alpha_val;
    {
      pipe_prefetch(pipe, pk + off.off[10], 2304);
      park<16>(Ah, Al);
      f32x16 acc, acc2;
      acc_zero(acc);
      acc_zero(acc2);
      tile_core<16>(acc, acc2, pipe.smem + pipe.cur * (T_SLOT_U4 * 16), 0, lane,
                    [&](int ks, int part) -> const uint4& { return part ? Al[ks] : Ah[ks]; }, [&](int, int) {});
      alpha_val = (acc[0] + acc2[0]) + btab[2432];   // row 0 lives in lanes 0..31, register 0
      pipe_advance(pipe);
    }
    {
      // ---- view-direction encoding (27 channels padded to 32 = 2 k-steps) ------------------------------
      uint4 Ph[2], Pl[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pe_channel<27>(vd, 16 * ks + j, 16 * ks + 8 + j, kb);
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t_split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
        Ph[ks] = make_uint4(h[0], h[1], h[2], h[3]);
        Pl[ks] = make_uint4(l[0], l[1], l[2], l[3]);
      }
      // ---- view layer: [vpe (2 k-steps) | feature (16)] -> 128, ReLU -> A[0..7] -----------------------
      park<16>(Bh, Bl);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < 3) pipe_prefetch(pipe, pk + off.off[10] + (t + 1) * 2304, 2304);
        else pipe_prefetch(pipe, pk + off.off[11], 1024);
        f32x16 acc, acc2;
        acc_init(acc, btab + 2304 + 32 * t, lane);
        acc_zero(acc2);
        const char* slot = pipe.smem + pipe.cur * (T_SLOT_U4 * 16);
        tile_core<2>(acc, acc2, slot, 0, lane, [&](int ks, int part) -> const uint4& { return part ? Pl[ks] : Ph[ks]; },
                     [&](int, int) {});
        tile_core<16>(acc, acc2, slot, 2, lane, [&](int ks, int part) -> const uint4& { return part ? Bl[ks] : Bh[ks]; },
                      [&](int, int) {});
        epi_all(acc, acc2, 0.f, Ah[2 * t], Al[2 * t], Ah[2 * t + 1], Al[2 * t + 1]);
        pipe_advance(pipe);
      }
    }
    // ---- rgb head slab (rows 0..2) ------------------------------------------------------------------
    {
      const bool more = tile + gridDim.x < ntiles;
      if (more) pipe_prefetch(pipe, pk + off.off[0], 512);   // next tile's first slab
      f32x16 acc, acc2;
      acc_zero(acc);
      acc_zero(acc2);
      tile_core<8>(acc, acc2, pipe.smem + pipe.cur * (T_SLOT_U4 * 16), 0, lane,
                   [&](int ks, int part) -> const uint4& { return part ? Al[ks] : Ah[ks]; }, [&](int, int) {});
      if (kb == 0 && valid) {
        float4 o;
        o.x = (acc[0] + acc2[0]) + btab[2433]; o.y = (acc[1] + acc2[1]) + btab[2434]; o.z = (acc[2] + acc2[2]) + btab[2435];
        o.w = alpha_val;
        *reinterpret_cast<float4*>(raw + pp * 4) = o;
      }
      pipe_advance(pipe);
    }
  }
}

extern "C" int fastnerf_mlp_t_fwd(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                                  const float* packed_fwd, float* raw, fn_stream_t stream) {
  FN_CHECK_ARG((kind == 0 || kind == 1) && n >= 0 && S >= 1, "kind in {0,1}, n>=0, S>=1");
  FN_CHECK_ARG(n == 0 || (rays11 && z && params && packed_fwd && raw), "null pointer");
  if (n == 0) return 0;
  const TOff O = t_offsets();
  const int64_t P = n * S;
  const int64_t ntiles = (P + TW - 1) / TW;
  int grid = t_num_cus();
  if (ntiles < grid) grid = (int)ntiles;
  static bool attr_done = false;
  if (!attr_done) {
    FN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_t_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               T_LDS_BYTES));
    attr_done = true;
  }
  hipLaunchKernelGGL(mlp_fwd_t_kernel, dim3(grid), dim3(TTHR), T_LDS_BYTES, fn::S(stream), P, S, rays11, z,
                     reinterpret_cast<const uint4*>(packed_fwd), raw, O);
  FN_LAUNCH_CHECK();
  return 0;
}
