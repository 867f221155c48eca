// x6_tile_proto.hip -- timing study (not product code): does another WAVE TILING of the bf16x6 hidden layer beat the product's?
//
// the bf16x6 forward / dX split every activation tile 4x redundantly (the four waves of a workgroup own 64 columns each and
// all read + split the same 64 rows).  Candidates that cut the redundancy WITHOUT a third LDS layout:
//   P   the product:   tile 64 points,  4 waves = 1(M) x 4(N), wave = 64 x 64,  two workgroups per CU (2 x 64 KiB of fp32 activations)
//   P1  the product's tiling with ONE workgroup per CU (LDS padded): what the second, independent workgroup is worth
//   B   tile 128 points, 4 waves = 2(M) x 2(N), wave = 64 x 128, one workgroup per CU (128 KiB): every row tile is split by TWO waves instead of
//       four (half the split VALU and half the LDS A-reads per MFMA), the weight stream per MFMA is unchanged (the two M-waves of a column half
//       both load it), 128 accumulator + 2 x 96 weight-piece registers per wave (one wave per SIMD: 512 registers available)
//   P32 the r04 product: P's tiling on v_mfma_f32_32x32x16_bf16 (k-steps of 16, gemm_seg6p: removed from csrc/ in 909934e, kept below for this study)
//   B32 B on the 32 x 32 x 16 shape: ONE wave per SIMD cannot issue the 16-cycle 16 x 16 x 32 instructions back to back (r04_power_limit section 5:
//       1442 vs 2244 TFLOP/s for one vs two waves per SIMD), the 32-cycle shape it can -- at 12.7 % less matrix throughput at the power limit
// The kernels are built from the PRODUCT's templates (csrc/mlp_common.h: gemm<MM_X6, NT, ...> = gemm_seg16, load_bias / init_acc / epilogue_fwd),
// so P here is the product's hidden layer minus saving, masks and the non-hidden phases; L hidden 256 x 256 layers run over P points with
// packed random weights (exact three-piece split on the host, the product's fragment order) and the ms per layer-tile-pass is printed.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/micro/x6_tile_proto.hip -o tools/micro/bin/x6_tile_proto
//   tools/micro/bin/x6_tile_proto [points = 786432] [layers = 8] [reps = 5]
#include "../../fast-learning-nerf_amd/csrc/mlp_common.h"

#include <math.h>
#include <string.h>
#include <vector>

namespace fn { void set_error(const char*, ...) {} }

template <int TMP, int WM, int WN, int NT, int WGPC>
__global__ void __launch_bounds__(WM * WN * 64, WGPC * WM * WN / 4)
hidden_kernel(int64_t P, int L, const float* __restrict__ packed, const float* __restrict__ bias, float* __restrict__ out) {
  static_assert(WN * NT * 32 == 256 && WM * 64 == TMP, "the waves tile TMP x 256");
  extern __shared__ __attribute__((aligned(16))) float Hs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int64_t ntiles = P / TMP;
  float sink = 0.f;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // "phase A": the tile's input (uniform(-1, 1) from a hash: real-looking data for the power management)
    for (int i = tid; i < TMP * 256; i += WM * WN * 64) {
      unsigned h = (unsigned)(tile * 131071 + i) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      Hs[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
      f32x4m acc[4][2 * NT];
      float bv[2 * NT];
      load_bias<NT>(bv, bias + l * 256, wn, lane);
      init_acc<NT, true>(acc, bv);
      gemm<MM_X6, NT, 0>(acc, Hs, 0, 32, wblock<MM_X6>(packed, (int64_t)l * 65536), 32, 0, wn * NT, wm, lane);
      __syncthreads();   // every wave has finished reading H
      epilogue_fwd<NT, true, false, true>(acc, bv, Hs, wm, wn, lane, nullptr, 256, TMP);
      __syncthreads();
    }
    sink += Hs[tid];
    __syncthreads();
  }
  if (sink == 1.2345e-30f) out[tid] = sink;
}


// ---- the r04 product's 32 x 32 x 16 k-loop (gemm_seg6p; removed from csrc/ when the 16 x 16 x 32 shape won, commit 909934e), without its saving tail ----
__device__ __forceinline__ void split3_pair_p(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  l = cvt_pk_bf16(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
struct Pieces { unsigned v[3][2][4]; };   // [piece h | m | l][row tile][pair of k]
__device__ __forceinline__ uint4 piece_frag(const Pieces& p, int pl, int mt) {
  return make_uint4(p.v[pl][mt][0], p.v[pl][mt][1], p.v[pl][mt][2], p.v[pl][mt][3]);
}
__device__ __forceinline__ void split_one_pair(const float4 (&ar)[2][2], Pieces& pn, int pair) {
  const int mt = pair >> 2, q = pair & 3;
  const float4& s = ar[mt][q >> 1];
  const float x0 = (q & 1) ? s.z : s.x, x1 = (q & 1) ? s.w : s.y;
  split3_pair_p(x0, x1, pn.v[0][mt][q], pn.v[1][mt][q], pn.v[2][mt][q]);
}
template <int NT, typename RF>
__device__ __forceinline__ void stage6(f32x16 (&acc)[2][NT], const Pieces& pc, const uint4 (&b)[NT][3], float4 (&ar)[2][2], Pieces& pn, RF&& refill) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  constexpr int NM = 12 * NT;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int i = (t * 2 + mt) * NT + nt;
        acc[mt][nt] = mfma_bf16(piece_frag(pc, PA[t], mt), b[nt][PB[t]], acc[mt][nt]);
#pragma unroll
        for (int pair = (i * 8) / NM; pair < ((i + 1) * 8) / NM; ++pair) {
          split_one_pair(ar, pn, pair);
          if ((pair & 3) == 3) refill(pair >> 2);
        }
      }
  interleave6<0, NM, 8 * 11>();
}
template <int NT>
__device__ __forceinline__ void gemm_seg6p(f32x16 (&acc)[2][NT], const float* __restrict__ As, int nks, const uint4* __restrict__ Bp, int KS, int nt0, int wm, int lane) {
  asm volatile("" : "+v"(lane));
  const int lrow = lane & 31, kb = lane >> 5;
  const float* arow[2];
  int axor[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = wm * 64 + mt * 32 + lrow;
    arow[mt] = As + m * 256;
    axor[mt] = m & 15;
  }
  const char* bptr[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bptr[nt] = reinterpret_cast<const char*>(Bp + ((int64_t)(nt0 + nt) * KS) * 192);
  unsigned blane = (unsigned)lane * 16u;
  auto load_a1 = [&](float4 (&a)[2][2], int mt, int ks) {
#pragma unroll
    for (int j = 0; j < 2; ++j) a[mt][j] = *reinterpret_cast<const float4*>(arow[mt] + (((ks * 4 + kb * 2 + j) ^ axor[mt]) << 2));
  };
  auto load_b = [&](uint4 (&b)[NT][3], int ks) {
    asm volatile("" : "+v"(blane));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[nt][pl] = *reinterpret_cast<const uint4*>((bptr[nt] + (ks * 192 + pl * 64) * 16) + blane);
  };
  float4 ar[2][2];
  Pieces p0, p1;
  uint4 b0[NT][3], b1[NT][3];
  load_b(b0, 0);
  load_a1(ar, 0, 0); load_a1(ar, 1, 0);
  load_b(b1, 1);
#pragma unroll
  for (int pair = 0; pair < 8; ++pair) {
    split_one_pair(ar, p0, pair);
    if ((pair & 3) == 3) load_a1(ar, pair >> 2, 1);
  }
  __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
  for (int ks = 0; ks < nks - 2; ks += 2) {
    const int k2 = ks + 2, k3 = ks + 3;
    stage6<NT>(acc, p0, b0, ar, p1, [&](int mt) { load_a1(ar, mt, k2); });
    load_b(b0, k2);
    stage6<NT>(acc, p1, b1, ar, p0, [&](int mt) { load_a1(ar, mt, k3); });
    load_b(b1, k3);
  }
  stage6<NT>(acc, p0, b0, ar, p1, [](int) {});
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(piece_frag(p1, PA[t], mt), b1[nt][PB[t]], acc[mt][nt]);
  __builtin_amdgcn_s_setprio(0);
}

template <int TMP, int WM, int WN, int NT, int WGPC>
__global__ void __launch_bounds__(WM * WN * 64, WGPC * WM * WN / 4)
hidden32_kernel(int64_t P, int L, const float* __restrict__ packed, const float* __restrict__ bias, float* __restrict__ out) {
  static_assert(WN * NT * 32 == 256 && WM * 64 == TMP, "the waves tile TMP x 256");
  extern __shared__ __attribute__((aligned(16))) float Hs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int64_t ntiles = P / TMP;
  float sink = 0.f;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int i = tid; i < TMP * 256; i += WM * WN * 64) {
      unsigned h = (unsigned)(tile * 131071 + i) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      Hs[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
      f32x16 acc[2][NT];
      float bv[NT];
      load_bias<NT>(bv, bias + l * 256, wn, lane);
      zero_acc<NT>(acc);
      gemm_seg6p<NT>(acc, Hs, 16, reinterpret_cast<const uint4*>(packed) + (int64_t)l * 65536 * 3 / 8, 16, wn * NT, wm, lane);
      __syncthreads();
      epilogue_fwd<NT, true, false, false>(acc, bv, Hs, wm, wn, lane, nullptr, 256, TMP);
      __syncthreads();
    }
    sink += Hs[tid];
    __syncthreads();
  }
  if (sink == 1.2345e-30f) out[tid] = sink;
}

// host-side exact split x = h + m + l into bf16 pieces (round to nearest even at every level), as csrc/mlp_common.h split3_pair
static unsigned short bf16_rne(float x) {
  unsigned u; memcpy(&u, &x, 4);
  const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(r >> 16);
}
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int TMP, int WM, int WN, int NT, int WGPC, bool S32 = false>
static float run(const char* name, int64_t P, int L, int reps, int lds_bytes, const float* packed, const float* bias, float* out, int ncu) {
  auto k = S32 ? hidden32_kernel<TMP, WM, WN, NT, WGPC> : hidden_kernel<TMP, WM, WN, NT, WGPC>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) { printf("%s: attr failed\n", name); return -1.f; }
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k));
  const int grid = ncu * WGPC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), lds_bytes, 0, P, L, packed, bias, out);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(WM * WN * 64), lds_bytes, 0, P, L, packed, bias, out);
  hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return -1.f; }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = 2.0 * 256 * 256 * (double)P * L;
  printf("%-3s tile %3d, %d(M) x %d(N) waves of 64 x %3d, %d WG/CU, LDS %6d B, %3d VGPRs (+%d spilled): %7.3f ms = %.4f ms / layer, %6.1f TFLOP/s algorithmic = %.3f of 416.7\n",
         name, TMP, WM, WN, NT * 32, WGPC, lds_bytes, fa.numRegs, (int)(fa.localSizeBytes / 4), ms, ms / L, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 416.7);
  return ms;
}

int main(int argc, char** argv) {
  const int64_t P = argc > 1 ? atoll(argv[1]) : 786432;
  const int L = argc > 2 ? atoi(argv[2]) : 8;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  // packed weights: per layer 16 column tiles x 8 k-steps x 3 planes x 64 lanes of uint4 (pack6_kernel's forward order on the 16 x 16 x 32 shape)
  // packed weights, both fragment orders: per layer [tile of TW columns][k-step of KW][plane h | m | l][lane][4 x (2 bf16)], lane = (column l % TW, k-chunk
  // l / TW of 8); 16 x 16 x 32 shape: TW 16, KW 32 (pack6_kernel's forward order); 32 x 32 x 16 shape: TW 32, KW 16 (the r04 order)
  std::vector<unsigned> pk((size_t)L * 65536 * 3 / 2), pk32(pk.size());
  std::vector<float> bias((size_t)L * 256);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; };
  for (int l = 0; l < L; ++l) {
    std::vector<float> W(256 * 256);
    for (auto& w : W) w = rnd() * 0.108f;                     // ~ uniform(+-sqrt(3 / 256)): activations keep their scale through ReLU layers
    for (int n = 0; n < 256; ++n) bias[l * 256 + n] = rnd() * 0.05f;
    for (int shape = 0; shape < 2; ++shape) {
      const int TW = shape ? 32 : 16, KW = shape ? 16 : 32, NTL = 256 / TW, KS = 256 / KW;
      std::vector<unsigned>& dst = shape ? pk32 : pk;
      for (int tile = 0; tile < NTL; ++tile)
        for (int ks = 0; ks < KS; ++ks)
          for (int ln = 0; ln < 64; ++ln) {
            unsigned* o = dst.data() + (((size_t)l * NTL * KS + tile * KS + ks) * 3 * 64 + ln) * 4;
            for (int q = 0; q < 4; ++q) {
              unsigned pc[3] = {0, 0, 0};
              for (int e = 0; e < 2; ++e) {
                const int kp = ks * KW + (ln / TW) * 8 + 2 * q + e, n = tile * TW + (ln % TW);
                float x = W[n * 256 + kp];
                for (int pl = 0; pl < 3; ++pl) { const unsigned short b = bf16_rne(x); x -= bf16_f(b); pc[pl] |= (unsigned)b << (16 * e); }
              }
              o[q] = pc[0]; o[64 * 4 + q] = pc[1]; o[128 * 4 + q] = pc[2];
            }
          }
    }
  }
  float *d_pk, *d_pk32, *d_bias, *d_out;
  hipMalloc(&d_pk, pk.size() * 4); hipMalloc(&d_pk32, pk.size() * 4); hipMalloc(&d_bias, bias.size() * 4); hipMalloc(&d_out, 4096);
  hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_pk32, pk32.data(), pk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
  printf("%lld points, %d hidden layers, %d CUs (%s)\n", (long long)P, L, ncu, prop.name);
  for (int round = 0; round < 2; ++round) {
    run<64, 1, 4, 2, 2>("P", P, L, reps, 64 * 256 * 4 + 16384, d_pk, d_bias, d_out, ncu);      // (+ the 16 KiB of the product's encoding tile: same occupancy)
    run<64, 1, 4, 2, 1>("P1", P, L, reps, 128 * 256 * 4, d_pk, d_bias, d_out, ncu);
    run<128, 2, 2, 4, 1>("B", P, L, reps, 128 * 256 * 4, d_pk, d_bias, d_out, ncu);
    run<64, 1, 4, 2, 2, true>("P32", P, L, reps, 64 * 256 * 4 + 16384, d_pk32, d_bias, d_out, ncu);      // the r04 product's shape and tiling
    run<128, 2, 2, 4, 1, true>("B32", P, L, reps, 128 * 256 * 4, d_pk32, d_bias, d_out, ncu);             // 2(M) x 2(N) waves of 64 x 128 on the shape ONE wave per SIMD can issue back to back
  }
  return 0;
}
