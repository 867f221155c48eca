// (-DSHAPE16=1: the same k-loop on v_mfma_f32_16x16x32_bf16 -- 48 MFMAs of 16 cycles per 32 channels of K instead of 2 x 12 of 32 cycles;
//  timing only: the epilogue then runs on accumulators in another layout.  profiles/r04_power_limit.md, section 5.)
// Prototype (timing study, not product code): what would a bf16x6 hidden layer cost if every activation were split ONCE -- in
// the epilogue that produces it -- and kept in LDS as three bf16 planes, instead of fp32 in LDS split by each of the four waves
// that multiply it (csrc/mlp.hip, gemm_seg6)?  VALU time does not hide under the partner wave's MFMAs on this chip
// (profiles/r03_mfma_valu_exclusion.md), so the k-loop's 84 VALU per 24 MFMAs are paid in full today.
//
// Scheme: one 512-thread workgroup per CU, tile = 64 points; planes[3][64 points][256 channels] bf16 = 96 KiB (16-byte chunks XOR-
// swizzled by the point); wave w owns output channels 32 w .. 32 w + 31 for all 64 points (two point tiles = two accumulators).
// TRANSPOSED product: A operand = weight pieces (rows = output channels; the packed layout of pack6_kernel is already right),
// B operand = activation pieces (columns = points) => a lane's accumulator registers are 4-channel runs of ONE point: the epilogue
// (bias, ReLU, split3) packs channel pairs and writes 8-byte pieces that are k-contiguous for the next layer -- no transpose.
// Per k-step (16 channels) and wave: 3 global 16-byte loads (weights, double-buffered), 6 ds_read_b128, 12 MFMAs, ~4 VALU.
//
// Runs L hidden 256 x 256 layers over P points and prints ms per layer; compare with the product forward's 4.31 ms / 9.06
// layer-equivalents = 0.476 ms per layer-equivalent at P = 786 432.   Build: hipcc --offload-arch=gfx950 -O3 -o x6_planes_proto ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  unsigned c10, c01;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(c10));
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(c01));
  const bf16x2v m10 = __builtin_bit_cast(bf16x2v, c10), m01 = __builtin_bit_cast(bf16x2v, c01);
  const f32x2v v = {x0, x1};
  const bf16x2v hv = __builtin_convertvector(v, bf16x2v);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hv, m10, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(hv, m01, x1, false);
  const f32x2v rv = {r0, r1};
  const bf16x2v mv = __builtin_convertvector(rv, bf16x2v);
  const f32x2v sv = {__builtin_amdgcn_fdot2_f32_bf16(mv, m10, r0, false), __builtin_amdgcn_fdot2_f32_bf16(mv, m01, r1, false)};
  h = __builtin_bit_cast(unsigned, hv); m = __builtin_bit_cast(unsigned, mv);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(sv, bf16x2v));
}
__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifndef SHAPE16
#define SHAPE16 0
#endif
constexpr int TM = 64, PLANE = TM * 512;   // bytes per plane
// byte offset of the 16-byte chunk `ch` (8 channels) of point p inside a plane
__device__ __forceinline__ int chunk_off(int p, int ch) { return p * 512 + ((ch ^ (p & 31)) << 4); }

template <int SAVE>
__global__ void __launch_bounds__(512, 2)
proto(int64_t P, int L, const uint4* __restrict__ wpack, const float* __restrict__ bias, float* __restrict__ save, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pcol = lane & 31, kb = lane >> 5;
  const int64_t ntiles = P / TM;
  float sink = 0.f;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // "phase A": fill the planes with this tile's input (pseudo-random pieces; stands in for the positional encoding layer)
    for (int i = tid; i < 3 * PLANE / 16; i += 512) {
      const unsigned x = (unsigned)(i * 2654435761u + tile * 40503u);
      reinterpret_cast<uint4*>(lds)[i] = make_uint4((x & 0x007f007f) | 0x3c003c00, ((x >> 3) & 0x007f007f) | 0x3c003c00,
                                                    ((x >> 5) & 0x007f007f) | 0x3c003c00, ((x >> 7) & 0x007f007f) | 0x3c003c00);
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      const uint4* wp = wpack + ((int64_t)l * 8 + wave) * 16 * 192 + lane;   // [layer][channel tile][ks][plane][lane]
#if !SHAPE16
      f32x16 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      float4 bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(bias + l * 256 + wave * 32 + 8 * g + 4 * kb);
      uint4 aw[4][3];   // weight fragments, three k-steps ahead (L2 latency is longer than one 12-MFMA k-step)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) aw[q][pl] = wp[q * 192 + pl * 64];
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      auto load_b = [&](uint4 (&b)[2][3], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[t][pl] = *reinterpret_cast<const uint4*>(lds + pl * PLANE + chunk_off(t * 32 + pcol, ks * 2 + kb));
      };
      auto mm = [&](const uint4 (&a)[3], const uint4 (&b)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t] = mfma(a[PA[q]], b[t][PB[q]], acc[t]);
      };
      uint4 b0[2][3], b1[2][3];
      load_b(b0, 0);
#pragma unroll 1
      for (int ks = 0; ks < 16; ks += 4) {   // activation fragments of k-step q + 1 are fetched during the MFMAs of q
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kq = ks + u;
          if (kq + 3 < 16) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) aw[(u + 3) & 3][pl] = wp[(kq + 3) * 192 + pl * 64];
          }
          if (u & 1) { if (kq + 1 < 16) load_b(b0, kq + 1); mm(aw[u], b1); }
          else { load_b(b1, kq + 1 < 16 ? kq + 1 : kq); mm(aw[u], b0); }
        }
      }
#else
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4 c16[2][4];   // [channel tile of 16][point tile of 16]
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) c16[c][t][r] = 0.f;
      float4 bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4*>(bias + l * 256 + wave * 32 + 8 * g + 4 * kb);
      const int p16 = lane & 15, kc = lane >> 4;   // B operand: point p16 (+ 16 t), channels 8 kc .. 8 kc + 7 of the 32-channel k-step
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      uint4 aw[2][2][3];   // [set][channel tile][piece]: weight fragments of a 32-channel k-step (same bytes as two 16-channel k-steps)
      auto load_w = [&](uint4 (&a)[2][3], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a[c][pl] = wp[(2 * ks + c) * 192 + pl * 64];
      };
      auto load_b = [&](uint4 (&b)[4][3], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[t][pl] = *reinterpret_cast<const uint4*>(lds + pl * PLANE + chunk_off(t * 16 + p16, ks * 4 + kc));
      };
      auto mm = [&](const uint4 (&a)[2][3], const uint4 (&b)[4][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              c16[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[c][PA[q]]), __builtin_bit_cast(bf16x8, b[t][PB[q]]), c16[c][t], 0, 0, 0);
      };
      uint4 b0[4][3], b1[4][3];
      load_w(aw[0], 0);
      load_b(b0, 0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {   // 8 k-steps of 32 channels
        if (ks + 1 < 8) load_w(aw[(ks + 1) & 1], ks + 1);
        if (ks & 1) { if (ks + 1 < 8) load_b(b0, ks + 1); mm(aw[1], b1); }
        else { if (ks + 1 < 8) load_b(b1, ks + 1); mm(aw[0], b0); }
      }
      f32x16 acc[2];     // (timing only: hand the 32 accumulator registers to the unchanged epilogue)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = c16[t][r >> 2][r & 3];
#endif
      __syncthreads();   // every wave has read the planes
      // epilogue: lane = (point t*32 + pcol, half kb); register r = channel 32 wave + (r & 3) + 8 (r >> 2) + 4 kb
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int p = t * 32 + pcol;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4] = {acc[t][4 * g] + bv[g].x, acc[t][4 * g + 1] + bv[g].y, acc[t][4 * g + 2] + bv[g].z, acc[t][4 * g + 3] + bv[g].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] * 0.0883883f : 0.f;   // ReLU (+ a scale that keeps the toy net bounded)
          uint2 h, m, lo;
          split3_pair(v[0], v[1], h.x, m.x, lo.x);
          split3_pair(v[2], v[3], h.y, m.y, lo.y);
          const int off = chunk_off(p, wave * 4 + g) + kb * 8;
          *reinterpret_cast<uint2*>(lds + off) = h;
          *reinterpret_cast<uint2*>(lds + PLANE + off) = m;
          *reinterpret_cast<uint2*>(lds + 2 * PLANE + off) = lo;
          if (SAVE) {   // the training forward's saved activation, fp32 [point][256]
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            __builtin_nontemporal_store(o.x, save + ((int64_t)l * P + tile * TM + p) * 256 + wave * 32 + 8 * g + 4 * kb);
            __builtin_nontemporal_store(o.y, save + ((int64_t)l * P + tile * TM + p) * 256 + wave * 32 + 8 * g + 4 * kb + 1);
            __builtin_nontemporal_store(o.z, save + ((int64_t)l * P + tile * TM + p) * 256 + wave * 32 + 8 * g + 4 * kb + 2);
            __builtin_nontemporal_store(o.w, save + ((int64_t)l * P + tile * TM + p) * 256 + wave * 32 + 8 * g + 4 * kb + 3);
          }
        }
      }
      __syncthreads();   // the next layer's input is complete
    }
    sink += __uint_as_float(reinterpret_cast<const unsigned*>(lds)[tid] << 16);
  }
  if (sink == 123.f) out[tid] = sink;
}

int main(int argc, char** argv) {
  const int64_t P = argc > 1 ? atoll(argv[1]) : 786432;
  const int L = argc > 2 ? atoi(argv[2]) : 8;
  const size_t wbytes = (size_t)L * 8 * 16 * 192 * 16;
  std::vector<unsigned> hw(wbytes / 4);
  unsigned x = 12345;
  for (auto& w : hw) { x = x * 1664525u + 1013904223u; w = (x & 0x807f807fu) | 0x3d003d00u; }   // bf16 pairs around +-0.03
  uint4* wpack; float *bias, *out, *save;
  (void)hipMalloc(&wpack, wbytes); (void)hipMemcpy(wpack, hw.data(), wbytes, hipMemcpyHostToDevice);
  (void)hipMalloc(&bias, L * 256 * 4); (void)hipMemset(bias, 0, L * 256 * 4);
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&save, (size_t)L * P * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int sv = 0; sv < 2; ++sv) {
    auto kern = sv ? proto<1> : proto<0>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PLANE);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 3 * PLANE, 0, P, L, wpack, bias, save, out);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double flop = 2.0 * 65536 * (double)P * L;
    printf("%s: P = %lld, %d layers: %.3f ms = %.3f ms / layer = %.1f algorithmic TFLOP/s (%.3f of 416.7); product forward: 0.476 ms / layer-equivalent\n",
           sv ? "saving fp32 activations" : "no saves", (long long)P, L, best, best / L, flop / best / 1e9, flop / best / 1e9 / 416.7);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
