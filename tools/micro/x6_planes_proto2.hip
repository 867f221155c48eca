// Prototype 2 (timing study, not product code): bf16x6 hidden layers with every activation split ONCE, in the epilogue that
// produces it, kept in LDS as three bf16 planes -- as x6_planes_proto.hip -- but with the product's wave tile: 4 waves per CU
// (one per SIMD, 512 registers each), each wave 64 output channels x 64 points = 2 x 2 accumulators, so that a k-step is
// 24 MFMAs for 6 weight loads (the product's weight bytes per MFMA) and 6 ds_read_b128 (1.5x the product's LDS bytes), and NO
// split arithmetic in the k-loop.  Round-4 finding behind it (profiles/r04_power_limit.md): the product's k-loops hide the split
// in CYCLES once it is interleaved with the MFMAs, but not in ENERGY -- the chip clocks down by what the VALU work burns.
//
// TRANSPOSED product (A = weight pieces, rows = output channels; B = activation pieces, columns = points): a lane's accumulator
// registers are 4-channel runs of ONE point, so the epilogue (bias, ReLU, split) writes 8-byte k-contiguous pieces.
// SAVE: the fp32 activations go out through a per-wave 8 KiB staging block (64 points x 32 channels, 16-byte chunks XOR-swizzled),
// read back row-wise: 128-byte row segments, 16 bytes per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o x6_planes_proto2 x6_planes_proto2.hip [-DPF=3] [-DSPLIT_DOT2=1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifndef SGB
#define SGB 1
#endif
#ifndef PF
#define PF 2       // weight fragments PF k-steps ahead (PF + 1 register sets)
#endif

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  l = cvt_pk_bf16(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int TM = 64, PLANE = TM * 512;   // bytes per plane
constexpr int STG = 8192;                  // staging bytes per wave
// byte offset of the 16-byte chunk `ch` (8 channels) of point p inside a plane
__device__ __forceinline__ int chunk_off(int p, int ch) { return p * 512 + ((ch ^ (p & 31)) << 4); }

template <int SAVE>
__global__ void __launch_bounds__(256, 1)
proto(int64_t P, int L, const uint4* __restrict__ wpack, const float* __restrict__ bias, float* __restrict__ save, float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  unsigned long long tk = 0, te = 0, tb = 0;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pcol = lane & 31, kb = lane >> 5;
  char* stg = lds + 3 * PLANE + wave * STG;
  const int64_t ntiles = P / TM;
  float sink = 0.f;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int i = tid; i < 3 * PLANE / 16; i += 256) {
      const unsigned x = (unsigned)(i * 2654435761u + tile * 40503u);
      reinterpret_cast<uint4*>(lds)[i] = make_uint4((x & 0x007f007f) | 0x3c003c00, ((x >> 3) & 0x007f007f) | 0x3c003c00,
                                                    ((x >> 5) & 0x007f007f) | 0x3c003c00, ((x >> 7) & 0x007f007f) | 0x3c003c00);
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      // [layer][channel tile][ks][plane][lane]
      const uint4* wp[2] = {wpack + ((int64_t)l * 8 + 2 * wave) * 16 * 192 + lane, wpack + ((int64_t)l * 8 + 2 * wave + 1) * 16 * 192 + lane};
      f32x16 acc[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
      float4 bv[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[c][g] = *reinterpret_cast<const float4*>(bias + l * 256 + wave * 64 + c * 32 + 8 * g + 4 * kb);
      constexpr int NS = PF + 1;
      uint4 aw[NS][2][3];
      auto load_w = [&](uint4 (&a)[2][3], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a[c][pl] = wp[c][ks * 192 + pl * 64];
      };
#pragma unroll
      for (int q = 0; q < PF; ++q) load_w(aw[q], q);
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
      auto load_b = [&](uint4 (&b)[2][3], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            b[t][pl] = *reinterpret_cast<const uint4*>(lds + pl * PLANE + chunk_off(t * 32 + pcol, ks * 2 + kb));
      };
      auto mm = [&](const uint4 (&a)[2][3], const uint4 (&b)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[c][t] = mfma(a[c][PA[q]], b[t][PB[q]], acc[c][t]);
      };
      const unsigned long long s0 = __builtin_readcyclecounter();
      uint4 b0[2][3], b1[2][3];
      load_b(b0, 0);
      // 16 k-steps, straight-line code: every guard folds, the weight sets rotate with period NS, the activation sets with period 2
#pragma unroll
      for (int kq = 0; kq < 16; ++kq) {
        if (kq + PF < 16) load_w(aw[(kq + PF) % NS], kq + PF);
        if (kq & 1) { if (kq + 1 < 16) load_b(b0, kq + 1); mm(aw[kq % NS], b1); }
        else { if (kq + 1 < 16) load_b(b1, kq + 1); mm(aw[kq % NS], b0); }
#if SGB
        // one memory instruction per MFMA shadow instead of a clump of 12 in front of the k-step
#pragma unroll
        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#endif
      }
      const unsigned long long s1 = __builtin_readcyclecounter();
      __syncthreads();   // every wave has read the planes
      const unsigned long long s2 = __builtin_readcyclecounter();
      // epilogue: lane = (point t*32 + pcol, half kb); register r = channel 64 wave + 32 c + (r & 3) + 8 (r >> 2) + 4 kb
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int p = t * 32 + pcol;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4] = {acc[c][t][4 * g] + bv[c][g].x, acc[c][t][4 * g + 1] + bv[c][g].y, acc[c][t][4 * g + 2] + bv[c][g].z,
                          acc[c][t][4 * g + 3] + bv[c][g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] * 0.0883883f : 0.f;   // ReLU (+ a scale that keeps the toy net bounded)
            uint2 h, m, lo;
            split3_pair(v[0], v[1], h.x, m.x, lo.x);
            split3_pair(v[2], v[3], h.y, m.y, lo.y);
            const int off = chunk_off(p, wave * 8 + c * 4 + g) + kb * 8;
            *reinterpret_cast<uint2*>(lds + off) = h;
            *reinterpret_cast<uint2*>(lds + PLANE + off) = m;
            *reinterpret_cast<uint2*>(lds + 2 * PLANE + off) = lo;
            if (SAVE) *reinterpret_cast<float4*>(stg + p * 128 + (((2 * g + kb) ^ (p & 7)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
        if (SAVE) {   // this wave's 64 points x 32 channels, row-wise: lane = (point i*8 + (lane >> 3), 16-byte chunk lane & 7)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int p = i * 8 + (lane >> 3), ch = lane & 7;
            const float4 v = *reinterpret_cast<const float4*>(stg + p * 128 + ((ch ^ (p & 7)) << 4));
            const f32x4v tv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(tv, reinterpret_cast<f32x4v*>(save + ((int64_t)l * P + tile * TM + p) * 256 + wave * 64 + c * 32 + ch * 4));
          }
        }
      }
      const unsigned long long s3 = __builtin_readcyclecounter();
      __syncthreads();   // the next layer's input is complete
      const unsigned long long s4 = __builtin_readcyclecounter();
      tk += s1 - s0; te += s3 - s2; tb += (s2 - s1) + (s4 - s3);
    }
    sink += __uint_as_float(reinterpret_cast<const unsigned*>(lds)[tid] << 16);
  }
  if (sink == 123.f) out[tid] = sink;
  if (blockIdx.x == 7 && lane == 0) { stamps[wave * 3] = tk; stamps[wave * 3 + 1] = te; stamps[wave * 3 + 2] = tb; }
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
  unsigned long long* stamps; (void)hipMalloc(&stamps, 4096);
  (void)hipMalloc(&save, (size_t)L * P * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int ldsb = 3 * PLANE + 4 * STG;
  for (int sv = 0; sv < 2; ++sv) {
    auto kern = sv ? proto<1> : proto<0>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), ldsb, 0, P, L, wpack, bias, save, out, stamps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    unsigned long long hs[12]; (void)hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
    const double nl = (double)(P / TM / 256) * L;
    for (int w = 0; w < 4; ++w) printf("   wave %d: cycles per layer-tile: k-loop %.0f  epilogue %.0f  barriers %.0f\n", w, hs[3*w] / nl, hs[3*w+1] / nl, hs[3*w+2] / nl);
    const double flop = 2.0 * 65536 * (double)P * L;
    printf("proto2 PF=%d %s: P = %lld, %d layers: %.3f ms = %.3f ms / layer = %.1f algorithmic TFLOP/s (%.3f of 416.7)\n", PF,
           sv ? "saving fp32 activations" : "no saves", (long long)P, L, best, best / L, flop / best / 1e9, flop / best / 1e9 / 416.7);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
