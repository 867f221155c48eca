// Prototype (timing study, not product code; profiles/r04_f16x3_study.md section 3): the f16x3 arithmetic with every activation split ONCE.
// Two fp16 pieces are 4 bytes per value -- the size of the fp32 tile the product keeps in LDS -- so a 64-point tile's activations fit the
// same 64 KiB as planes[2][64 points][256 channels] fp16 (h | l' = (x - h) 2^12), two workgroups per CU as in the product, and the k-loop
// has NO split arithmetic at all: per k-step of 32 channels and wave 8 ds_read_b128 (activation pieces), 8 global 16-byte loads (weight
// pieces), 48 v_mfma_f32_16x16x32_f16 (4 x 4 tiles x 3 products; the cross terms in a second accumulator set).  TRANSPOSED product (A =
// weights, B = activations) so that a lane's accumulator registers are 4-channel runs of ONE point: the epilogue (bias, ReLU, split) writes
// 8-byte pieces that are k-contiguous for the next layer.  The product's wave tile: 4 waves x (64 channels x 64 points).
// Runs L hidden 256 x 256 layers over P points; compare with the product's f16x3 forward (2.6 ms / 9.06 layer-equivalents = 0.29 ms, saving 0.35 ms).
// Build: hipcc --offload-arch=gfx950 -O3 -o f16x3_planes_proto tools/micro/f16x3_planes_proto.hip
#include <hip/hip_runtime.h>
#ifndef PIN
#define PIN 0
#endif
#ifndef PRIO
#define PRIO 0
#endif
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2h(float x0, float x1, unsigned& h, unsigned& l) {
  const f32x2 v = {x0, x1};
  const f16x2 hv = __builtin_convertvector(v, f16x2);
  const f32x2 rv = {__builtin_fmaf((float)hv.x, -4096.f, x0 * 4096.f), __builtin_fmaf((float)hv.y, -4096.f, x1 * 4096.f)};
  h = __builtin_bit_cast(unsigned, hv);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, f16x2));
}
__device__ __forceinline__ f32x4 mfma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
constexpr int TM = 64, PLANE = TM * 512;   // bytes per plane
__device__ __forceinline__ int chunk_off(int p, int ch) { return p * 512 + ((ch ^ (p & 31)) << 4); }   // 16-byte chunk `ch` (8 channels) of point p

template <int SAVE>
__global__ void __launch_bounds__(256, 2)
proto(int64_t P, int L, const uint4* __restrict__ wpack, const float* __restrict__ bias, float* __restrict__ save, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, kc = lane >> 4;
  const int64_t ntiles = P / TM;
  float sink = 0.f;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int i = tid; i < 2 * PLANE / 16; i += 256) {   // stands in for the positional-encoding layer
      const unsigned x = (unsigned)(i * 2654435761u + tile * 40503u);
      reinterpret_cast<uint4*>(lds)[i] = make_uint4((x & 0x03ff03ff) | 0x38003800, ((x >> 3) & 0x03ff03ff) | 0x38003800,
                                                    ((x >> 5) & 0x03ff03ff) | 0x38003800, ((x >> 7) & 0x03ff03ff) | 0x38003800);
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
      const uint4* wp = wpack + ((int64_t)l * 4 + wave) * 8 * 4 * 2 * 64 + lane;   // [layer][wave][ks][channel tile][plane][lane]
      f32x4 acc[4][4], acc2[4][4];   // [channel tile][point tile]
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[c][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      float4 bv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = *reinterpret_cast<const float4*>(bias + l * 256 + wave * 64 + 16 * c + 4 * kc);
      uint4 aw[2][4][2];   // [buffer][channel tile][piece]
      auto load_w = [&](uint4 (&a)[4][2], int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) a[c][pl] = wp[((ks * 4 + c) * 2 + pl) * 64];
      };
      auto load_b = [&](uint4 (&b)[2], int t, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[pl] = *reinterpret_cast<const uint4*>(lds + pl * PLANE + chunk_off(t * 16 + p16, ks * 4 + kc));
      };
      load_w(aw[0], 0);
      uint4 b0[2], b1[2];
      load_b(b0, 0, 0);
#if PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) load_w(aw[(ks + 1) & 1], ks + 1);
        uint4 (&a)[4][2] = aw[ks & 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // point tile t: its pieces in b0 / b1 alternately, the next tile's fetched during its 12 MFMAs
          uint4 (&b)[2] = (t & 1) ? b1 : b0;
          uint4 (&bn)[2] = (t & 1) ? b0 : b1;
          if (t < 3) load_b(bn, t + 1, ks); else if (ks + 1 < 8) load_b(bn, 0, ks + 1);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc2[c][t] = mfma(a[c][1], b[0], acc2[c][t]);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc2[c][t] = mfma(a[c][0], b[1], acc2[c][t]);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c][t] = mfma(a[c][0], b[0], acc[c][t]);
#if PIN   // a point tile's 12 MFMAs with its 2 LDS reads and 2 of the k-step's 8 weight loads spread between them
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
        }
      }
#if PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      __syncthreads();   // every wave has read the planes
      // epilogue: lane = (point t*16 + p16, channel group kc); register r = channel 64 wave + 16 c + 4 kc + r
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int p = t * 16 + p16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s = acc[c][t][r] + acc2[c][t][r] * (1.f / 4096.f) + (&bv[c].x)[r];
            v[r] = s > 0.f ? s * 0.0883883f : 0.f;   // ReLU (+ a scale that keeps the toy net bounded)
          }
          uint2 h, lo;
          split2h(v[0], v[1], h.x, lo.x);
          split2h(v[2], v[3], h.y, lo.y);
          const int ch = wave * 64 + 16 * c + 4 * kc;             // first of the 4 channels
          const int off = chunk_off(p, ch >> 3) + (ch & 7) * 2;
          *reinterpret_cast<uint2*>(lds + off) = h;
          *reinterpret_cast<uint2*>(lds + PLANE + off) = lo;
          if (SAVE) {   // the training forward's saved activation, fp32 [point][256]: 16 bytes per lane, 64 contiguous bytes per point and tile
            float* d = save + ((int64_t)l * P + tile * TM + p) * 256 + ch;
            __builtin_nontemporal_store(v[0], d); __builtin_nontemporal_store(v[1], d + 1);
            __builtin_nontemporal_store(v[2], d + 2); __builtin_nontemporal_store(v[3], d + 3);
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
  const size_t wbytes = (size_t)L * 4 * 8 * 4 * 2 * 64 * 16;
  std::vector<unsigned> hw(wbytes / 4);
  unsigned x = 12345;
  for (auto& w : hw) { x = x * 1664525u + 1013904223u; w = (x & 0x83ff83ffu) | 0x28002800u; }   // fp16 pairs around +-0.03
  uint4* wpack; float *bias, *out, *save;
  (void)hipMalloc(&wpack, wbytes); (void)hipMemcpy(wpack, hw.data(), wbytes, hipMemcpyHostToDevice);
  (void)hipMalloc(&bias, L * 256 * 4); (void)hipMemset(bias, 0, L * 256 * 4);
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&save, (size_t)L * P * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int sv = 0; sv < 2; ++sv) {
    auto kern = sv ? proto<1> : proto<0>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PLANE);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(512), dim3(256), 2 * PLANE, 0, P, L, wpack, bias, save, out);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double flop = 2.0 * 65536 * (double)P * L;
    printf("%s: P = %lld, %d layers: %.3f ms = %.3f ms / layer = %.1f algorithmic TFLOP/s (%.3f of 833.3); product f16x3 forward: 0.29 ms / layer-equivalent (0.35 saving)\n",
           sv ? "saving fp32 activations" : "no saves", (long long)P, L, best, best / L, flop / best / 1e9, flop / best / 1e9 / 833.3);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
