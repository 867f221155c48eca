// Does the operand-split arithmetic of the bf16x6 kernels (dependent cvt_pk / dot2c chains, ds_write_b128 of the pieces) keep its
// speed while the partner wave of the SIMD streams v_mfma_f32_32x32x16_bf16?  512-thread workgroups, one per CU: waves 0..3 =
// MFMA stream (4 accumulators, operands re-read from LDS every 12 MFMAs when LDSREAD), waves 4..7 = split stream.  Times (HIP
// events) of: MFMA waves only, split waves only, both.  "both" ~ max => they overlap; ~ sum => they do not.
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec_split coexec_split.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef LB
#define LB 2
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int DOT2>
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  if (DOT2) {
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
  } else {
    const f32x2v v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    const f32x2v rv = {r0, r1};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, bf16x2v));
    const f32x2v sv = {r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(sv, bf16x2v));
  }
}

// roles: bit0 = MFMA waves run, bit1 = split waves run.  VKIND 0: split (DOT2), 1: split (subtract form), 2: plain independent fma
template <int VKIND, int LDSREAD, int LDSWRITE, int GAP = 0>
__global__ void __launch_bounds__(512, LB) k(float* out, const uint4* __restrict__ data, int it_m, int it_v, int roles) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = data[i];
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float s = 0.f;
  const long long t0 = wall_clock64();
  const bool mfma_wave = (GAP == 12 || GAP == 13) ? w >= 4 : w < 4;
  if (mfma_wave) {
    if (!(roles & 1)) return;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = lds[i * 64 + lane]; b[i] = lds[1024 + i * 64 + lane]; }
    for (int it = 0; it < it_m; ++it) {
      if (LDSREAD) for (int i = 0; i < 3; ++i) { a[i] = lds[((it * 3 + i) * 64 + lane) & 2047]; b[i] = lds[2048 + (((it * 3 + i) * 64 + lane) & 2047)]; }
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        if (GAP == 10) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[q % 3].x), __uint_as_float(b[(q / 4) % 3].x), acc[q & 3], 0, 0, 0);
        else if (GAP == 11) { f32x4 t4 = {acc[q & 3][0], acc[q & 3][1], acc[q & 3][2], acc[q & 3][3]}; t4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[q % 3]), __builtin_bit_cast(bf16x8, b[(q / 4) % 3]), t4, 0, 0, 0); acc[q & 3][0] = t4[0]; acc[q & 3][1] = t4[1]; acc[q & 3][2] = t4[2]; acc[q & 3][3] = t4[3]; }
        else
        acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q % 3]), __builtin_bit_cast(bf16x8, b[(q / 4) % 3]), acc[q & 3], 0, 0, 0);
        if (GAP == 1) asm volatile("s_nop 0");
        if (GAP == 2) asm volatile("s_nop 7");
        if (GAP == 3) { unsigned t_; asm volatile("s_mov_b32 %0, 0" : "=s"(t_)); }
        if (GAP == 4) asm volatile("s_branch 1f\n1:");
        if (GAP == 5) asm volatile("s_setprio 0");
        if (GAP == 6) asm volatile("s_sleep 0");
        if (GAP == 7) asm volatile("s_setprio 1\n s_setprio 0");
      }
    }
    for (int a2 = 0; a2 < 4; ++a2) s += acc[a2][0] + acc[a2][7];
  } else {
    if (!(roles & 2)) return;
    if (GAP == 13 || GAP == 14) __builtin_amdgcn_s_setprio(3);
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = __uint_as_float((data[threadIdx.x + e * 512].x & 0x007fffffu) | 0x3f800000u);
    float c = __uint_as_float((data[threadIdx.x].y & 0x007fffffu) | 0x3f000000u);
    uint4 accu = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < it_v; ++it) {
      if (VKIND == 2) {
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], c, c);
      } else {
        uint4 h, m, l;
        split3_pair<VKIND == 0>(v[0], v[1], h.x, m.x, l.x); split3_pair<VKIND == 0>(v[2], v[3], h.y, m.y, l.y);
        split3_pair<VKIND == 0>(v[4], v[5], h.z, m.z, l.z); split3_pair<VKIND == 0>(v[6], v[7], h.w, m.w, l.w);
        if (LDSWRITE) {
          uint4* d = lds + 4096 + ((w - 4) * 6 + (it & 1) * 3) * 64 + lane;
          d[0] = h; d[64] = m; d[128] = l;
        } else {
          accu.x ^= h.x ^ m.y ^ l.z; accu.y ^= h.y ^ m.z ^ l.w; accu.z ^= h.z ^ m.w ^ l.x; accu.w ^= h.w ^ m.x ^ l.y;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], c, c);   // next values (8 more VALU)
      }
    }
    for (int e = 0; e < 8; ++e) s += v[e];
    s += (float)(accu.x ^ accu.y ^ accu.z ^ accu.w);
    if (LDSWRITE) s += (float)lds[4096 + lane].x;
  }
  if (s == 123.456f) out[threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out + 1024)[mfma_wave ? (w & 3) : 4 + (w & 3)] = wall_clock64() - t0;
}

static int G_NWG = 256;
template <int VKIND, int LR, int LW, int GAP = 0>
float run(float* out, const uint4* data, int it_m, int it_v, int roles) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<VKIND, LR, LW, GAP>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(G_NWG), dim3(512), 98304, 0, out, data, it_m, it_v, roles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
  }
  return best;
}
template <int VKIND, int LR, int LW, int GAP = 0>
void all(float* out, const uint4* data, const char* name) {
  const int IM = 20000;   // x 12 MFMAs
  const float tm = run<VKIND, LR, LW, GAP>(out, data, IM, 0, 1);
  // calibrate the split stream to about the MFMA time
  const float tv1 = run<VKIND, LR, LW, GAP>(out, data, 0, 20000, 2);
  const int IV = (int)(20000 * tm / tv1);
  const float tv = run<VKIND, LR, LW, GAP>(out, data, 0, IV, 2);
  const float tb = run<VKIND, LR, LW, GAP>(out, data, IM, IV, 3);
  long long sp[8]; (void)hipMemcpy(sp, out + 1024, 64, hipMemcpyDeviceToHost);
  printf("%-44s mfma-only %7.3f ms (%5.0f TF)  valu-only %7.3f ms (%d its)  both %7.3f ms   overlap %.0f %%   spans (10 ns): mfma wave %lld, valu wave %lld\n", name, tm,
         (double)G_NWG * 4 * IM * 12 * 32768.0 / tm / 1e9, tv, IV, tb, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv), sp[0], sp[4]);
}
int main(int argc, char** argv) {
  if (argc > 1) G_NWG = atoi(argv[1]);
  const int zero = argc > 2 ? atoi(argv[2]) : 0;
  printf("workgroups %d, %s operands\n", G_NWG, zero ? "zero" : "random");
  float* out; uint4* data;
  (void)hipMalloc(&out, 8192); (void)hipMalloc(&data, 8192 * 16);
  unsigned* h = (unsigned*)malloc(8192 * 16);
  unsigned x = 12345;
  for (int i = 0; i < 8192 * 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0u : ((x & 0x807f807fu) | 0x3f003f00u); }   // bf16 pairs in [0.5, 1)
  (void)hipMemcpy(data, h, 8192 * 16, hipMemcpyHostToDevice);
  all<2, 0, 0>(out, data, "independent fma | MFMA regs");
  if (argc > 3) {
    all<2, 0, 0, 1>(out, data, "fma | MFMA, s_nop 0 between");
    all<2, 0, 0, 2>(out, data, "fma | MFMA, s_nop 7 between");
    all<2, 0, 0, 3>(out, data, "fma | MFMA, s_mov between");
    all<2, 0, 0, 4>(out, data, "fma | MFMA, s_branch between");
    all<2, 0, 0, 5>(out, data, "fma | MFMA, s_setprio 0 between");
    all<2, 0, 0, 6>(out, data, "fma | MFMA, s_sleep 0 between");
    all<2, 0, 0, 7>(out, data, "fma | MFMA, s_setprio 1,0 between");
    all<2, 0, 0, 10>(out, data, "fma | MFMA f32 32x32x2");
    all<2, 0, 0, 11>(out, data, "fma | MFMA bf16 16x16x32");
    all<2, 0, 0, 12>(out, data, "fma | MFMA, VALU waves older");
    all<2, 0, 0, 13>(out, data, "fma | MFMA, VALU waves older + prio 3");
    all<2, 0, 0, 14>(out, data, "fma | MFMA, VALU waves prio 3");
    return 0;
  }
  all<0, 0, 0>(out, data, "dot2 split | MFMA regs");
  all<1, 0, 0>(out, data, "subtract split | MFMA regs");
  all<0, 0, 1>(out, data, "dot2 split + ds_write | MFMA regs");
  all<0, 1, 0>(out, data, "dot2 split | MFMA + ds_read_b128");
  all<0, 1, 1>(out, data, "dot2 split + ds_write | MFMA + ds_read_b128");
  all<2, 1, 0>(out, data, "independent fma | MFMA + ds_read_b128");
  return 0;
}
