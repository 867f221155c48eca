// Do a matrix-only wave and a VALU-only wave that share ONE SIMD overlap?  512-thread workgroups, one per CU:
// waves 0..3 run a bare v_mfma_f32_32x32x16_bf16 stream (4 accumulators), waves 4..7 (same SIMDs, younger) a bare
// stream of ONE VALU opcode over 16 independent registers (inline asm, nothing for the compiler to fold); every
// wave reports its own s_memtime span.  From the spans: VALU instructions the partner got through per MFMA while the
// matrix wave was running.  swap = 1 exchanges the roles (VALU waves older).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define OP16(STR) \
  asm volatile(STR : "+v"(r[0]), "+v"(r[1]) : "v"(c)); asm volatile(STR : "+v"(r[2]), "+v"(r[3]) : "v"(c)); \
  asm volatile(STR : "+v"(r[4]), "+v"(r[5]) : "v"(c)); asm volatile(STR : "+v"(r[6]), "+v"(r[7]) : "v"(c)); \
  asm volatile(STR : "+v"(r[8]), "+v"(r[9]) : "v"(c)); asm volatile(STR : "+v"(r[10]), "+v"(r[11]) : "v"(c)); \
  asm volatile(STR : "+v"(r[12]), "+v"(r[13]) : "v"(c)); asm volatile(STR : "+v"(r[14]), "+v"(r[15]) : "v"(c));

template <int KIND>
__device__ __forceinline__ void valu16(unsigned (&r)[16], unsigned c) {   // 16 instructions, two per asm statement
  if (KIND == 0) { OP16("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2") }
  if (KIND == 1) { OP16("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2") }
  if (KIND == 2) { OP16("v_max_f32 %0, %0, %2\n v_max_f32 %1, %1, %2") }
  if (KIND == 3) { OP16("v_cvt_pk_bf16_f32 %0, %0, %2\n v_cvt_pk_bf16_f32 %1, %1, %2") }
  if (KIND == 4) { OP16("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2") }
  if (KIND == 5) { OP16("v_perm_b32 %0, %0, %2, %2\n v_perm_b32 %1, %1, %2, %2") }
  if (KIND == 6) { OP16("v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %1, %1, %2, 31") }
  if (KIND == 7) { OP16("v_and_b32 %0, %0, %2\n v_and_b32 %1, %1, %2") }
  if (KIND == 8) { OP16("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1") }
  if (KIND == 9) { OP16("v_mov_b32 %0, %2\n v_mov_b32 %1, %2") }
}
static const char* NAMES[] = {"v_fma_f32", "v_add_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "v_add_u32", "v_perm_b32",
                              "v_alignbit_b32", "v_and_b32", "v_lshlrev_b32", "v_mov_b32"};

template <int KIND>
__global__ void __launch_bounds__(512, 1) k(long long* out, int it_m, int it_v, int swap, int same_wave) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  uint4 av = make_uint4(threadIdx.x, 1, 2, 3), bv = make_uint4(4, 5, 6, threadIdx.x);
  unsigned r[16];
  for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 77 + i;
  unsigned c = threadIdx.x | 0x3f800000u;
  __syncthreads();
  const long long t0 = clock64();
  if (swap != 2 && (w < 4) != (swap != 0)) {
    for (int it = 0; it < it_m; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[q & 3], 0, 0, 0);
        if (same_wave) { __builtin_amdgcn_sched_barrier(0); valu16<KIND>(r, c); __builtin_amdgcn_sched_barrier(0); }
      }
    }
  } else {
    for (int it = 0; it < it_v; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) valu16<KIND>(r, c);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  for (int i = 0; i < 16; ++i) s += (float)r[i];
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + w) * 2] = t1 - t0; out[(blockIdx.x * 8 + w) * 2 + 1] = (long long)s; }
}

template <int KIND>
void run(long long* out, long long* h, int it_m, int it_v, int swap, int sw, double& tm, double& tv) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, it_m, it_v, swap, sw);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, out, 256 * 8 * 2 * 8, hipMemcpyDeviceToHost);
  tm = tv = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) ((swap != 2 && (w < 4) != (swap != 0)) ? tm : tv) += (double)h[(b * 8 + w) * 2];
  tm /= 1024; tv /= (swap == 2 ? 2048 : 1024);
}

template <int KIND>
void all(long long* out, long long* h) {
  const int M = 2000, NM = M * 8;   // 16000 MFMAs
  double tm0, tv0, tm, tv, d;
  run<KIND>(out, h, M, 0, 0, 0, tm0, d);
  const int V = 2000, NV = V * 128;
  run<KIND>(out, h, 0, V, 0, 0, d, tv0);
  printf("%-18s alone: %5.1f clk/MFMA, %5.2f clk/VALU |", NAMES[KIND], tm0 / NM, tv0 / NV);
  for (int swap = 0; swap < 2; ++swap) {
    run<KIND>(out, h, M, V, swap, 0, tm, tv);
    // VALU instructions retired while the matrix wave was still running (the rest ran alone at tv0/NV)
    const double during = tv > tm ? NV - (tv - tm) / (tv0 / NV) : NV;
    printf(" %s: %5.1f clk/MFMA, partner VALU per MFMA %5.2f |", swap ? "VALU older" : "matrix older", tm / NM, during / (tv > tm ? NM : NM * tv / tm));
  }
  run<KIND>(out, h, M, 0, 0, 1, tm, d);
  printf(" same wave 16/MFMA: +%5.2f clk/VALU |", (tm / NM - tm0 / NM) / 16);
  run<KIND>(out, h, 0, V, 2, 0, d, tv);   // swap == 2: BOTH waves of a SIMD run the VALU stream
  printf(" two VALU waves: %5.2f clk/VALU each\n", tv / NV);
}

int main() {
  long long *out, *h = (long long*)malloc(256 * 8 * 2 * 8);
  (void)hipMalloc(&out, 256 * 8 * 2 * 8);
  all<0>(out, h); all<1>(out, h); all<2>(out, h); all<3>(out, h); all<4>(out, h);
  all<5>(out, h); all<6>(out, h); all<7>(out, h); all<8>(out, h); all<9>(out, h);
  return 0;
}
