// v_dot2c_f32_bf16 (VOP2, accumulates in place; what hipcc selects for __builtin_amdgcn_fdot2_f32_bf16 on gfx950) against the
// three-address v_dot2_f32_bf16 (VOP3P) in the residual x - bf16(x) of the operand split: are they the same function?
// Build: hipcc --offload-arch=gfx950 -O3 -o dot2_probe dot2_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  const f32x2v v = {x0, x1};
  const bf16x2v hv = __builtin_convertvector(v, bf16x2v);
  const unsigned hu = __builtin_bit_cast(unsigned, hv);
  unsigned c10, c01;
  asm("s_mov_b32 %0, 0xbf80" : "=s"(c10));
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(c01));
  const float ref0 = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2v, c10), x0, false);
  const float ref1 = __builtin_amdgcn_fdot2_f32_bf16(hv, __builtin_bit_cast(bf16x2v, c01), x1, false);
  float a0, a1, b0, b1, d0, d1;
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(a0) : "s"(c10), "v"(hu), "v"(x0));                      // constant in an SGPR, src0
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(a1) : "s"(c01), "v"(hu), "v"(x1));
  unsigned v10 = 0xbf80u, v01 = 0xbf800000u;
  asm("" : "+v"(v10), "+v"(v01));
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(b0) : "v"(hu), "v"(v10), "v"(x0));                      // all VGPRs
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(b1) : "v"(hu), "v"(v01), "v"(x1));
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(d0) : "v"(v10), "v"(hu), "v"(x0));                      // operands swapped
  asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(d1) : "v"(v01), "v"(hu), "v"(x1));
  const float e0 = x0 - __uint_as_float(hu << 16), e1 = x1 - __uint_as_float(hu & 0xffff0000u);       // the exact residual
  unsigned* o = out + 10 * i;
  o[0] = __float_as_uint(ref0); o[1] = __float_as_uint(ref1); o[2] = __float_as_uint(a0); o[3] = __float_as_uint(a1);
  o[4] = __float_as_uint(b0); o[5] = __float_as_uint(b1); o[6] = __float_as_uint(d0); o[7] = __float_as_uint(d1);
  o[8] = __float_as_uint(e0); o[9] = __float_as_uint(e1);
}
int main() {
  const int n = 1 << 16;
  float* hx = (float*)malloc(2 * n * 4);
  unsigned s = 1;
  for (int i = 0; i < 2 * n; ++i) { s = s * 1664525u + 1013904223u; unsigned u = (s & 0x807fffffu) | ((100 + (s >> 23) % 50) << 23); hx[i] = *(float*)&u; }
  float* dx; unsigned* dout;
  (void)hipMalloc(&dx, 2 * n * 4); (void)hipMalloc(&dout, 10 * n * 4);
  (void)hipMemcpy(dx, hx, 2 * n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  unsigned* ho = (unsigned*)malloc(10 * n * 4);
  (void)hipMemcpy(ho, dout, 10 * n * 4, hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int v = 0; v < 4; ++v)
      for (int j = 0; j < 2; ++j) if (ho[10 * i + 2 * v + j] != ho[10 * i + 8 + j]) { if (bad[v]++ < 2) printf("variant %d lane %d: x %08x h %04x got %08x exact %08x\n", v, j, *(unsigned*)&hx[2 * i + j], 0, ho[10 * i + 2 * v + j], ho[10 * i + 8 + j]); }
  printf("mismatches vs the exact residual out of %d: v_dot2c (builtin) %d, v_dot2 SGPR const %d, v_dot2 VGPR const %d, v_dot2 VGPR const, swapped %d\n", 2 * n, bad[0], bad[1], bad[2], bad[3]);
  return 0;
}
