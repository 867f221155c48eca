// mlp_layout.h -- parameter / packed-weight / activation layouts of the 8x256 NeRF MLP
// (model.py:8-63 with D=8, W=256, skips=[4], use_viewdirs=True, input_ch=63, input_ch_views=27).
#pragma once
#include <stdint.h>

namespace fnl {
constexpr int W = 256, WH = 128, IN_PE = 63, IN_PEP = 64, IN_V = 27, IN_VP = 32;

// ---- flat parameter buffer: model.parameters() order, native [out][in] row-major ----
constexpr int64_t L_W(int i) {  // pts_linears.i.weight
  return i == 0 ? 0
       : i <= 5 ? (int64_t)(IN_PE * W + W) + (int64_t)(i - 1) * (W * W + W)
                : (int64_t)(IN_PE * W + W) + 4 * (int64_t)(W * W + W) + (int64_t)((W + IN_PE) * W + W) +
                      (int64_t)(i - 6) * (W * W + W);
}
constexpr int L_K(int i) { return i == 0 ? IN_PE : (i == 5 ? W + IN_PE : W); }  // fan-in
constexpr int64_t L_B(int i) { return L_W(i) + (int64_t)W * L_K(i); }
constexpr int64_t V_W = L_B(7) + W;                       // views_linears.0.weight [128][283]
constexpr int64_t V_B = V_W + (int64_t)WH * (W + IN_V);
constexpr int64_t F_W = V_B + WH;                         // feature_linear.weight [256][256]
constexpr int64_t F_B = F_W + (int64_t)W * W;
constexpr int64_t A_W = F_B + W;                          // alpha_linear.weight [1][256]
constexpr int64_t A_B = A_W + W;
constexpr int64_t R_W = A_B + 1;                          // rgb_linear.weight [3][128]
constexpr int64_t R_B = R_W + 3 * WH;
constexpr int64_t N_PARAMS = R_B + 3;
static_assert(N_PARAMS == 595844, "parameter count");

// ---- packed forward weights (fragment order; see mlp.hip) ----
// layer ids: 0..7 = pts_linears, 8 = feature_linear, 9 = views_linears.0
constexpr int PF_KP(int l) { return l == 0 ? 64 : (l == 5 ? 320 : (l == 9 ? 288 : 256)); }
constexpr int PF_N(int l) { return l == 9 ? 128 : 256; }
constexpr int64_t PF_OFF(int l) {
  int64_t o = 0;
  for (int i = 0; i < l; ++i) o += (int64_t)PF_KP(i) * PF_N(i);
  return o;
}
constexpr int64_t PF_TOTAL = PF_OFF(10);
static_assert(PF_TOTAL == 593920, "packed fwd size");

// ---- packed transposed weights for dX (fragment order) ----
// ids: 0 = views(feat part) K=128; 1 = feature; 2..8 = pts_linears 7,6,5(h part),4,3,2,1
constexpr int PB_K(int j) { return j == 0 ? 128 : 256; }
constexpr int64_t PB_OFF(int j) {
  int64_t o = 0;
  for (int i = 0; i < j; ++i) o += (int64_t)PB_K(i) * 256;
  return o;
}
constexpr int64_t PB_TOTAL = PB_OFF(9);
static_assert(PB_TOTAL == 557056, "packed bwd size");

// ---- saved activations, SoA over P points ----
constexpr int ACT_DENSE = 64 + 8 * 256 + 256 + 32 + 128;   // floats per point
static_assert(ACT_DENSE == 2528, "act floats");
// + per tile of 64*k points: 8 layers x 4k waves x 64 ballot words (uint64) of the ReLU masks
// (= 256 bytes per point for either tile size)
constexpr int ACT_FLOATS = ACT_DENSE + 64;                   // 2592 per point (+ one tile of slack)
inline __host__ __device__ int64_t act_pe(int64_t P) { return 0; }
inline __host__ __device__ int64_t act_h(int64_t P, int l) { return P * 64 + (int64_t)l * P * 256; }
inline __host__ __device__ int64_t act_feat(int64_t P) { return P * (64 + 2048); }
inline __host__ __device__ int64_t act_vpe(int64_t P) { return P * (64 + 2048 + 256); }
inline __host__ __device__ int64_t act_hv(int64_t P) { return P * (64 + 2048 + 256 + 32); }
inline __host__ __device__ int64_t act_mask(int64_t P) { return P * ACT_DENSE; }  // uint64 words from here
// ---- pre-activation gradients, SoA ----
constexpr int DACT_FLOATS = 8 * 256 + 256 + 128;
inline __host__ __device__ int64_t dact_y(int64_t P, int l) { return (int64_t)l * P * 256; }
inline __host__ __device__ int64_t dact_feat(int64_t P) { return 8 * P * 256; }
inline __host__ __device__ int64_t dact_yv(int64_t P) { return 9 * P * 256; }
}  // namespace fnl
