// mlp_layout.h -- parameter / packed-weight / activation layouts of the 8x256 MLPs on the hot path.
//
// kind 0: NeRF            (nerf-ours/model.py:8-63,     input PE 63, parameters() order
//                          pts_linears.*, views_linears.0, feature_linear, alpha_linear, rgb_linear)
// kind 1: MLPNet fg       (nerf++-ours/nerf_network.py:70-142, input PE 63, order base_layers.*,
//                          sigma_layers.0, base_remap_layers.0, rgb_layers.0, rgb_layers.2)
// kind 2: MLPNet bg       (same, input PE 84 = 4-D inverted-sphere point)
// All three share the structure  L0..L7 (skip into L5) -> {sigma/alpha, remap/feature} -> view layer -> rgb.
#pragma once
#include <stdint.h>

namespace fnl {
struct NetLayout {
  int64_t LW[8], LB[8];                    // trunk weights / biases (flat offsets, [out][in] row-major)
  int64_t VW, VB, FW, FB, AW, AB, RW, RB;  // view layer, feature/remap, alpha/sigma, rgb
  int64_t PF[10], PB[9];                   // packed (fragment order) offsets: fwd L0..L7,F,V ; bwd Vt,Ft,L7t..L1t
  int64_t n_params, pf_total, pb_total;
  int in_pe, pe_pad;                       // 63 -> 64, 84 -> 96
  int kind;
};

inline NetLayout make_layout(int kind) {
  NetLayout L{};
  L.kind = kind;
  L.in_pe = (kind == 2) ? 84 : 63;
  L.pe_pad = (kind == 2) ? 96 : 64;
  int64_t o = 0;
  for (int i = 0; i < 8; ++i) {
    const int fan = (i == 0) ? L.in_pe : (i == 5 ? 256 + L.in_pe : 256);
    L.LW[i] = o; o += (int64_t)256 * fan;
    L.LB[i] = o; o += 256;
  }
  auto V = [&]() { L.VW = o; o += 128 * 283; L.VB = o; o += 128; };
  auto F = [&]() { L.FW = o; o += 256 * 256; L.FB = o; o += 256; };
  auto A = [&]() { L.AW = o; o += 256; L.AB = o; o += 1; };
  auto R = [&]() { L.RW = o; o += 3 * 128; L.RB = o; o += 3; };
  if (kind == 0) { V(); F(); A(); R(); } else { A(); F(); V(); R(); }
  L.n_params = o;
  int64_t p = 0;
  for (int l = 0; l < 10; ++l) {
    L.PF[l] = p;
    const int kp = (l == 0) ? L.pe_pad : (l == 5 ? L.pe_pad + 256 : (l == 9 ? 288 : 256));
    p += (int64_t)kp * (l == 9 ? 128 : 256);
  }
  L.pf_total = p;
  p = 0;
  for (int j = 0; j < 9; ++j) { L.PB[j] = p; p += (int64_t)(j == 0 ? 128 : 256) * 256; }
  L.pb_total = p;
  return L;
}

// ---- saved activations, SoA over P points: pe | h0..h7 | feat | vpe32 | hv128 | ReLU sign words ----
constexpr int ACT_REST = 8 * 256 + 256 + 32 + 128;   // 2464 floats per point after the pe block
constexpr int ACT_MASK = 64;                         // 256 bytes of sign words per point (one 64-bit word per lane, layer and wave: mlp_common.h epilogue_fwd)
inline __host__ __device__ int64_t act_pe(int64_t P, int pe_pad) { return 0; }
inline __host__ __device__ int64_t act_h(int64_t P, int pe_pad, int l) { return P * pe_pad + (int64_t)l * P * 256; }
inline __host__ __device__ int64_t act_feat(int64_t P, int pe_pad) { return P * (pe_pad + 2048); }
inline __host__ __device__ int64_t act_vpe(int64_t P, int pe_pad) { return P * (pe_pad + 2048 + 256); }
inline __host__ __device__ int64_t act_hv(int64_t P, int pe_pad) { return P * (pe_pad + 2048 + 256 + 32); }
inline __host__ __device__ int64_t act_mask(int64_t P, int pe_pad) { return P * (pe_pad + ACT_REST); }  // uint64 words
inline int64_t act_floats(int64_t P, int pe_pad) { return P * (pe_pad + ACT_REST + ACT_MASK) + 8192; }
// ---- pre-activation gradients, SoA ----
constexpr int DACT_FLOATS = 8 * 256 + 256 + 128;
inline __host__ __device__ int64_t dact_y(int64_t P, int l) { return (int64_t)l * P * 256; }
inline __host__ __device__ int64_t dact_feat(int64_t P) { return 8 * P * 256; }
inline __host__ __device__ int64_t dact_yv(int64_t P) { return 9 * P * 256; }
}  // namespace fnl
