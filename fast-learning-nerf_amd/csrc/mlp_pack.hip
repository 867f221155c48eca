// mlp_pack.hip -- weight packing for the fp32-width MLP kernels (fragment order of the MFMA that consumes them) and the net-size queries.
#include "mlp_common.h"

// =========================================================================================
// weight packing
// =========================================================================================
struct PackDesc {
  int64_t src_off;   // flat offset of the [out][in] weight
  int64_t dst_off;   // packed offset
  int ld;            // source row length (fan-in)
  int n_rows;        // fwd: N (out features); bwd: K (= out features)
  int n_cols;        // fwd: Kp (padded fan-in);  bwd: 256 (in features written)
  int segA_pad, segA_valid, segB_valid;  // fwd: k' -> source column mapping
  int col0;          // bwd: first source column
  int transposed;
};
struct PackTable {
  PackDesc d[19];
};

// fwd  : dst[((nt*KS+ks)*64 + l)*4 + t] = W'[nt*32 + (l&31)][ks*8 + (l>>5)*4 + t]
// bwd  : dst[((jt*KS+ks)*64 + l)*4 + t] = W [ks*8 + (l>>5)*4 + t][col0 + jt*32 + (l&31)]
__global__ void __launch_bounds__(256) pack_kernel(PackTable tab, const float* __restrict__ params,
                                                    float* __restrict__ pf, float* __restrict__ pb) {
  const PackDesc d = tab.d[blockIdx.y];
  const int64_t total = (int64_t)d.n_rows * d.n_cols;
  float* dst = (d.transposed ? pb : pf) + d.dst_off;
  const float* src = params + d.src_off;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e & 3);
    const int l = (int)((e >> 2) & 63);
    const int64_t blk = e >> 8;  // tile*KS + ks
    float v = 0.f;
    if (!d.transposed) {
      const int KS = d.n_cols / 8;
      const int nt = (int)(blk / KS), ks = (int)(blk % KS);
      const int n = nt * 32 + (l & 31);
      const int kp = ks * 8 + (l >> 5) * 4 + t;
      int col = -1;
      if (kp < d.segA_pad) {
        if (kp < d.segA_valid) col = kp;
      } else {
        const int q = kp - d.segA_pad;
        if (q < d.segB_valid) col = d.segA_valid + q;
      }
      if (col >= 0) v = src[(int64_t)n * d.ld + col];
    } else {
      const int KS = d.n_rows / 8;
      const int jt = (int)(blk / KS), ks = (int)(blk % KS);
      const int o = ks * 8 + (l >> 5) * 4 + t;
      const int c = d.col0 + jt * 32 + (l & 31);
      v = src[(int64_t)o * d.ld + c];
    }
    dst[e] = v;
  }
}

__global__ void __launch_bounds__(256) pack6_kernel(PackTable tab, const float* __restrict__ params,
                                                     uint4* __restrict__ pf, uint4* __restrict__ pb) {
  const PackDesc d = tab.d[blockIdx.y];
  constexpr int TW = 16, KW = 32;
  const int KS = (d.transposed ? d.n_rows : d.n_cols) / KW;
  const int NTL = (d.transposed ? d.n_cols : d.n_rows) / TW;
  const int64_t total = (int64_t)NTL * KS * 64;          // one thread = the three planes of one (tile, k-step, lane)
  uint4* dst = (d.transposed ? pb : pf) + d.dst_off * 3 / 8;   // dst_off: floats of the fp32 packing = 8/3 of these uint4
  const float* src = params + d.src_off;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(e & 63);
    const int64_t blk = e >> 6;
    const int tile = (int)(blk / KS), ks = (int)(blk % KS);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kp = ks * KW + (l / TW) * 8 + j;
      v[j] = 0.f;
      if (!d.transposed) {
        const int n = tile * TW + (l % TW);
        int col = -1;
        if (kp < d.segA_pad) { if (kp < d.segA_valid) col = kp; }
        else { const int q = kp - d.segA_pad; if (q < d.segB_valid) col = d.segA_valid + q; }
        if (col >= 0) v[j] = src[(int64_t)n * d.ld + col];
      } else {
        v[j] = src[(int64_t)kp * d.ld + d.col0 + tile * TW + (l % TW)];
      }
    }
    unsigned h[4], m[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      split3_pair(v[2 * q], v[2 * q + 1], h[q], m[q], lo[q]);
    }
    uint4* o = dst + (blk * 3) * 64 + l;
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[64] = make_uint4(m[0], m[1], m[2], m[3]);
    o[128] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

static PackTable make_pack_table(const NetLayout& L) {
  PackTable T;
  int n = 0;
  for (int l = 0; l < 8; ++l) {
    PackDesc d{};
    d.src_off = L.LW[l]; d.dst_off = L.PF[l]; d.n_rows = 256;
    d.ld = (l == 0) ? L.in_pe : (l == 5 ? 256 + L.in_pe : 256);
    d.n_cols = (l == 0) ? L.pe_pad : (l == 5 ? L.pe_pad + 256 : 256);
    if (l == 0) { d.segA_pad = L.pe_pad; d.segA_valid = L.in_pe; d.segB_valid = 0; }
    else if (l == 5) { d.segA_pad = L.pe_pad; d.segA_valid = L.in_pe; d.segB_valid = 256; }
    else { d.segA_pad = 256; d.segA_valid = 256; d.segB_valid = 0; }
    T.d[n++] = d;
  }
  { PackDesc d{}; d.src_off = L.FW; d.dst_off = L.PF[8]; d.ld = 256; d.n_rows = 256; d.n_cols = 256;
    d.segA_pad = 256; d.segA_valid = 256; T.d[n++] = d; }
  { PackDesc d{}; d.src_off = L.VW; d.dst_off = L.PF[9]; d.ld = 283; d.n_rows = 128; d.n_cols = 288;
    d.segA_pad = 256; d.segA_valid = 256; d.segB_valid = 27; T.d[n++] = d; }
  // transposed: views(feat part), feature, trunk 7,6,5(h part),4,3,2,1
  { PackDesc d{}; d.transposed = 1; d.src_off = L.VW; d.dst_off = L.PB[0]; d.ld = 283; d.n_rows = 128; d.n_cols = 256; d.col0 = 0; T.d[n++] = d; }
  { PackDesc d{}; d.transposed = 1; d.src_off = L.FW; d.dst_off = L.PB[1]; d.ld = 256; d.n_rows = 256; d.n_cols = 256; d.col0 = 0; T.d[n++] = d; }
  const int order[7] = {7, 6, 5, 4, 3, 2, 1};
  for (int j = 0; j < 7; ++j) {
    const int l = order[j];
    PackDesc d{}; d.transposed = 1; d.src_off = L.LW[l]; d.dst_off = L.PB[2 + j]; d.n_rows = 256; d.n_cols = 256;
    d.ld = (l == 5) ? 256 + L.in_pe : 256;
    d.col0 = (l == 5) ? L.in_pe : 0;
    T.d[n++] = d;
  }
  return T;
}


extern "C" int64_t fastnerf_net_floats(int kind, int what) {
  if (kind < 0 || kind > 2) return -1;
  const NetLayout& L = layout_of(kind);
  return what == 0 ? L.n_params : what == 1 ? L.pf_total : what == 2 ? L.pb_total : what == 3 ? L.pe_pad : -1;
}
extern "C" int64_t fastnerf_mlp_act_floats(int kind, int64_t P) {
  if (kind < 0 || kind > 2 || P < 0) return -1;
  return act_floats(P, layout_of(kind).pe_pad);
}

extern "C" int fastnerf_mlp_pack_ex(int kind, const float* params, float* packed_fwd, float* packed_bwd,
                                    fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && params && packed_fwd && packed_bwd, "kind in 0..2, non-null pointers");
  static const PackTable T[3] = {make_pack_table(layout_of(0)), make_pack_table(layout_of(1)),
                                 make_pack_table(layout_of(2))};
  hipLaunchKernelGGL(pack_kernel, dim3(64, 19), dim3(256), 0, fn::S(stream), T[kind], params, packed_fwd, packed_bwd);
  FN_LAUNCH_CHECK();
  return 0;
}
// MM_X6 packing: 1.5x the floats of the fp32 packing (three bf16 planes)
extern "C" int64_t fastnerf_mlp_x6_packed_floats(int kind, int which) {
  if (kind < 0 || kind > 2 || (which != 1 && which != 2)) return -1;
  const NetLayout& L = layout_of(kind);
  return (which == 1 ? L.pf_total : L.pb_total) * 3 / 2;
}
extern "C" int fastnerf_mlp_x6_pack(int kind, const float* params, float* packed_fwd, float* packed_bwd, fn_stream_t stream) {
  FN_CHECK_ARG(kind >= 0 && kind <= 2 && params && packed_fwd && packed_bwd, "kind in 0..2, non-null pointers");
  static const PackTable T[3] = {make_pack_table(layout_of(0)), make_pack_table(layout_of(1)),
                                 make_pack_table(layout_of(2))};
  hipLaunchKernelGGL(pack6_kernel, dim3(32, 19), dim3(256), 0, fn::S(stream), T[kind], params,
                     reinterpret_cast<uint4*>(packed_fwd), reinterpret_cast<uint4*>(packed_bwd));
  FN_LAUNCH_CHECK();
  return 0;
}
extern "C" int fastnerf_mlp_pack(const float* params, float* packed_fwd, float* packed_bwd, fn_stream_t stream) {
  return fastnerf_mlp_pack_ex(0, params, packed_fwd, packed_bwd, stream);
}

