// rays.hip -- ray generation, NDC warp, ray packing, coarse stratified sampling and the
// stand-alone positional encoder for gfx950.  All kernels are HBM-write bound and tiny next
// to the MLP; they use one lane per output row with coalesced 12/44-byte records.
//
// Reference semantics (nerf-ours/): get_rays run_nerf_helpers.py:68-78, ndc_rays :91-108,
// render() prologue render.py:59-80, coarse sampler render.py:244-266, Embedder :15-63.
// fp32 op order follows the reference (mul and add rounded separately; build uses
// -ffp-contract=off).
#include <stdarg.h>
#include "common.h"

namespace fn {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
}  // namespace fn

extern "C" int fastnerf_version(void) { return 1; }
extern "C" const char* fastnerf_last_error(void) { return fn::g_err.c_str(); }
extern "C" int fastnerf_device_cus(void) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    fn::set_error("fastnerf_device_cus: no HIP device");
    return -2;
  }
  return p.multiProcessorCount;
}

struct Cam {
  float r[12];  // c2w 3x4 row-major
};

__device__ __forceinline__ void pixel_ray(float row, float col, float fx, float fy, float cx, float cy,
                                          const float* __restrict__ c, float d[3]) {
  const float dx = (col - cx) / fx;
  const float dy = -(row - cy) / fy;
  const float dz = -1.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    d[k] = fadd(fadd(fmul(dx, c[4 * k + 0]), fmul(dy, c[4 * k + 1])), fmul(dz, c[4 * k + 2]));
}

__global__ void gen_rays_kernel(int H, int W, float fx, float fy, float cx, float cy, Cam cam,
                                float* __restrict__ ro, float* __restrict__ rd) {
  const int64_t n = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / W), col = (int)(i % W);
    float d[3];
    pixel_ray((float)row, (float)col, fx, fy, cx, cy, cam.r, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rd[i * 3 + k] = d[k];
      ro[i * 3 + k] = cam.r[4 * k + 3];
    }
  }
}

__global__ void gen_rays_pixels_kernel(int64_t n, const int32_t* __restrict__ pix, const float* __restrict__ poses,
                                       float fx, float fy, float cx, float cy, float* __restrict__ ro,
                                       float* __restrict__ rd) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int img = pix[i * 3 + 0];
    const float* c = poses + (int64_t)img * 12;
    float cc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) cc[k] = c[k];
    float d[3];
    pixel_ray((float)pix[i * 3 + 1], (float)pix[i * 3 + 2], fx, fy, cx, cy, cc, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rd[i * 3 + k] = d[k];
      ro[i * 3 + k] = cc[4 * k + 3];
    }
  }
}

__device__ __forceinline__ void ndc_one(float sx, float sy, float near, const float o[3], const float d[3],
                                        float no[3], float nd[3]) {
  const float t = -(near + o[2]) / d[2];
  float p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = fadd(o[k], fmul(t, d[k]));
  no[0] = fmul(sx, p[0]) / p[2];
  no[1] = fmul(sy, p[1]) / p[2];
  no[2] = fadd(1.0f, fmul(2.0f, near) / p[2]);
  nd[0] = fmul(sx, fsub(d[0] / d[2], p[0] / p[2]));
  nd[1] = fmul(sy, fsub(d[1] / d[2], p[1] / p[2]));
  nd[2] = fmul(-2.0f, near) / p[2];
}

__global__ void ndc_kernel(int64_t n, float sx, float sy, float near, const float* __restrict__ ro,
                           const float* __restrict__ rd, float* __restrict__ oo, float* __restrict__ od) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float o[3] = {ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2]};
    float d[3] = {rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]};
    float no[3], nd[3];
    ndc_one(sx, sy, near, o, d, no, nd);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      oo[i * 3 + k] = no[k];
      od[i * 3 + k] = nd[k];
    }
  }
}

__global__ void pack_rays_kernel(int64_t n, const float* __restrict__ ro, const float* __restrict__ rd, float near,
                                 float far, int ndc, float sx, float sy, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float o[3] = {ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2]};
    float d[3] = {rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2]};
    const float nrm = sqrtf(fadd(fadd(fmul(d[0], d[0]), fmul(d[1], d[1])), fmul(d[2], d[2])));
    float v[3] = {d[0] / nrm, d[1] / nrm, d[2] / nrm};
    if (ndc) {
      float no[3], nd[3];
      ndc_one(sx, sy, 1.0f, o, d, no, nd);
#pragma unroll
      for (int k = 0; k < 3; ++k) { o[k] = no[k]; d[k] = nd[k]; }
    }
    float* r = out + i * 11;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = near; r[7] = far;
    r[8] = v[0]; r[9] = v[1]; r[10] = v[2];
  }
}

// torch.linspace(0,1,S) fp32: symmetric evaluation (start+step*i below the midpoint,
// end-step*(S-1-i) above it).
__device__ __forceinline__ float lin01(int i, int S) {
  const float step = 1.0f / (float)(S - 1);
  return (i < S / 2) ? fmul(step, (float)i) : fsub(1.0f, fmul(step, (float)(S - 1 - i)));
}

__device__ __forceinline__ float coarse_depth(float near, float far, int i, int S, int lindisp) {
  if (S == 1) return near;
  const float t = lin01(i, S);
  if (!lindisp) return fadd(fmul(near, fsub(1.0f, t)), fmul(far, t));
  return 1.0f / fadd(fmul(1.0f / near, fsub(1.0f, t)), fmul(1.0f / far, t));
}

__global__ void sample_coarse_kernel(int64_t n, int S, const float* __restrict__ rays, int lindisp, int perturb,
                                     const float* __restrict__ t_rand, uint64_t seed, float* __restrict__ z) {
  const int64_t total = n * S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / S;
    const int s = (int)(i % S);
    const float near = rays[r * 11 + 6], far = rays[r * 11 + 7];
    const float zc = coarse_depth(near, far, s, S, lindisp);
    float out = zc;
    if (perturb) {
      const float zl = coarse_depth(near, far, s > 0 ? s - 1 : 0, S, lindisp);
      const float zu = coarse_depth(near, far, s < S - 1 ? s + 1 : S - 1, S, lindisp);
      const float lower = (s > 0) ? fmul(0.5f, fadd(zc, zl)) : zc;
      const float upper = (s < S - 1) ? fmul(0.5f, fadd(zu, zc)) : zc;
      float u;
      if (t_rand) {
        u = t_rand[i];
      } else {
        uint32_t o[4];
        philox4x32((uint32_t)i, (uint32_t)(i >> 32), 0x636f6172u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        u = u01(o[0]);
      }
      out = fadd(lower, fmul(fsub(upper, lower), u));
    }
    z[i] = out;
  }
}

__global__ void posenc_kernel(int64_t n, int L, const float* __restrict__ x, float* __restrict__ out) {
  const int C = 3 + 6 * L;
  const int64_t total = n * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / C;
    const int c = (int)(i % C);
    float v;
    if (c < 3) {
      v = x[p * 3 + c];
    } else {
      const int j = c - 3, k = j / 6, t = j % 6;
      const float a = fmul(x[p * 3 + (t % 3)], (float)(1 << k));
      v = (t < 3) ? sinf(a) : cosf(a);
    }
    out[i] = v;
  }
}

static inline int grid_for(int64_t n, int block = 256) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

extern "C" int fastnerf_gen_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                                 float* rays_o, float* rays_d, fn_stream_t stream) {
  FN_CHECK_ARG(H > 0 && W > 0 && c2w_host && rays_o && rays_d, "H,W>0 and non-null pointers");
  Cam cam;
  for (int i = 0; i < 12; ++i) cam.r[i] = c2w_host[i];
  hipLaunchKernelGGL(gen_rays_kernel, dim3(grid_for((int64_t)H * W)), dim3(256), 0, fn::S(stream), H, W, fx, fy, cx,
                     cy, cam, rays_o, rays_d);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_gen_rays_pixels(int64_t n, const int32_t* pix, const float* poses, float fx, float fy,
                                        float cx, float cy, float* rays_o, float* rays_d, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && (n == 0 || (pix && poses && rays_o && rays_d)), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gen_rays_pixels_kernel, dim3(grid_for(n)), dim3(256), 0, fn::S(stream), n, pix, poses, fx, fy,
                     cx, cy, rays_o, rays_d);
  FN_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Epoch ray generation on the device.  The reference walks every leaf of every tree in Python, draws
// its pixels with torch.randint, gathers rays / colours from [n,H,W,3] host arrays and shuffles the epoch with one
// randperm (tree.py:377-428, 569-626; nerf++-ours/tree.py:548-607 for the variance-weighted picks).  Here ONE launch
// produces the epoch's rows (rays_o, rays_d, rgb, (image, leaf) tag) in their final shuffled order:
//   output row j  <-  source index i = perm(j)   (perm: a keyed bijection of [0, N): 6-round Feistel network on the next
//                      even power of two, cycle-walked into range -- no permutation array, no [N,3] pixel list)
//   i -> its leaf l by binary search in the exclusive prefix sums of the per-leaf ray counts (the host plan)
//   i -> pixel: Philox4x32-10 keyed by the seed, counter = i.  Local indices below n_weighted[l] are drawn from the
//        leaf's clipped-variance distribution by inverse-CDF over the cumulative weights of the pixels sorted by leaf
//        (image_process.py:58-93); the rest uniformly from the leaf's integer ranges (tree.py:598-599).
//   pixel -> ray from the pose (get_rays, run_nerf_helpers.py:68-78), colour gathered from the device images.
// plan rows: image, leaf, count, row_lo, row_hi, col_lo, col_hi (int32).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint64_t feistel_perm(uint64_t j, uint64_t n, int half_bits, uint32_t k0, uint32_t k1) {
  const uint32_t mask = (half_bits >= 32) ? 0xffffffffu : ((1u << half_bits) - 1u);
  uint64_t x = j;
  do {   // cycle walking: the domain 2^(2*half_bits) is < 4n, so < 4 iterations on average
    uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      const uint32_t f = mix32(r ^ (round & 1 ? k1 : k0) ^ (0x9E3779B9u * (uint32_t)(round + 1))) & mask;
      const uint32_t nl = r;
      r = l ^ f;
      l = nl;
    }
    x = ((uint64_t)l << half_bits) | r;
  } while (x >= n);
  return x;
}

struct EpochArgs {
  int64_t N;
  int64_t n_out, batch;   // shard: n_out output rows; output row r is epoch row (r / per_batch) * batch + row0 + (r % per_batch) * stride
  int row0, stride, per_batch;
  int L, n_img, H, W, shuffle, half_bits;
  float fx, fy, cx, cy;
  uint32_t k0, k1;
};

__global__ void __launch_bounds__(256) epoch_rays_kernel(EpochArgs a, const int32_t* __restrict__ plan, const int64_t* __restrict__ offs,
                                                          const float* __restrict__ images, const float* __restrict__ poses,
                                                          const int32_t* __restrict__ n_weighted, const int64_t* __restrict__ seg_beg,
                                                          const int64_t* __restrict__ seg_end, const int32_t* __restrict__ order,
                                                          const double* __restrict__ cum, float* __restrict__ ro,
                                                          float* __restrict__ rd, float* __restrict__ rgb,
                                                          int32_t* __restrict__ tag, int32_t* __restrict__ pix) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < a.n_out; r += (int64_t)gridDim.x * blockDim.x) {
    // the epoch row this output row holds (all rows: j = r; a rank's shard: its rows row0 :: stride of every batch of the epoch)
    const int64_t jrow = a.stride == 1 && a.row0 == 0 ? r : (r / a.per_batch) * a.batch + a.row0 + (r % a.per_batch) * (int64_t)a.stride;
    const int64_t i = a.shuffle ? (int64_t)feistel_perm((uint64_t)jrow, (uint64_t)a.N, a.half_bits, a.k0, a.k1) : jrow;
    const int64_t j = r;
    // leaf of source index i: the last l with offs[l] <= i
    int lo = 0, hi = a.L;   // invariant: offs[lo] <= i < offs[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    const int32_t* pl = plan + (int64_t)lo * 7;
    const int img = pl[0], leaf = pl[1];
    uint32_t r4[4];
    philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), 0x51ED270Bu, 0u, a.k0, a.k1, r4);
    int row, col;
    const int64_t k = i - offs[lo];
    if (n_weighted && k < n_weighted[lo]) {
      // inverse CDF over this leaf's segment of the cumulative weights (fp64: 53 random bits)
      const int64_t b = seg_beg[lo], e = seg_end[lo];
      const double before = b > 0 ? cum[b - 1] : 0.0;
      const double tot = cum[e - 1] - before;
      const double u = ((double)(((uint64_t)r4[2] << 21) ^ (uint64_t)(r4[3] >> 11))) * (1.0 / 9007199254740992.0);
      const double target = before + u * tot;
      int64_t l2 = b, h2 = e - 1;   // first position with cum > target
      while (l2 < h2) {
        const int64_t m = (l2 + h2) >> 1;
        if (cum[m] > target) h2 = m; else l2 = m + 1;
      }
      const int32_t flat = order[l2];
      row = (flat / a.W) % a.H;
      col = flat % a.W;
    } else {
      row = pl[3] + (int)(((uint64_t)r4[0] * (uint64_t)(uint32_t)(pl[4] - pl[3])) >> 32);
      col = pl[5] + (int)(((uint64_t)r4[1] * (uint64_t)(uint32_t)(pl[6] - pl[5])) >> 32);
    }
    const float* c = poses + (int64_t)img * 12;
    float cc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) cc[q] = c[q];
    float d[3];
    pixel_ray((float)row, (float)col, a.fx, a.fy, a.cx, a.cy, cc, d);
    const float* px = images + (((int64_t)img * a.H + row) * a.W + col) * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      rd[j * 3 + q] = d[q];
      ro[j * 3 + q] = cc[4 * q + 3];
      rgb[j * 3 + q] = px[q];
    }
    tag[j * 2] = img;
    tag[j * 2 + 1] = leaf;
    if (pix) { pix[j * 3] = img; pix[j * 3 + 1] = row; pix[j * 3 + 2] = col; }
  }
}

// rows of an epoch of N rows that rank `row0` of `stride` ranks holds when every batch of `batch` rows is dealt out row0 :: stride
// (run_nerf.py:472-478 takes batches of N_rand consecutive rows of the shuffled epoch)
extern "C" int64_t fastnerf_epoch_shard_rows(int64_t N, int64_t batch, int row0, int stride) {
  if (N < 0 || batch < 1 || stride < 1 || row0 < 0 || row0 >= stride) return -1;
  const int64_t per = batch > row0 ? (batch - row0 + stride - 1) / stride : 0;
  const int64_t full = N / batch, tail = N - full * batch;
  return full * per + (tail > row0 ? (tail - row0 + stride - 1) / stride : 0);
}

extern "C" int fastnerf_epoch_rays_shard(int64_t N, int L, const int32_t* plan, const int64_t* offs, const float* images,
                                         const float* poses, int n_img, int H, int W, float fx, float fy, float cx, float cy,
                                         uint64_t seed, int shuffle, const int32_t* n_weighted, const int64_t* seg_beg,
                                         const int64_t* seg_end, const int32_t* order, const double* cum, int64_t batch, int row0,
                                         int stride, float* rays_o, float* rays_d, float* rgb, int32_t* tag, int32_t* pix,
                                         fn_stream_t stream) {
  FN_CHECK_ARG(N >= 0 && L >= 1 && n_img >= 1 && H >= 1 && W >= 1, "N>=0, L>=1, n_img>=1, H,W>=1");
  FN_CHECK_ARG(batch >= 1 && stride >= 1 && row0 >= 0 && row0 < stride, "batch>=1, 0 <= row0 < stride");
  if (N == 0 || fastnerf_epoch_shard_rows(N, batch, row0, stride) == 0) return 0;   // (a rank without rows: nothing to write, no buffers needed)
  FN_CHECK_ARG(plan && offs && images && poses && rays_o && rays_d && rgb && tag, "null pointer");
  FN_CHECK_ARG(!n_weighted || (seg_beg && seg_end && order && cum), "weighted picks need seg_beg, seg_end, order, cum");
  FN_CHECK_ARG((int64_t)n_img * H * W < ((int64_t)1 << 31), "pixel ids are int32");
  EpochArgs a;
  a.N = N; a.L = L; a.n_img = n_img; a.H = H; a.W = W; a.shuffle = shuffle;
  a.n_out = fastnerf_epoch_shard_rows(N, batch, row0, stride);
  a.batch = batch; a.row0 = row0; a.stride = stride;
  a.per_batch = (int)(batch > row0 ? (batch - row0 + stride - 1) / stride : 1);
  a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
  int bits = 2;
  while (bits < 62 && ((uint64_t)1 << bits) < (uint64_t)N) ++bits;
  a.half_bits = (bits + 1) / 2;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32);
  hipLaunchKernelGGL(epoch_rays_kernel, dim3(grid_for(a.n_out)), dim3(256), 0, fn::S(stream), a, plan, offs, images, poses, n_weighted,
                     seg_beg, seg_end, order, cum, rays_o, rays_d, rgb, tag, pix);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_epoch_rays(int64_t N, int L, const int32_t* plan, const int64_t* offs, const float* images,
                                   const float* poses, int n_img, int H, int W, float fx, float fy, float cx, float cy,
                                   uint64_t seed, int shuffle, const int32_t* n_weighted, const int64_t* seg_beg,
                                   const int64_t* seg_end, const int32_t* order, const double* cum, float* rays_o,
                                   float* rays_d, float* rgb, int32_t* tag, int32_t* pix, fn_stream_t stream) {
  return fastnerf_epoch_rays_shard(N, L, plan, offs, images, poses, n_img, H, W, fx, fy, cx, cy, seed, shuffle, n_weighted, seg_beg, seg_end,
                                   order, cum, N > 0 ? N : 1, 0, 1, rays_o, rays_d, rgb, tag, pix, stream);
}

// -1./(W/(2.*focal)) is evaluated in double by Python and then applied as an fp32 scalar
static inline float ndc_scale(int dim, double focal) { return (float)(-1.0 / ((double)dim / (2.0 * focal))); }

extern "C" int fastnerf_ndc_rays(int64_t n, int H, int W, double focal, float near, const float* rays_o,
                                 const float* rays_d, float* out_o, float* out_d, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && (n == 0 || (rays_o && rays_d && out_o && out_d)), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ndc_kernel, dim3(grid_for(n)), dim3(256), 0, fn::S(stream), n, ndc_scale(W, focal), ndc_scale(H, focal),
                     near, rays_o, rays_d, out_o, out_d);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_pack_rays(int64_t n, const float* rays_o, const float* rays_d, float near, float far, int ndc,
                                  int H, int W, double focal, float* rays11, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && (n == 0 || (rays_o && rays_d && rays11)), "null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pack_rays_kernel, dim3(grid_for(n)), dim3(256), 0, fn::S(stream), n, rays_o, rays_d, near, far,
                     ndc, ndc ? ndc_scale(W, focal) : 0.f, ndc ? ndc_scale(H, focal) : 0.f, rays11);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_sample_coarse(int64_t n, int S, const float* rays11, int lindisp, int perturb,
                                      const float* t_rand, uint64_t seed, float* z, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 1 && (n == 0 || (rays11 && z)), "n>=0, S>=1, non-null pointers");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sample_coarse_kernel, dim3(grid_for(n * S)), dim3(256), 0, fn::S(stream), n, S, rays11, lindisp,
                     perturb, t_rand, seed, z);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_posenc(int64_t n, int L, const float* x, float* out, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && L >= 0 && L <= 16 && (n == 0 || (x && out)), "n>=0, 0<=L<=16, non-null pointers");
  if (n == 0) return 0;
  hipLaunchKernelGGL(posenc_kernel, dim3(grid_for(n * (3 + 6 * L))), dim3(256), 0, fn::S(stream), n, L, x, out);
  FN_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------------------
// nerf++ (ddp_train_nerf.py:54-69, 355-361)
// ---------------------------------------------------------------------------------------
__global__ void intersect_sphere_kernel(int64_t n, const float* __restrict__ rays, float* __restrict__ fg_far,
                                        int* __restrict__ n_outside) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* r = rays + i * 11;
    const float o[3] = {r[0], r[1], r[2]}, d[3] = {r[3], r[4], r[5]};
    const float dd = fadd(fadd(fmul(d[0], d[0]), fmul(d[1], d[1])), fmul(d[2], d[2]));
    const float od = fadd(fadd(fmul(d[0], o[0]), fmul(d[1], o[1])), fmul(d[2], o[2]));
    const float d1 = -od / dd;
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = fadd(o[c], fmul(d1, d[c]));
    const float pn2 = fadd(fadd(fmul(p[0], p[0]), fmul(p[1], p[1])), fmul(p[2], p[2]));
    if (pn2 >= 1.0f && n_outside) atomicAdd(n_outside, 1);
    fg_far[i] = fadd(d1, fmul(sqrtf(fsub(1.0f, pn2)), 1.0f / sqrtf(dd)));
  }
}

__global__ void fg_depths_kernel(int64_t n, int S, float near, const float* __restrict__ fg_far,
                                 const float* __restrict__ t_rand, int perturb, uint64_t seed, float* __restrict__ z) {
  const int64_t total = n * S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / S;
    const int s = (int)(i % S);
    const float step = fsub(fg_far[r], near) / (float)(S - 1);
    auto zat = [&](int k) { return fadd(near, fmul((float)k, step)); };   // near + i * step
    const float zc = zat(s);
    float out = zc;
    if (perturb) {
      const float lower = (s > 0) ? fmul(0.5f, fadd(zc, zat(s - 1))) : zc;
      const float upper = (s < S - 1) ? fmul(0.5f, fadd(zat(s + 1), zc)) : zc;
      float u;
      if (t_rand) {
        u = t_rand[i];
      } else {
        uint32_t o[4];
        philox4x32((uint32_t)i, (uint32_t)(i >> 32), 0x66676470u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        u = u01(o[0]);
      }
      out = fadd(lower, fmul(fsub(upper, lower), u));
    }
    z[i] = out;
  }
}

extern "C" int fastnerf_pp_intersect_sphere(int64_t n, const float* rays11, float* fg_far, int* n_outside,
                                            fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && (n == 0 || (rays11 && fg_far)), "null pointer");
  if (n == 0) return 0;
  if (n_outside) FN_HIP(hipMemsetAsync(n_outside, 0, sizeof(int), fn::S(stream)));
  hipLaunchKernelGGL(intersect_sphere_kernel, dim3(grid_for(n)), dim3(256), 0, fn::S(stream), n, rays11, fg_far,
                     n_outside);
  FN_LAUNCH_CHECK();
  return 0;
}

extern "C" int fastnerf_pp_fg_depths(int64_t n, int S, float near, const float* fg_far, int perturb,
                                     const float* t_rand, uint64_t seed, float* z, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 2 && (n == 0 || (fg_far && z)), "n>=0, S>=2, non-null pointers");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fg_depths_kernel, dim3(grid_for(n * S)), dim3(256), 0, fn::S(stream), n, S, near, fg_far, t_rand,
                     perturb, seed, z);
  FN_LAUNCH_CHECK();
  return 0;
}


// perturb_samples (ddp_train_nerf.py:72-81): stratified jitter of arbitrary sorted depths, z -> lower + (upper - lower) * u with
// the mid points as interval borders; u injected ([n,S]) or Philox(seed).
__global__ void pp_perturb_kernel(int64_t n, int S, const float* __restrict__ zin, const float* __restrict__ t_rand,
                                  uint64_t seed, float* __restrict__ z) {
  const int64_t total = n * S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % S);
    const float zc = zin[i];
    const float lower = (s > 0) ? fmul(0.5f, fadd(zc, zin[i - 1])) : zc;
    const float upper = (s < S - 1) ? fmul(0.5f, fadd(zin[i + 1], zc)) : zc;
    float u;
    if (t_rand) {
      u = t_rand[i];
    } else {
      uint32_t o[4];
      philox4x32((uint32_t)i, (uint32_t)(i >> 32), 0x70727462u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
      u = u01(o[0]);
    }
    z[i] = fadd(lower, fmul(fsub(upper, lower), u));
  }
}
extern "C" int fastnerf_pp_perturb_samples(int64_t n, int S, const float* z_in, const float* t_rand, uint64_t seed, float* z_out,
                                           fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 1 && (n == 0 || (z_in && z_out)) && z_in != z_out, "n>=0, S>=1, non-null distinct pointers");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pp_perturb_kernel, dim3(grid_for(n * S)), dim3(256), 0, fn::S(stream), n, S, z_in, t_rand, seed, z_out);
  FN_LAUNCH_CHECK();
  return 0;
}

// depth2pts_outside (ddp_model.py:16-45): the 4-D inverted-sphere point (x', y', z', 1/r) of every background sample and its
// conventional depth; the arithmetic (and its order) is the one the fused background MLP kernels evaluate per point.
__global__ void pp_depth2pts_kernel(int64_t n, int S, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                    const float* __restrict__ depth, float* __restrict__ pts, float* __restrict__ depth_real) {
  const int64_t total = n * S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / S;
    const float o[3] = {ray_o[r * 3], ray_o[r * 3 + 1], ray_o[r * 3 + 2]}, d[3] = {ray_d[r * 3], ray_d[r * 3 + 1], ray_d[r * 3 + 2]};
    const float dep = depth[i];
    const float dd = fadd(fadd(fmul(d[0], d[0]), fmul(d[1], d[1])), fmul(d[2], d[2]));
    const float od = fadd(fadd(fmul(d[0], o[0]), fmul(d[1], o[1])), fmul(d[2], o[2]));
    const float d1 = -od / dd;
    float pm_[3], ps[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pm_[c] = fadd(o[c], fmul(d1, d[c]));
    const float pmn = sqrtf(fadd(fadd(fmul(pm_[0], pm_[0]), fmul(pm_[1], pm_[1])), fmul(pm_[2], pm_[2])));
    const float dcos = 1.0f / sqrtf(dd);
    const float d2 = fmul(sqrtf(fsub(1.0f, fmul(pmn, pmn))), dcos);
    const float d12 = fadd(d1, d2);
#pragma unroll
    for (int c = 0; c < 3; ++c) ps[c] = fadd(o[c], fmul(d12, d[c]));
    float ax[3] = {fsub(fmul(o[1], ps[2]), fmul(o[2], ps[1])), fsub(fmul(o[2], ps[0]), fmul(o[0], ps[2])),
                   fsub(fmul(o[0], ps[1]), fmul(o[1], ps[0]))};
    const float an = sqrtf(fadd(fadd(fmul(ax[0], ax[0]), fmul(ax[1], ax[1])), fmul(ax[2], ax[2])));
#pragma unroll
    for (int c = 0; c < 3; ++c) ax[c] = ax[c] / an;
    const float theta = asinf(fmul(pmn, dep));
    const float ang = fsub(asinf(pmn), theta);
    const float ca = cosf(ang), sa = sinf(ang);
    const float cr[3] = {fsub(fmul(ax[1], ps[2]), fmul(ax[2], ps[1])), fsub(fmul(ax[2], ps[0]), fmul(ax[0], ps[2])),
                         fsub(fmul(ax[0], ps[1]), fmul(ax[1], ps[0]))};
    const float dot = fadd(fadd(fmul(ax[0], ps[0]), fmul(ax[1], ps[1])), fmul(ax[2], ps[2]));
    const float omc = fsub(1.0f, ca);
    float pn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pn[c] = fadd(fadd(fmul(ps[c], ca), fmul(cr[c], sa)), fmul(fmul(ax[c], dot), omc));
    const float nn = sqrtf(fadd(fadd(fmul(pn[0], pn[0]), fmul(pn[1], pn[1])), fmul(pn[2], pn[2])));
    pts[i * 4] = pn[0] / nn; pts[i * 4 + 1] = pn[1] / nn; pts[i * 4 + 2] = pn[2] / nn; pts[i * 4 + 3] = dep;
    if (depth_real) depth_real[i] = fadd(fmul(fmul(1.0f / fadd(dep, 1e-6f), cosf(theta)), dcos), d1);
  }
}
extern "C" int fastnerf_pp_depth2pts_outside(int64_t n, int S, const float* ray_o, const float* ray_d, const float* depth, float* pts,
                                             float* depth_real, fn_stream_t stream) {
  FN_CHECK_ARG(n >= 0 && S >= 1 && (n == 0 || (ray_o && ray_d && depth && pts)), "n>=0, S>=1, non-null pointers");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pp_depth2pts_kernel, dim3(grid_for(n * S)), dim3(256), 0, fn::S(stream), n, S, ray_o, ray_d, depth, pts,
                     depth_real);
  FN_LAUNCH_CHECK();
  return 0;
}


// nerf++ ray generator (nerf_sample_ray_split.py:10-34): OpenCV convention, pixel centres at +0.5,
// d = R * K^-1 * [u, v, 1], evaluated in fp64 like the reference's numpy code and rounded once.
struct Mat9d { double m[9]; };
__global__ void pp_gen_rays_kernel(int H, int W, Mat9d RK, float ox, float oy, float oz, float* __restrict__ ro,
                                   float* __restrict__ rd) {
  const int64_t n = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double u = (double)((float)(i % W) + 0.5f), v = (double)((float)(i / W) + 0.5f);
#pragma unroll
    for (int k = 0; k < 3; ++k) rd[i * 3 + k] = (float)(RK.m[3 * k] * u + RK.m[3 * k + 1] * v + RK.m[3 * k + 2]);
    ro[i * 3] = ox; ro[i * 3 + 1] = oy; ro[i * 3 + 2] = oz;
  }
}

extern "C" int fastnerf_pp_gen_rays(int H, int W, const double* intrinsics_host, const double* c2w_host, float* rays_o,
                                    float* rays_d, fn_stream_t stream) {
  FN_CHECK_ARG(H > 0 && W > 0 && intrinsics_host && c2w_host && rays_o && rays_d, "H,W>0 and non-null pointers");
  // K^-1 of the upper-left 3x3 of the 4x4 intrinsics, then R * K^-1 (both row-major 4x4 inputs)
  const double* K = intrinsics_host;
  const double a = K[0], b = K[1], c = K[2], d = K[4], e = K[5], f = K[6], g = K[8], h = K[9], i = K[10];
  const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  FN_CHECK_ARG(det != 0.0, "singular intrinsics");
  const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                         (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                         (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
  Mat9d RK;
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += c2w_host[4 * r + k] * inv[3 * k + cc];
      RK.m[3 * r + cc] = s;
    }
  hipLaunchKernelGGL(pp_gen_rays_kernel, dim3(grid_for((int64_t)H * W)), dim3(256), 0, fn::S(stream), H, W, RK,
                     (float)c2w_host[3], (float)c2w_host[7], (float)c2w_host[11], rays_o, rays_d);
  FN_LAUNCH_CHECK();
  return 0;
}
