/* fastnerf.h -- C ABI of the MI355X-native NeRF training inner loop.
 *
 * Drop-in boundary for the renderer / quadtree path of
 * wen-yuan-zhang/Fast-Learning-NeRF (nerf-ours).  The reference has no FFI: the
 * path sits behind Python functions.  Every entry point below names the
 * reference function (file:line, relative to nerf-ours/) whose work it
 * replaces; INTEGRATION.md shows the ctypes stub a maintainer adds to call it.
 *
 * Conventions
 *   - all `float*`/`int*` arguments are DEVICE pointers unless the name ends in
 *     `_host`; buffers are owned by the caller (PyTorch's allocator); nothing
 *     is retained past return.
 *   - fp32, contiguous row-major.  rays are [N,11] = o(3) d(3) near far
 *     viewdir(3) (render.py:74-80).
 *   - every call enqueues work on `stream` (a hipStream_t) and returns without
 *     synchronising.
 *   - return 0 on success, <0 on error (-1 bad argument, -2 HIP error);
 *     fastnerf_last_error() returns a thread-local message.  Never aborts.
 */
#ifndef FASTNERF_H
#define FASTNERF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* fn_stream_t; /* hipStream_t */

int fastnerf_version(void);
const char* fastnerf_last_error(void);
/* device properties used by bench/tests: returns CU count (or <0) */
int fastnerf_device_cus(void);

/* ---- rays ------------------------------------------------------------- */
/* get_rays (run_nerf_helpers.py:68-78): all H*W pixels of one camera.
 * c2w_host: 12 floats (3x4 row-major).  rays_o/rays_d: [H,W,3]. */
int fastnerf_gen_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                      float* rays_o, float* rays_d, fn_stream_t stream);
/* rays for selected pixels of many cameras (tree.py:617-619 gather restated as
 * on-the-fly generation): pix [n,3] int32 (image, row, col); poses [n_img,3,4]. */
int fastnerf_gen_rays_pixels(int64_t n, const int32_t* pix, const float* poses, float fx, float fy, float cx,
                             float cy, float* rays_o, float* rays_d, fn_stream_t stream);
/* ndc_rays (run_nerf_helpers.py:91-108), in -> out [n,3] each. */
int fastnerf_ndc_rays(int64_t n, int H, int W, double focal, float near, const float* rays_o, const float* rays_d,
                      float* out_o, float* out_d, fn_stream_t stream);
/* render() prologue (render.py:59-80): viewdirs = d/|d| (before ndc), optional
 * ndc warp, pack [n,11]. */
int fastnerf_pack_rays(int64_t n, const float* rays_o, const float* rays_d, float near, float far, int ndc, int H,
                       int W, double focal, float* rays11, fn_stream_t stream);
/* coarse depths (render.py:244-266).  t_rand: [n,S] injected U[0,1) jitter, or
 * NULL; when NULL and perturb!=0 a Philox stream keyed by (seed, ray, sample)
 * is used.  z: [n,S]. */
int fastnerf_sample_coarse(int64_t n, int S, const float* rays11, int lindisp, int perturb, const float* t_rand,
                           uint64_t seed, float* z, fn_stream_t stream);
/* Embedder.embed (run_nerf_helpers.py:15-63): x [n,3] -> out [n, 3+6L]. */
int fastnerf_posenc(int64_t n, int L, const float* x, float* out, fn_stream_t stream);

/* ---- MLP (model.py:8-63, D=8 W=256 skip@4 use_viewdirs) ---------------- */
#define FASTNERF_NET_PARAMS 595844      /* floats per net, model.parameters() order */
#define FASTNERF_PACKED_FWD 593920      /* floats: fragment-ordered forward weights  */
#define FASTNERF_PACKED_BWD 557056      /* floats: fragment-ordered transposed weights */
/* saved activations: n*S*FASTNERF_ACT_FLOATS + FASTNERF_ACT_SLACK floats
 * (per point pe64 + 8*h256 + feat256 + vpe32 + hv128 + 64 floats of ReLU sign words) */
#define FASTNERF_ACT_FLOATS 2592
#define FASTNERF_ACT_SLACK 8192   /* kind 0; in general use fastnerf_mlp_act_floats() */
/* per-point pre-activation gradients (floats): 8*dY256 + dfeat256 + dYv128 */
#define FASTNERF_DACT_FLOATS 2432

/* re-layout of one net's flat parameters into the MFMA fragment order used by
 * mlp_fwd (packed_fwd) and mlp_bwd_dx (packed_bwd). */
int fastnerf_mlp_pack(const float* params, float* packed_fwd, float* packed_bwd, fn_stream_t stream);
/* run_network (run_nerf.py:50-64) + NeRF.forward: points are generated on the
 * fly from rays11 [n,11] and z [n,S] (pts = o + d*z, render.py:268), encoded
 * (L=10 / L=4) and pushed through the MLP.  raw: [n,S,4] (rgb logits, sigma).
 * act: NULL (inference) or a buffer of n*S*FASTNERF_ACT_FLOATS + FASTNERF_ACT_SLACK floats that
 * receives the activations backward needs. */
int fastnerf_mlp_fwd(int64_t n, int S, const float* rays11, const float* z, const float* params,
                     const float* packed_fwd, float* raw, float* act, fn_stream_t stream);
/* backward of the above w.r.t. the parameters: draw [n,S,4] -> grads (same
 * layout as params, OVERWRITTEN).  dact: scratch n*S*FASTNERF_DACT_FLOATS;
 * partial: scratch of fastnerf_mlp_bwd_partial_floats() floats. */
int64_t fastnerf_mlp_bwd_partial_floats(void);
int fastnerf_mlp_bwd(int64_t n, int S, const float* draw, const float* act, const float* params,
                     const float* packed_bwd, float* dact, float* partial, float* grads, fn_stream_t stream);

/* ---- compositing / hierarchical sampling ------------------------------ */
/* raw2outputs (render.py:149-192).  noise: [n,S] scaled sigma noise or NULL.
 * outputs: rgb_map [n,3], disp [n], acc [n], weights [n,S], depth [n]. */
int fastnerf_raw2outputs_fwd(int64_t n, int S, const float* raw, const float* z, const float* rays11,
                             const float* noise, int white_bkgd, float* rgb_map, float* disp, float* acc,
                             float* weights, float* depth, fn_stream_t stream);
/* d(rgb_map)/d(raw): g_rgb [n,3] -> draw [n,S,4]. */
int fastnerf_raw2outputs_bwd(int64_t n, int S, const float* raw, const float* z, const float* rays11,
                             const float* noise, int white_bkgd, const float* g_rgb, float* draw,
                             fn_stream_t stream);
/* sample_pdf (run_nerf_helpers.py:112-155) on bins = mid(z), weights[1:-1],
 * followed by sort(cat[z, z_samples]) (render.py:279-283).  u: [n,Ni] injected
 * uniforms or NULL; det!=0 -> linspace(0,1,Ni); else Philox(seed).
 * z_out: [n,S+Ni] sorted; z_samples: [n,Ni] (may be NULL); z_std [n] (may be NULL). */
int fastnerf_sample_pdf_merge(int64_t n, int S, int Ni, const float* z, const float* weights, int det,
                              const float* u, uint64_t seed, float* z_out, float* z_samples, float* z_std,
                              fn_stream_t stream);

/* stand-alone sample_pdf(bins [n,M], weights [n,M-1]) -> samples [n,Ni]
 * (run_nerf_helpers.py:112-155 as called by third parties, e.g. tests). */
int fastnerf_sample_pdf(int64_t n, int M, int Ni, const float* bins, const float* weights, int det, const float* u,
                        uint64_t seed, float* samples, fn_stream_t stream);

/* ---- loss / optimiser / quadtree loss map ------------------------------ */
/* img2mse (run_nerf_helpers.py:9) for fine and coarse maps + their gradients
 * + the per-(image, leaf) max |gt - pred| table (tree.py:538, 632-642).
 * loss2: 2 floats (fine mse, coarse mse), accumulated from zero by the call.
 * leaf_tag: [n,2] int32 (image, leaf) or NULL; table: [n_img*max_leaves] uint32
 * bit patterns of non-negative floats (atomicMax), or NULL.
 * grad_scale multiplies 2/(3n) (data-parallel: n_local/n_global). */
int fastnerf_mse_leafmax(int64_t n, const float* rgb, const float* rgb0, const float* target, float grad_scale,
                         float* g_rgb, float* g_rgb0, float* loss2, const int32_t* leaf_tag, int max_leaves,
                         uint32_t* table, fn_stream_t stream);
/* torch.optim.Adam step (run_nerf.py:99,494) over a flat buffer. */
int fastnerf_adam_step(int64_t n, float* params, const float* grads, float* m, float* v, double lr, double beta1,
                       double beta2, double eps, int step, fn_stream_t stream);

/* ---- generalised MLP entry points: kind 0 = NeRF (model.py), 1 = nerf++ MLPNet foreground,
 * 2 = nerf++ MLPNet background (4-D inverted-sphere input, samples consumed far->near;
 * nerf++-ours/nerf_network.py:70-142, ddp_model.py:110-124) ------------------------------------- */
/* what: 0 parameter floats, 1 packed-forward floats, 2 packed-backward floats, 3 padded PE width */
int64_t fastnerf_net_floats(int kind, int what);
int64_t fastnerf_mlp_act_floats(int kind, int64_t n_points);
int fastnerf_mlp_pack_ex(int kind, const float* params, float* packed_fwd, float* packed_bwd, fn_stream_t stream);
int fastnerf_mlp_fwd_ex(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                        const float* packed_fwd, float* raw, float* act, fn_stream_t stream);
int fastnerf_mlp_bwd_ex(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                        const float* packed_bwd, float* dact, float* partial, float* grads, fn_stream_t stream);

/* ---- split-bf16 ("bf16x3") math mode, all three net kinds ------------------------------------------
 * Same network functions and call protocol as fastnerf_mlp_pack_ex / fwd_ex / bwd_ex (run_nerf.py:91-107
 * run_network -> model.py:37-63, autograd backward of the same), computed on the bf16 matrix cores: every fp32
 * operand is carried as a (hi, lo) bf16 pair and products are hi*hi + hi*lo + lo*hi with fp32 accumulation
 * (csrc/mlp_bf16.hip).  Buffers are opaque and sized by fastnerf_mlp_bf16_floats (in 4-byte units):
 * what 1 packed forward weights, 2 packed backward weights, 3 saved activations for n_points, 4 pre-activation
 * gradients for n_points.  act == NULL in fwd: inference, nothing saved. */
int64_t fastnerf_mlp_bf16_floats(int kind, int what, int64_t n_points);
int64_t fastnerf_mlp_bf16_partial_floats(void);
int fastnerf_mlp_bf16_pack(int kind, const float* params, float* packed_fwd, float* packed_bwd, fn_stream_t stream);
int fastnerf_mlp_bf16_fwd(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                          const float* packed_fwd, float* raw, float* act, fn_stream_t stream);
int fastnerf_mlp_bf16_bwd(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                          const float* packed_bwd, float* dact, float* partial, float* grads, fn_stream_t stream);

/* Inference forward (no saved activations) with options.  flags & FN_FWD_SKIP_DEAD_RGB (kind 0 only): a
 * 64-point tile whose samples ALL have sigma <= 0 skips the feature layer, the view layer and the colour head and reports
 * colour logits 0 -- valid when nothing reads the colour of a sample whose weight is exactly zero: compositing without sigma
 * noise (render.py:162-182: alpha = 1 - exp(-relu(sigma) * dist) = 0, weight = alpha * T = 0) and everything downstream of it
 * (rgb / disp / acc maps, inverse-CDF weights, every gradient).  The sigma channel is always exact. */
#define FN_FWD_SKIP_DEAD_RGB 1
int fastnerf_mlp_bf16_fwd_flags(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                                const float* packed_fwd, float* raw, int flags, fn_stream_t stream);
int fastnerf_mlp_fwd_flags_ex(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                              const float* packed_fwd, float* raw, int flags, fn_stream_t stream);   /* exact-fp32 kernels */

/* ---- fused forward of render_rays (render.py:238-299): coarse sampler -> coarse MLP -> compositing ->
 * [sample_pdf + merge -> fine MLP -> compositing], enqueued by one call on `stream`.  math_mode 0 = exact fp32,
 * 1 = split-bf16 (x3), 2 = bf16x6; packed_* must come from the matching pack entry point; act0 / act1 == NULL: inference.  perturb /
 * t_rand / seed0 as fastnerf_sample_coarse, det / u / seed1 as fastnerf_sample_pdf_merge, noise* as
 * fastnerf_raw2outputs_fwd.  N_importance == 0: coarse pass only (the *_f / *1 arguments are ignored).  All buffers
 * are caller-owned device memory with the shapes of the individual entry points. */
int fastnerf_render_rays_fwd(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11, int lindisp,
                             int perturb, int det, int white_bkgd, const float* t_rand, const float* u,
                             const float* noise0, const float* noise1, uint64_t seed0, uint64_t seed1,
                             const float* params_c, const float* packed_c, const float* params_f, const float* packed_f,
                             float* z0, float* raw0, float* act0, float* rgb0, float* disp0, float* acc0, float* w0,
                             float* depth0, float* z1, float* z_samples, float* z_std, float* raw1, float* act1,
                             float* rgb1, float* disp1, float* acc1, float* w1, float* depth1, fn_stream_t stream);
/* The same with fastnerf_mlp_bf16_fwd_flags options for its inference launches (act0 / act1 == NULL, noise0 / noise1 == NULL;
 * ignored otherwise): what the first pass of a compacted training step uses -- raw0 / raw1 then carry colour
 * logits 0 on tiles without a live sample, every other output is bit-identical. */
int fastnerf_render_rays_fwd_ex(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11, int lindisp,
                                int perturb, int det, int white_bkgd, const float* t_rand, const float* u,
                                const float* noise0, const float* noise1, uint64_t seed0, uint64_t seed1,
                                const float* params_c, const float* packed_c, const float* params_f, const float* packed_f,
                                float* z0, float* raw0, float* act0, float* rgb0, float* disp0, float* acc0, float* w0,
                                float* depth0, float* z1, float* z_samples, float* z_std, float* raw1, float* act1,
                                float* rgb1, float* disp1, float* acc1, float* w1, float* depth1, int flags, fn_stream_t stream);

/* Backward of the same chain w.r.t. the network parameters (what loss.backward() does for this path,
 * run_nerf.py:493): fine pass into grads_f, coarse pass into grads_c (sample positions are detached, so the coarse net
 * only sees g_rgb0).  Two distinct nets when N_importance > 0.  draw_ws: n*(N_samples+N_importance)*4 floats;
 * dact_ws / partial_ws as for the mlp_bwd entry points. */
int fastnerf_render_rays_bwd(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11, int white_bkgd,
                             const float* g_rgb, const float* g_rgb0, const float* noise0, const float* noise1,
                             const float* z0, const float* raw0, const float* act0, const float* z1, const float* raw1,
                             const float* act1, const float* params_c, const float* packed_bwd_c, const float* params_f,
                             const float* packed_bwd_f, float* draw_ws, float* dact_ws, float* partial_ws, float* grads_c,
                             float* grads_f, fn_stream_t stream);

/* ---- nerf++-ours additions ------------------------------------------- */
/* get_rays_single_image (nerf_sample_ray_split.py:10-34): intrinsics_host / c2w_host are 4x4 row-major
 * doubles; rays_o / rays_d [H*W,3]. */
int fastnerf_pp_gen_rays(int H, int W, const double* intrinsics_host, const double* c2w_host, float* rays_o,
                         float* rays_d, fn_stream_t stream);
/* nerf++ quadtree fork: per-(image, leaf) sum of |gt-pred| over rays and channels (fp64) and ray count
 * feeding the MEAN split criterion (nerf++-ours/tree.py:622); the caller zeroes sum / count.  A ray's term is rounded to a
 * multiple of 2^-30 before it is added: the sums are exact (< 2^23), order independent and shard independent. */
int fastnerf_leaf_sumcount(int64_t n, const float* rgb, const float* target, const int32_t* leaf_tag, int max_leaves,
                           double* sum, int32_t* count, fn_stream_t stream);
/* intersect_sphere (ddp_train_nerf.py:54-69); *n_outside counts rays whose camera is not inside the
 * unit sphere (the reference raises in that case). */
int fastnerf_pp_intersect_sphere(int64_t n, const float* rays11, float* fg_far, int* n_outside, fn_stream_t stream);
/* foreground depths near + i*step, optionally perturbed (ddp_train_nerf.py:355-361, 72-81). */
int fastnerf_pp_fg_depths(int64_t n, int S, float near, const float* fg_far, int perturb, const float* t_rand,
                          uint64_t seed, float* z, fn_stream_t stream);
/* nerf++ sample_pdf + sort-merge (ddp_train_nerf.py:84-133, 369-382). */
int fastnerf_pp_sample_pdf_merge(int64_t n, int S, int Ni, const float* z, const float* weights, int det,
                                 const float* u, uint64_t seed, float* z_out, float* z_samples, fn_stream_t stream);
/* perturb_samples (ddp_train_nerf.py:72-81): z_out = lower + (upper - lower) * u over the mid-point intervals of the sorted
 * depths z_in [n,S]; t_rand [n,S] injected or NULL (Philox(seed)).  z_out must not alias z_in. */
int fastnerf_pp_perturb_samples(int64_t n, int S, const float* z_in, const float* t_rand, uint64_t seed, float* z_out,
                                fn_stream_t stream);
/* stand-alone nerf++ sample_pdf (ddp_train_nerf.py:84-133): bins [n,M], weights [n,M-1] -> samples [n,Ni]. */
int fastnerf_pp_sample_pdf(int64_t n, int M, int Ni, const float* bins, const float* weights, int det, const float* u,
                           uint64_t seed, float* samples, fn_stream_t stream);
/* depth2pts_outside (ddp_model.py:16-45): ray_o / ray_d [n,3], depth [n,S] (inverse distance to the sphere origin) ->
 * pts [n,S,4] = (x', y', z', 1/r), depth_real [n,S] (may be NULL). */
int fastnerf_pp_depth2pts_outside(int64_t n, int S, const float* ray_o, const float* ray_d, const float* depth, float* pts,
                                  float* depth_real, fn_stream_t stream);
/* fg (part 0) / bg (part 1) compositing of NerfNet.forward (ddp_model.py:97-135) and its backward.
 * raw is the MLP output in network order (bg: far->near); z is always stored near->far.
 * fwd outputs: rgb_map [n,3], weights [n,S], depth [n] (may be NULL), lambda [n] (part 0 only).
 * bwd inputs: g_rgb [n,3], g_lambda [n] (part 0; may be NULL) -> draw [n,S,4]. */
int fastnerf_pp_composite_fwd(int64_t n, int S, int part, const float* raw, const float* z, const float* rays11,
                              const float* fg_far, float* rgb_map, float* weights, float* depth, float* lambda,
                              fn_stream_t stream);
int fastnerf_pp_composite_bwd(int64_t n, int S, int part, const float* raw, const float* z, const float* rays11,
                              const float* fg_far, const float* g_rgb, const float* g_lambda, float* draw,
                              fn_stream_t stream);

/* ---- host quadtree (tree.py), no device work --------------------------- */
typedef struct fn_tree fn_tree; /* opaque: per-image DFS leaf lists */
fn_tree* fastnerf_tree_create(int H, int W, int n_images, int max_depth);
void fastnerf_tree_destroy(fn_tree* t);
int fastnerf_tree_num_leaves(const fn_tree* t, int image);
int fastnerf_tree_max_leaves(const fn_tree* t);
double fastnerf_tree_min_area(const fn_tree* t, int image);
/* leaves of one image in DFS enumeration order: out [n_leaves,4] doubles x0,y0,x1,y1 */
int fastnerf_tree_get_leaves(const fn_tree* t, int image, double* out_host);
int fastnerf_tree_set_leaves(fn_tree* t, int image, int n_leaves, const double* boxes_host, double min_area);
/* per-leaf ray count + integer pixel ranges (tree.py:578-581, 598-599):
 * out [n_leaves,5] int32: count, row_lo, row_hi, col_lo, col_hi (hi exclusive).
 * last_epoch!=0 evaluates a fresh depth-1 tree (tree.py:390-400). */
int fastnerf_tree_leaf_plan(const fn_tree* t, int image, double ray_num_per_pixel, int last_epoch,
                            int32_t* out_host);
/* adjust_tree_multiThread (tree.py:533-557, 629-652) driven by the reduced
 * table: table_host [n_images, max_leaves] floats (max |gt-pred| per leaf,
 * negative = leaf had no ray).  Returns total leaves after the split, <0 on error. */
int64_t fastnerf_tree_adjust(fn_tree* t, const float* table_host, int max_leaves, double thres);
/* nerf++ fork: mean criterion from fp64 sums + counts ([n_images, max_leaves] each). */
int64_t fastnerf_tree_adjust_mean(fn_tree* t, const double* sum_host, const int32_t* count_host, int max_leaves,
                                  double thres);


/* ---- epoch ray generation on the device (gen_rays_v3_multiThread + gen_rays_v3_1_subThread, tree.py:377-428, 569-626;
 * the variance-weighted picks of nerf++-ours/tree.py:566-578 + image_process.py:58-93) ------------------------------
 * fastnerf_tree_epoch_plan (host): the plans of all trees in one call, rows of 7 int32 = image, leaf, ray count,
 * row_lo, row_hi, col_lo, col_hi; out_host may be NULL to query sizes; returns rows, *n_rays_host = total rays.
 * fastnerf_epoch_rays (device): N rows (rays_o, rays_d, rgb [N,3], tag [N,2] = image, leaf; pix [N,3] optional) in
 * their final shuffled order (replaces the per-leaf torch.randint draws, the [n,H,W,3] gathers and the epoch's
 * torch.randperm).  plan [L,7] and offs [L+1] (int64 exclusive prefix sums of the counts) are device copies of the
 * host plan; images [n_img,H,W,3], poses [n_img,3,4].  Weighted picks (all five pointers or none): per leaf the first
 * n_weighted[l] rays are drawn with probability proportional to the weights whose running sum over the pixels sorted
 * by leaf is cum (fp64); order = flat pixel ids in that order; seg_beg / seg_end = the leaf's range in it. */
int64_t fastnerf_tree_epoch_plan(const fn_tree* t, double ray_num_per_pixel, int last_epoch, int32_t* out_host,
                                 int64_t* n_rays_host);
int fastnerf_epoch_rays(int64_t N, int L, const int32_t* plan, const int64_t* offs, const float* images,
                        const float* poses, int n_img, int H, int W, float fx, float fy, float cx, float cy,
                        uint64_t seed, int shuffle, const int32_t* n_weighted, const int64_t* seg_beg,
                        const int64_t* seg_end, const int32_t* order, const double* cum, float* rays_o, float* rays_d,
                        float* rgb, int32_t* tag, int32_t* pix, fn_stream_t stream);
/* One rank's rows of the same epoch (data parallel): the reference cuts the shuffled epoch into batches of `batch` = N_rand
 * consecutive rows (run_nerf.py:472-478); rank row0 of `stride` ranks steps rows row0 :: stride of every batch.  Only those rows are
 * generated -- output row r = epoch row (r / per) * batch + row0 + (r % per) * stride, per = ceil((batch - row0) / stride) -- bit-identical to
 * the corresponding rows of fastnerf_epoch_rays with the same seed (the row -> ray map is a keyed bijection of the row index).
 * fastnerf_epoch_shard_rows: the number of output rows (host arithmetic; -1 on bad arguments). */
int64_t fastnerf_epoch_shard_rows(int64_t N, int64_t batch, int row0, int stride);
int fastnerf_epoch_rays_shard(int64_t N, int L, const int32_t* plan, const int64_t* offs, const float* images,
                              const float* poses, int n_img, int H, int W, float fx, float fy, float cx, float cy,
                              uint64_t seed, int shuffle, const int32_t* n_weighted, const int64_t* seg_beg,
                              const int64_t* seg_end, const int32_t* order, const double* cum, int64_t batch, int row0, int stride,
                              float* rays_o, float* rays_d, float* rgb, int32_t* tag, int32_t* pix, fn_stream_t stream);

/* sigma noise (render.py:162, `torch.randn(raw[..., 3].shape) * raw_noise_std`): out[0..n) = N(0, std^2) draws of a Philox4x32-10 stream keyed
 * by seed (Box-Muller), one launch for the noise of both passes of a render_rays call.  out must be 16-byte aligned. */
int fastnerf_gauss_noise(int64_t n, float std, uint64_t seed, float* out, fn_stream_t stream);

/* ---- exact zero-gradient point compaction of the training backward -------------------------------------------
 * loss.backward() (run_nerf.py:493) spends most of its time on samples whose d(loss)/d(raw) is exactly zero
 * (sigma + noise <= 0 => alpha = 0 => weight = 0 and relu' = 0; render.py:162,182): all of their pre-activation
 * gradients are exact zeros.  These entry points run the backward on the other ("live") samples only.
 * fastnerf_compact_live: live_idx[0..count) = ascending indices of the points p with draw[p*4..p*4+3] != 0,
 * count_out[0] = count, count_out[1] = n_points (device values: no host round trip).  ws: fastnerf_compact_ws_ints()
 * int32 of scratch. */
int64_t fastnerf_compact_ws_ints(int64_t n_points);
int fastnerf_compact_live(int64_t n_points, const float* draw, int32_t* live_idx, int32_t* count_out, int32_t* ws,
                          fn_stream_t stream);
/* NeRF.forward + Embedder.embed (model.py:38-63) over a live list, saving activations for the backward in list
 * order; act sized by fastnerf_mlp_bf16_floats(kind, 3, n*S). */
int fastnerf_mlp_bf16_fwd_live(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                               const float* packed_fwd, float* act, const int32_t* live_idx, const int32_t* live_cnt,
                               fn_stream_t stream);
/* backward of the MLP over the same list: draw is the full [n*S,4] gradient, read through live_idx. */
int fastnerf_mlp_bf16_bwd_live(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                               const float* packed_bwd, float* dact, float* partial, float* grads,
                               const int32_t* live_idx, const int32_t* live_cnt, fn_stream_t stream);
/* the exact-fp32 twins (v_mfma_f32_32x32x2_f32 kernels; act sized by fastnerf_mlp_act_floats) */
int fastnerf_mlp_fwd_live_ex(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                             const float* packed_fwd, float* act, const int32_t* live_idx, const int32_t* live_cnt,
                             fn_stream_t stream);
int fastnerf_mlp_bwd_live_ex(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                             const float* packed_bwd, float* dact, float* partial, float* grads,
                             const int32_t* live_idx, const int32_t* live_cnt, fn_stream_t stream);
/* the whole backward of render_rays (render.py:238-299 under loss.backward()) with compaction, for a forward that
 * saved nothing: per pass compositing backward -> live list -> saving forward over the list -> dX / dW
 * (math_mode 1: split-bf16, 0: exact fp32).
 * live_ws: 4 + n*(N_samples+N_importance) + fastnerf_compact_ws_ints(...) int32; counts_out: NULL or 4 int32
 * (live, total of the fine pass; live, total of the coarse pass). */
int fastnerf_render_rays_bwd_live(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11, int white_bkgd,
                                  const float* g_rgb, const float* g_rgb0, const float* noise0, const float* noise1,
                                  const float* z0, const float* raw0, const float* z1, const float* raw1,
                                  const float* params_c, const float* packed_fwd_c, const float* packed_bwd_c,
                                  const float* params_f, const float* packed_fwd_f, const float* packed_bwd_f,
                                  float* draw_ws, float* act_ws, float* dact_ws, float* partial_ws, int32_t* live_ws,
                                  float* grads_c, float* grads_f, int32_t* counts_out, fn_stream_t stream);

/* ---- "bf16x6" math mode: fp32-WIDTH products on the bf16 matrix cores (csrc/mlp_*.hip, MM_X6) -------------------------------
 * Same network functions and call protocol as fastnerf_mlp_pack_ex / fwd_ex / fwd_flags_ex / bwd_ex / fwd_live_ex / bwd_live_ex
 * (run_nerf.py:50-64 run_network -> model.py:37-63, autograd backward of the same).  Every fp32 operand is decomposed EXACTLY
 * into three bf16 pieces (8 + 8 + 8 significand bits) and a product is the sum of the six piece products whose weight is
 * >= 2^-16 of it, accumulated in fp32 on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16 in the forward / dX, v_mfma_f32_32x32x16_bf16 in dW): the dropped terms are <= 2^-24 of the product, the rounding
 * fp32 itself applies to it.  Weights are packed as three bf16 planes in the MFMA's fragment order (opaque; fastnerf_mlp_x6_packed_floats(kind, 1 | 2) floats);
 * saved activations / gradient workspaces are those of the exact-fp32 kernels (fastnerf_mlp_act_floats,
 * n*S*FASTNERF_DACT_FLOATS, fastnerf_mlp_bwd_partial_floats).  fwd: act == NULL -> inference, flags as
 * fastnerf_mlp_fwd_flags_ex (ignored when act != NULL). */
int64_t fastnerf_mlp_x6_packed_floats(int kind, int which);
int fastnerf_mlp_x6_pack(int kind, const float* params, float* packed_fwd, float* packed_bwd, fn_stream_t stream);
int fastnerf_mlp_x6_fwd(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                        const float* packed_fwd, float* raw, float* act, int flags, fn_stream_t stream);
int fastnerf_mlp_x6_bwd(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                        const float* packed_bwd, float* dact, float* partial, float* grads, fn_stream_t stream);
int fastnerf_mlp_x6_fwd_live(int kind, int64_t n, int S, const float* rays11, const float* z, const float* params,
                             const float* packed_fwd, float* act, const int32_t* live_idx, const int32_t* live_cnt,
                             fn_stream_t stream);
int fastnerf_mlp_x6_bwd_live(int kind, int64_t n, int S, const float* draw, const float* act, const float* params,
                             const float* packed_bwd, float* dact, float* partial, float* grads, const int32_t* live_idx,
                             const int32_t* live_cnt, fn_stream_t stream);

/* ---- one optimisation step per call (run_nerf.py:479-508: render -> img2mse fine + coarse -> loss.backward() ->
 * optimizer.step(), the epoch loss map of :505-506 fed inside the loss launch) ----------------------------------------
 * fastnerf_train_step enqueues the phases selected by `phases` on `stream`, using exactly the entry points above in the
 * order the reference's loop implies, so its results are bit-identical to calling them one by one; what it removes is
 * host work (one call instead of ~10 and no per-step allocations: the caller keeps every buffer alive in `args`).
 * Data parallel: FN_STEP_FORWARD | FN_STEP_BWD_FINE, then the caller starts the all-reduce of the fine net's gradient
 * (grads + net_floats) on a side stream, FN_STEP_BWD_COARSE runs beside it, all-reduce of the coarse half, FN_STEP_UPDATE.
 * Two distinct NeRF nets with view directions (or one net when N_importance == 0); coarse net first in params / grads /
 * adam_m / adam_v (the order of `grad_vars`, run_nerf.py:87-97).  Buffer shapes are those of the individual entry points;
 * act0 / act1 are used by the plain backward (live == 0), act_ws / live_ws / counts by the compacted one (live != 0). */
#define FN_STEP_FORWARD 1     /* pack rays, forward, loss + d(loss)/d(rgb maps) + leaf table */
#define FN_STEP_BWD_FINE 2    /* backward of the fine pass -> grads + net_floats (no-op when N_importance == 0) */
#define FN_STEP_BWD_COARSE 4  /* backward of the coarse pass -> grads */
#define FN_STEP_UPDATE 8      /* Adam over both nets + re-pack of their weights */
typedef struct fn_step_args {
  /* the batch */
  int64_t n;
  const float *rays_o, *rays_d, *target;            /* [n,3] each */
  const float *t_rand, *u, *noise0, *noise1;        /* injected randoms or NULL (fastnerf_render_rays_fwd) */
  uint64_t seed0, seed1;
  const int32_t* leaf_tag;                          /* [n,2] or NULL */
  uint32_t* table;                                  /* leaf-error table or NULL */
  /* networks, optimiser state */
  int64_t net_floats;                               /* FASTNERF_NET_PARAMS */
  float *params, *grads, *adam_m, *adam_v;          /* (1 or 2) x net_floats */
  float *packed_fwd_c, *packed_bwd_c, *packed_fwd_f, *packed_bwd_f;
  /* tensors of the step (outputs of the forward, kept for the backward) */
  float *rays11, *z0, *raw0, *act0, *rgb0, *disp0, *acc0, *w0, *depth0;
  float *z1, *z_samples, *z_std, *raw1, *act1, *rgb1, *disp1, *acc1, *w1, *depth1;
  float *g_rgb, *g_rgb0, *loss2;                    /* [n,3], [n,3], [2] */
  float *draw_ws, *act_ws, *dact_ws, *partial_ws;   /* scratch, sized as for fastnerf_render_rays_bwd(_live) */
  int32_t *live_ws, *counts;                        /* compacted backward: scratch, optional int32[4] live / total counts */
  double focal, lr, beta1, beta2, eps;
  float near_plane, far_plane, grad_scale;
  int32_t math_mode;                                /* 0 exact fp32 MFMA, 1 split-bf16 (x3), 2 bf16x6 (fp32 width) */
  int32_t N_samples, N_importance, lindisp, perturb, white_bkgd, ndc, H, W;
  int32_t live;                                     /* != 0: forward without saving + compacted backward */
  int32_t fwd_flags;                                /* FN_FWD_* of the first forward of a compacted step */
  int32_t max_leaves, adam_t;
} fn_step_args;
int64_t fastnerf_step_args_size(void);              /* sizeof(fn_step_args): binding sanity check */
int fastnerf_train_step(const fn_step_args* args, int phases, fn_stream_t stream);

/* ---- exchange steps of the data-parallel path ---------------------------------------------------
 * One process per GPU; every rank renders its shard of the ray batch end to end.  The reference's strategies for contrast:
 * nn.DataParallel around the MLP (nerf-ours/run_nerf.py:70,82,90), torch DDP (nerf++-ours/ddp_train_nerf.py:150-184).
 * A PyTorch host gets the same collectives from torch.distributed (backend "nccl" = RCCL; parallel.py, the default route);
 * these entry points serve a host without it.  librccl is resolved at the first call, the library does not link against it.
 *   fastnerf_comm_unique_id   rank 0: a fresh RCCL id (FASTNERF_COMM_ID_BYTES bytes) to hand to every rank out of band
 *   fastnerf_comm_init        collective over all ranks, on the caller's current HIP device
 *   fastnerf_allreduce_grads  in place SUM over ranks of the flat fp32 gradient buffer (or a slice of it: the fine net's half
 *                             can go while the coarse net's backward runs on another stream), then * scale (1 / world for the
 *                             global-batch mean of img2mse, run_nerf_helpers.py:9); enqueued on `stream`, no host sync
 *   fastnerf_allreduce_leaf_table  in place MAX over ranks of the per-(image, leaf) table of fastnerf_mse_leafmax (uint32 bit
 *                             patterns of non-negative floats: exact, order independent => identical on 1 or 8 GPUs)
 *   fastnerf_allreduce_leaf_sumcount  in place SUM over ranks of the fp64 sums / int32 counts of fastnerf_leaf_sumcount (the
 *                             nerf++ fork's MEAN rule, nerf++-ours/tree.py:609-632); the sums hold multiples of 2^-30, so the
 *                             result is exact: bit-identical to one rank accumulating every ray
 *   fastnerf_leaf_table_reset / _read   zero the device table / copy it to the host as floats and wait for it (the split
 *                             rule of tree.py:629-652 runs on the host: fastnerf_tree_adjust)                              */
#define FASTNERF_COMM_ID_BYTES 128
typedef struct fn_comm fn_comm; /* opaque: one RCCL communicator */
int fastnerf_comm_unique_id(char* id);
int fastnerf_comm_init(fn_comm** out, const char* id, int rank, int world);
int fastnerf_comm_destroy(fn_comm* comm);
int fastnerf_allreduce_grads(fn_comm* comm, float* grads, int64_t n, float scale, fn_stream_t stream);
int fastnerf_allreduce_leaf_table(fn_comm* comm, uint32_t* table, int64_t n, fn_stream_t stream);
int fastnerf_allreduce_leaf_sumcount(fn_comm* comm, double* sum, int32_t* count, int64_t n, fn_stream_t stream);
int fastnerf_leaf_table_reset(uint32_t* table, int64_t n, fn_stream_t stream);
int fastnerf_leaf_table_read(const uint32_t* table, float* host_out, int64_t n, fn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
