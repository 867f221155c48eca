// render.cpp -- host-side sequencing of the fused forward of render_rays (render.py:238-299): one C-ABI call
// enqueues  coarse sampler -> coarse MLP -> compositing -> [inverse-CDF + merge -> fine MLP -> compositing]
// on the caller's stream.  No kernels here; every stage is one of the library's own entry points.
#include <stdint.h>
#include "../../include/fastnerf.h"

namespace fn { void set_error(const char* fmt, ...); }

extern "C" int fastnerf_render_rays_fwd_ex(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11,
                                        int lindisp, int perturb, int det, int white_bkgd, const float* t_rand, const float* u,
                                        const float* noise0, const float* noise1, uint64_t seed0, uint64_t seed1,
                                        const float* params_c, const float* packed_c, const float* params_f,
                                        const float* packed_f, float* z0, float* raw0, float* act0, float* rgb0,
                                        float* disp0, float* acc0, float* w0, float* depth0, float* z1,
                                        float* z_samples, float* z_std, float* raw1, float* act1, float* rgb1,
                                        float* disp1, float* acc1, float* w1, float* depth1, int flags, fn_stream_t stream) {
  if (math_mode < 0 || math_mode > 2 || n < 0 || N_samples < 2 || N_importance < 0) {
    fn::set_error("fastnerf_render_rays_fwd: bad argument: math_mode in {0,1,2}, n>=0, N_samples>=2, N_importance>=0");
    return -1;
  }
  if (N_importance > 0 && N_samples < 3) {
    // the reference fails here too: weights[..., 1:-1] is empty and sample_pdf indexes an empty cdf (run_nerf_helpers.py:147)
    fn::set_error("fastnerf_render_rays_fwd: hierarchical sampling needs N_samples >= 3 (the inner weights of 2 samples are empty)");
    return -1;
  }
  if (n == 0) return 0;
  if (!rays11 || !params_c || !packed_c || !z0 || !raw0 || !rgb0 || !disp0 || !acc0 || !w0 || !depth0) {
    fn::set_error("fastnerf_render_rays_fwd: null pointer (coarse pass)");
    return -1;
  }
  int rc;
  // (options only where their precondition holds: an inference launch, no sigma noise in that pass)
  auto mlp = [&](int64_t nn, int S, const float* z, const float* params, const float* packed, float* raw, float* act,
                 const float* noise) {
    if (math_mode == 2) return fastnerf_mlp_x6_fwd(0, nn, S, rays11, z, params, packed, raw, act, (!act && !noise) ? flags : 0, stream);
    if (!act && !noise && flags)
      return math_mode ? fastnerf_mlp_bf16_fwd_flags(0, nn, S, rays11, z, params, packed, raw, flags, stream)
                       : fastnerf_mlp_fwd_flags_ex(0, nn, S, rays11, z, params, packed, raw, flags, stream);
    return math_mode ? fastnerf_mlp_bf16_fwd(0, nn, S, rays11, z, params, packed, raw, act, stream)
                     : fastnerf_mlp_fwd_ex(0, nn, S, rays11, z, params, packed, raw, act, stream);
  };
  if ((rc = fastnerf_sample_coarse(n, N_samples, rays11, lindisp, perturb, t_rand, seed0, z0, stream))) return rc;
  if ((rc = mlp(n, N_samples, z0, params_c, packed_c, raw0, act0, noise0))) return rc;
  if ((rc = fastnerf_raw2outputs_fwd(n, N_samples, raw0, z0, rays11, noise0, white_bkgd, rgb0, disp0, acc0, w0, depth0, stream)))
    return rc;
  if (N_importance == 0) return 0;
  if (!params_f || !packed_f || !z1 || !z_samples || !z_std || !raw1 || !rgb1 || !disp1 || !acc1 || !w1 || !depth1) {
    fn::set_error("fastnerf_render_rays_fwd: null pointer (fine pass)");
    return -1;
  }
  const int S1 = N_samples + N_importance;
  if ((rc = fastnerf_sample_pdf_merge(n, N_samples, N_importance, z0, w0, det, u, seed1, z1, z_samples, z_std, stream)))
    return rc;
  if ((rc = mlp(n, S1, z1, params_f, packed_f, raw1, act1, noise1))) return rc;
  return fastnerf_raw2outputs_fwd(n, S1, raw1, z1, rays11, noise1, white_bkgd, rgb1, disp1, acc1, w1, depth1, stream);
}

// Backward of the same chain (autograd of render.py:238-299 w.r.t. the network parameters; sample positions are
// detached in the reference, so the coarse net only sees d(loss)/d(rgb0)): compositing backward -> MLP backward for the
// fine pass (into grads_f) and the coarse pass (into grads_c).  draw_ws: n * (N_samples + N_importance) * 4 floats.
extern "C" int fastnerf_render_rays_fwd(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11,
                                        int lindisp, int perturb, int det, int white_bkgd, const float* t_rand, const float* u,
                                        const float* noise0, const float* noise1, uint64_t seed0, uint64_t seed1,
                                        const float* params_c, const float* packed_c, const float* params_f,
                                        const float* packed_f, float* z0, float* raw0, float* act0, float* rgb0,
                                        float* disp0, float* acc0, float* w0, float* depth0, float* z1,
                                        float* z_samples, float* z_std, float* raw1, float* act1, float* rgb1,
                                        float* disp1, float* acc1, float* w1, float* depth1, fn_stream_t stream) {
  return fastnerf_render_rays_fwd_ex(math_mode, n, N_samples, N_importance, rays11, lindisp, perturb, det, white_bkgd, t_rand, u,
                                     noise0, noise1, seed0, seed1, params_c, packed_c, params_f, packed_f, z0, raw0, act0, rgb0,
                                     disp0, acc0, w0, depth0, z1, z_samples, z_std, raw1, act1, rgb1, disp1, acc1, w1, depth1, 0,
                                     stream);
}

// passes: bit 0 = the fine pass (N_importance > 0 only), bit 1 = the coarse pass (the only one when N_importance == 0)
static int rr_bwd(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11,
                  int white_bkgd, const float* g_rgb, const float* g_rgb0, const float* noise0,
                  const float* noise1, const float* z0, const float* raw0, const float* act0,
                  const float* z1, const float* raw1, const float* act1, const float* params_c,
                  const float* packed_bwd_c, const float* params_f, const float* packed_bwd_f,
                  float* draw_ws, float* dact_ws, float* partial_ws, float* grads_c, float* grads_f,
                  int passes, fn_stream_t stream) {
  if (math_mode < 0 || math_mode > 2 || n <= 0 || N_samples < 2 || N_importance < 0) {
    fn::set_error("fastnerf_render_rays_bwd: bad argument: math_mode in {0,1,2}, n>0, N_samples>=2, N_importance>=0");
    return -1;
  }
  if (!rays11 || !z0 || !raw0 || !act0 || !params_c || !packed_bwd_c || !draw_ws || !dact_ws || !partial_ws || !grads_c) {
    fn::set_error("fastnerf_render_rays_bwd: null pointer (coarse pass)");
    return -1;
  }
  int rc;
  auto mlp = [&](int S, const float* act, const float* params, const float* packed, float* grads) {
    if (math_mode == 2) return fastnerf_mlp_x6_bwd(0, n, S, draw_ws, act, params, packed, dact_ws, partial_ws, grads, stream);
    return math_mode ? fastnerf_mlp_bf16_bwd(0, n, S, draw_ws, act, params, packed, dact_ws, partial_ws, grads, stream)
                     : fastnerf_mlp_bwd_ex(0, n, S, draw_ws, act, params, packed, dact_ws, partial_ws, grads, stream);
  };
  const float* g_coarse = g_rgb;
  if (N_importance > 0) {
    if (!g_rgb || !g_rgb0 || !z1 || !raw1 || !act1 || !params_f || !packed_bwd_f || !grads_f) {
      fn::set_error("fastnerf_render_rays_bwd: null pointer (fine pass)");
      return -1;
    }
    const int S1 = N_samples + N_importance;
    if (passes & 1) {
      if ((rc = fastnerf_raw2outputs_bwd(n, S1, raw1, z1, rays11, noise1, white_bkgd, g_rgb, draw_ws, stream))) return rc;
      if ((rc = mlp(S1, act1, params_f, packed_bwd_f, grads_f))) return rc;
    }
    g_coarse = g_rgb0;
  }
  if (!g_coarse) {
    fn::set_error("fastnerf_render_rays_bwd: null gradient");
    return -1;
  }
  if (!(passes & 2)) return 0;
  if ((rc = fastnerf_raw2outputs_bwd(n, N_samples, raw0, z0, rays11, noise0, white_bkgd, g_coarse, draw_ws, stream))) return rc;
  return mlp(N_samples, act0, params_c, packed_bwd_c, grads_c);
}

extern "C" int fastnerf_render_rays_bwd(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11,
                                        int white_bkgd, const float* g_rgb, const float* g_rgb0, const float* noise0,
                                        const float* noise1, const float* z0, const float* raw0, const float* act0,
                                        const float* z1, const float* raw1, const float* act1, const float* params_c,
                                        const float* packed_bwd_c, const float* params_f, const float* packed_bwd_f,
                                        float* draw_ws, float* dact_ws, float* partial_ws, float* grads_c, float* grads_f,
                                        fn_stream_t stream) {
  return rr_bwd(math_mode, n, N_samples, N_importance, rays11, white_bkgd, g_rgb, g_rgb0, noise0, noise1, z0, raw0, act0, z1, raw1,
                act1, params_c, packed_bwd_c, params_f, packed_bwd_f, draw_ws, dact_ws, partial_ws, grads_c, grads_f, 3, stream);
}


// Training backward with exact zero-gradient point compaction (math_mode 1: split-bf16 kernels, 0: exact-fp32 kernels).  The forward ran WITHOUT saving
// activations (the inference kernels); per pass:  compositing backward -> list of the points with a non-zero
// d(loss)/d(raw) (fastnerf_compact_live) -> forward over that list, saving activations -> dX / dW over that list.
// The gradients equal fastnerf_render_rays_bwd's up to fp32 summation order (the dead points' terms are exact zeros).
// No host round trip: the list length stays on the device.  live_ws: 4 + n*S1 + fastnerf_compact_ws_ints(n*S1) int32;
// act_ws: fastnerf_mlp_bf16_floats(0, 3, n*S1) / fastnerf_mlp_act_floats(0, n*S1) floats; counts_out (optional): 4 int32 = live/total fine, live/total coarse.
static int rr_bwd_live(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11, int white_bkgd,
                       const float* g_rgb, const float* g_rgb0, const float* noise0, const float* noise1,
                       const float* z0, const float* raw0, const float* z1, const float* raw1,
                       const float* params_c, const float* packed_fwd_c, const float* packed_bwd_c,
                       const float* params_f, const float* packed_fwd_f, const float* packed_bwd_f,
                       float* draw_ws, float* act_ws, float* dact_ws, float* partial_ws, int32_t* live_ws,
                       float* grads_c, float* grads_f, int32_t* counts_out, int passes, fn_stream_t stream) {
  if (math_mode < 0 || math_mode > 2 || n <= 0 || N_samples < 2 || N_importance < 0) {
    fn::set_error("fastnerf_render_rays_bwd_live: bad argument: math_mode in {0,1,2}, n>0, N_samples>=2, N_importance>=0");
    return -1;
  }
  if (!rays11 || !z0 || !raw0 || !params_c || !packed_fwd_c || !packed_bwd_c || !draw_ws || !act_ws || !dact_ws ||
      !partial_ws || !live_ws || !grads_c) {
    fn::set_error("fastnerf_render_rays_bwd_live: null pointer (coarse pass)");
    return -1;
  }
  const int S1 = N_samples + N_importance;
  int32_t* cnt = live_ws;            // [4]: (live, total) of the fine pass, of the coarse pass
  int32_t* idx = live_ws + 4;
  int32_t* cws = idx + n * (int64_t)S1;
  int rc;
  auto pass = [&](int S, const float* z, const float* raw, const float* noise, const float* g, const float* params,
                  const float* pf, const float* pb, float* grads, int32_t* cnt_out) -> int {
    if ((rc = fastnerf_raw2outputs_bwd(n, S, raw, z, rays11, noise, white_bkgd, g, draw_ws, stream))) return rc;
    if ((rc = fastnerf_compact_live(n * (int64_t)S, draw_ws, idx, cnt_out, cws, stream))) return rc;
    if (math_mode == 2) {
      if ((rc = fastnerf_mlp_x6_fwd_live(0, n, S, rays11, z, params, pf, act_ws, idx, cnt_out, stream))) return rc;
      return fastnerf_mlp_x6_bwd_live(0, n, S, draw_ws, act_ws, params, pb, dact_ws, partial_ws, grads, idx, cnt_out, stream);
    }
    if (math_mode) {
      if ((rc = fastnerf_mlp_bf16_fwd_live(0, n, S, rays11, z, params, pf, act_ws, idx, cnt_out, stream))) return rc;
      return fastnerf_mlp_bf16_bwd_live(0, n, S, draw_ws, act_ws, params, pb, dact_ws, partial_ws, grads, idx, cnt_out, stream);
    }
    if ((rc = fastnerf_mlp_fwd_live_ex(0, n, S, rays11, z, params, pf, act_ws, idx, cnt_out, stream))) return rc;
    return fastnerf_mlp_bwd_live_ex(0, n, S, draw_ws, act_ws, params, pb, dact_ws, partial_ws, grads, idx, cnt_out, stream);
  };
  const float* g_coarse = g_rgb;
  int32_t* c_fine = counts_out ? counts_out : cnt;
  int32_t* c_coarse = counts_out ? counts_out + 2 : cnt + 2;
  if (N_importance > 0) {
    if (!g_rgb || !g_rgb0 || !z1 || !raw1 || !params_f || !packed_fwd_f || !packed_bwd_f || !grads_f) {
      fn::set_error("fastnerf_render_rays_bwd_live: null pointer (fine pass)");
      return -1;
    }
    if ((passes & 1) && (rc = pass(S1, z1, raw1, noise1, g_rgb, params_f, packed_fwd_f, packed_bwd_f, grads_f, c_fine))) return rc;
    g_coarse = g_rgb0;
  }
  if (!g_coarse) {
    fn::set_error("fastnerf_render_rays_bwd_live: null gradient");
    return -1;
  }
  if (!(passes & 2)) return 0;
  return pass(N_samples, z0, raw0, noise0, g_coarse, params_c, packed_fwd_c, packed_bwd_c, grads_c, c_coarse);
}

extern "C" int fastnerf_render_rays_bwd_live(int math_mode, int64_t n, int N_samples, int N_importance, const float* rays11, int white_bkgd,
                                             const float* g_rgb, const float* g_rgb0, const float* noise0, const float* noise1,
                                             const float* z0, const float* raw0, const float* z1, const float* raw1,
                                             const float* params_c, const float* packed_fwd_c, const float* packed_bwd_c,
                                             const float* params_f, const float* packed_fwd_f, const float* packed_bwd_f,
                                             float* draw_ws, float* act_ws, float* dact_ws, float* partial_ws, int32_t* live_ws,
                                             float* grads_c, float* grads_f, int32_t* counts_out, fn_stream_t stream) {
  return rr_bwd_live(math_mode, n, N_samples, N_importance, rays11, white_bkgd, g_rgb, g_rgb0, noise0, noise1, z0, raw0, z1, raw1,
                     params_c, packed_fwd_c, packed_bwd_c, params_f, packed_fwd_f, packed_bwd_f, draw_ws, act_ws, dact_ws,
                     partial_ws, live_ws, grads_c, grads_f, counts_out, 3, stream);
}


// ---------------------------------------------------------------------------------------------------------------------
// One optimisation step of the reference's loop (run_nerf.py:479-508: render -> img2mse (fine + coarse) -> loss.backward() ->
// optimizer.step(); the leaf-error table of :505-506 is fed inside the loss launch) enqueued by ONE call -- or by one call
// per phase when the caller interleaves its gradient all-reduce (data parallel: the fine net's gradient is final after
// FN_STEP_BWD_FINE and travels while FN_STEP_BWD_COARSE runs).  Exactly the launches the entry points above make, in the same
// order, on the caller's stream: results are bit-identical to calling them one by one.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int64_t fastnerf_step_args_size(void) { return (int64_t)sizeof(fn_step_args); }

extern "C" int fastnerf_train_step(const fn_step_args* a, int phases, fn_stream_t stream) {
  if (!a || a->math_mode < 0 || a->math_mode > 2 || a->n <= 0 || a->N_samples < 2 || a->N_importance < 0) {
    fn::set_error("fastnerf_train_step: bad argument: args != NULL, math_mode in {0,1,2}, n>0, N_samples>=2, N_importance>=0");
    return -1;
  }
  const bool two = a->N_importance > 0;
  if (!a->params || !a->grads || !a->packed_fwd_c || !a->packed_bwd_c || (two && (!a->packed_fwd_f || !a->packed_bwd_f)) ||
      a->net_floats <= 0) {
    fn::set_error("fastnerf_train_step: null network buffer");
    return -1;
  }
  const float* params_c = a->params;
  const float* params_f = two ? a->params + a->net_floats : nullptr;
  float* grads_c = a->grads;
  float* grads_f = two ? a->grads + a->net_floats : nullptr;
  const float* g_fine = two ? a->g_rgb : nullptr;       // d(loss)/d(rgb_map) of the pass that produces the image
  int rc;
  if (phases & FN_STEP_FORWARD) {
    if (!a->rays_o || !a->rays_d || !a->target || !a->rays11 || !a->g_rgb || (two && !a->g_rgb0) || !a->loss2) {
      fn::set_error("fastnerf_train_step: null batch / output buffer");
      return -1;
    }
    if ((rc = fastnerf_pack_rays(a->n, a->rays_o, a->rays_d, a->near_plane, a->far_plane, a->ndc, a->H, a->W, a->focal, a->rays11,
                                 stream))) return rc;
    const bool save = !a->live;
    if ((rc = fastnerf_render_rays_fwd_ex(a->math_mode, a->n, a->N_samples, a->N_importance, a->rays11, a->lindisp,
                                          (a->perturb || a->t_rand) ? 1 : 0, a->perturb ? 0 : 1, a->white_bkgd, a->t_rand, a->u,
                                          a->noise0, a->noise1, a->seed0, a->seed1, params_c, a->packed_fwd_c, params_f,
                                          a->packed_fwd_f, a->z0, a->raw0, save ? a->act0 : nullptr, a->rgb0, a->disp0, a->acc0,
                                          a->w0, a->depth0, a->z1, a->z_samples, a->z_std, a->raw1, save ? a->act1 : nullptr,
                                          a->rgb1, a->disp1, a->acc1, a->w1, a->depth1, save ? 0 : a->fwd_flags, stream)))
      return rc;
    if ((rc = fastnerf_mse_leafmax(a->n, two ? a->rgb1 : a->rgb0, two ? a->rgb0 : nullptr, a->target, a->grad_scale, a->g_rgb,
                                   two ? a->g_rgb0 : nullptr, a->loss2, a->leaf_tag, a->max_leaves, a->table, stream)))
      return rc;
  }
  const int passes = ((phases & FN_STEP_BWD_FINE) ? 1 : 0) | ((phases & FN_STEP_BWD_COARSE) ? 2 : 0);
  if (passes) {
    const float* g_a = two ? g_fine : a->g_rgb;
    const float* g_b = two ? a->g_rgb0 : nullptr;
    if (a->live)
      rc = rr_bwd_live(a->math_mode, a->n, a->N_samples, a->N_importance, a->rays11, a->white_bkgd, g_a, g_b, a->noise0, a->noise1,
                       a->z0, a->raw0, a->z1, a->raw1, params_c, a->packed_fwd_c, a->packed_bwd_c, params_f, a->packed_fwd_f,
                       a->packed_bwd_f, a->draw_ws, a->act_ws, a->dact_ws, a->partial_ws, a->live_ws, grads_c, grads_f, a->counts,
                       passes, stream);
    else
      rc = rr_bwd(a->math_mode, a->n, a->N_samples, a->N_importance, a->rays11, a->white_bkgd, g_a, g_b, a->noise0, a->noise1, a->z0,
                  a->raw0, a->act0, a->z1, a->raw1, a->act1, params_c, a->packed_bwd_c, params_f, a->packed_bwd_f, a->draw_ws,
                  a->dact_ws, a->partial_ws, grads_c, grads_f, passes, stream);
    if (rc) return rc;
  }
  if (phases & FN_STEP_UPDATE) {
    if (!a->adam_m || !a->adam_v || a->adam_t < 1) {
      fn::set_error("fastnerf_train_step: update phase needs adam_m, adam_v and adam_t >= 1");
      return -1;
    }
    const int64_t total = a->net_floats * (two ? 2 : 1);
    if ((rc = fastnerf_adam_step(total, a->params, a->grads, a->adam_m, a->adam_v, a->lr, a->beta1, a->beta2, a->eps, a->adam_t,
                                 stream))) return rc;
    auto pack = [&](const float* p, float* pf, float* pb) {
      if (a->math_mode == 2) return fastnerf_mlp_x6_pack(0, p, pf, pb, stream);
      return a->math_mode ? fastnerf_mlp_bf16_pack(0, p, pf, pb, stream) : fastnerf_mlp_pack_ex(0, p, pf, pb, stream);
    };
    if ((rc = pack(params_c, a->packed_fwd_c, a->packed_bwd_c))) return rc;
    if (two && (rc = pack(params_f, a->packed_fwd_f, a->packed_bwd_f))) return rc;
  }
  return 0;
}
