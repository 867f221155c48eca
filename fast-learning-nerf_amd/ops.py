"""Thin tensor-level wrappers over the C ABI (one per entry point).

PyTorch is plumbing here: it owns device memory and the stream; every op calls
straight into libfastnerf.so with raw pointers.  No op has a CPU fallback."""
import os
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, require_gpu, stream  # noqa: F401 (re-exported: tree.py uses ops.ptr / ops.stream)

NET_PARAMS = 595844
PACKED_FWD = 593920
PACKED_BWD = 557056
ACT_FLOATS = 2592
ACT_SLACK = 8192
DACT_FLOATS = 2432


# ---- matrix-core math mode of the 8x256 MLP kernels ---------------------------------------------
# 'fp32'   : v_mfma_f32_32x32x2_f32 (csrc/mlp_*.hip), every kind: the products and sums of an fp32 FMA chain
# 'bf16x6' : csrc/mlp_*.hip MM_X6 -- every fp32 operand decomposed EXACTLY into three bf16 pieces, a product = its six piece
#            products of weight >= 2^-16, fp32 accumulation on v_mfma_f32_32x32x16_bf16: fp32-WIDTH products (the dropped terms
#            are <= 2^-24 of the product) at 2.67x the matrix rate of the fp32 instruction; same buffers as 'fp32' except the
#            packed weights (three bf16 planes)
# 'bf16x3' : 3-term split-bf16 (two pieces, 16 significand bits) on the same instruction (csrc/mlp_bf16.hip): NARROWER than
#            fp32 (products ~2^-17 relative), ~2.5x the rate of 'fp32'; rendered RGB still within 1e-6 of the fp32 kernels
# (the two-fp16-piece 'f16x3' experiment of round 4 was deleted in round 6: unguarded fp16 range; its record is profiles/r04_f16x3_*)
import os
MATH_MODES = ('fp32', 'bf16x3', 'bf16x6')
_MODE_ID = {'fp32': 0, 'bf16x3': 1, 'bf16x6': 2}
_MATH = os.environ.get('FASTNERF_MATH', 'bf16x6')   # default: the reference's arithmetic width
assert _MATH in MATH_MODES, 'FASTNERF_MATH must be one of ' + ', '.join(MATH_MODES)


def get_math():
    return _MATH


def set_math(mode):
    """Switch the math mode.  Packed weights / saved activations are mode-specific: models re-pack on their
    next packed() call; do not mix buffers produced under different modes."""
    global _MATH
    assert mode in MATH_MODES
    _MATH = mode


def mode_id():
    """math_mode argument of the fused C-ABI entry points (fastnerf_render_rays_*, fastnerf_train_step)."""
    return _MODE_ID[_MATH]


def _x6():
    return _MATH == 'bf16x6'


# math mode a packed-weight buffer was produced under: a Python attribute on the tensor object AND a registry by storage
# address (a tensor that went through the PyTorch dispatcher -- torch.ops.fastnerf.mlp_fwd -- is a fresh object without the
# attribute), so that weights packed under one mode and used under the other are an error, never garbage
_PACK_TAGS = {}


def _tag_packed(t, tag):
    """Tag a packed-weight buffer with its math mode.  The persistent buffers of NeRF.packed() / NerfNet.packed() are re-packed in
    place on every step: ONE finalizer per tensor object (a token the object carries), later calls only update the tag."""
    t._fn_math = tag
    key = (t.device.index, t.data_ptr(), t.numel())
    token = getattr(t, '_fn_token', None)
    entry = _PACK_TAGS.get(key)
    if token is not None and entry is not None and entry[1] is token:
        if entry[0] != tag:
            _PACK_TAGS[key] = (tag, token)
        return
    token = object()
    t._fn_token = token
    _PACK_TAGS[key] = (tag, token)
    # the entry lives as long as the tensor object mlp_pack handed out: a freed block that the caching allocator hands to an
    # unrelated tensor must not inherit the tag
    weakref.finalize(t, _drop_tag, key, token)


def _drop_tag(key, token):
    entry = _PACK_TAGS.get(key)
    if entry is not None and entry[1] is token:
        del _PACK_TAGS[key]


def packed_tag(t):
    tag = getattr(t, '_fn_math', None)
    if tag:
        return tag
    entry = _PACK_TAGS.get((t.device.index, t.data_ptr(), t.numel()))
    return entry[0] if entry else None


def _split(kind):
    return _MATH == 'bf16x3' and int(kind) in (0, 1, 2)


def act_floats(P, kind=0):
    """Size (floats) of the saved-activation buffer for P points of a net of the given kind."""
    if _split(kind):
        return int(lib().fastnerf_mlp_bf16_floats(int(kind), 3, int(P)))
    return int(lib().fastnerf_mlp_act_floats(int(kind), int(P)))


def dact_floats(P, kind=0):
    """Size (floats) of the pre-activation-gradient workspace for P points."""
    if _split(kind):
        return int(lib().fastnerf_mlp_bf16_floats(int(kind), 4, int(P)))
    return int(P) * DACT_FLOATS


def packed_floats(kind, which):
    """which: 1 forward, 2 backward packed-weight buffer (floats) under the current math mode."""
    if _split(kind):
        return int(lib().fastnerf_mlp_bf16_floats(int(kind), int(which), 0))
    if _x6():
        return int(lib().fastnerf_mlp_x6_packed_floats(int(kind), int(which)))
    return net_floats(kind, which)


def net_floats(kind, what=0):
    """what: 0 parameters, 1 packed-forward, 2 packed-backward, 3 padded PE width."""
    return int(lib().fastnerf_net_floats(int(kind), int(what)))


def _f32(t):
    return t.contiguous().float()


def gen_rays(H, W, K, c2w):
    """get_rays (run_nerf_helpers.py:68-78) for every pixel of one camera -> ([H,W,3], [H,W,3])."""
    c2w_t = torch.as_tensor(c2w, dtype=torch.float32)
    dev = c2w_t.device if c2w_t.is_cuda else torch.device('cuda')
    host = np.ascontiguousarray(c2w_t.detach().cpu().numpy()[:3, :4], dtype=np.float32)
    ro = torch.empty(H, W, 3, device=dev, dtype=torch.float32)
    rd = torch.empty(H, W, 3, device=dev, dtype=torch.float32)
    check(lib().fastnerf_gen_rays(H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]),
                                  host.ctypes.data, ptr(ro), ptr(rd), stream()), 'fastnerf_gen_rays')
    return ro, rd


def gen_rays_pixels(pix, poses, K):
    """Rays for selected (image,row,col) pixels; pix [n,3] int32 cuda, poses [n_img,3,4] cuda."""
    require_gpu(pix, poses)
    n = pix.shape[0]
    ro = torch.empty(n, 3, device=pix.device, dtype=torch.float32)
    rd = torch.empty(n, 3, device=pix.device, dtype=torch.float32)
    check(lib().fastnerf_gen_rays_pixels(n, ptr(pix.contiguous().int()), ptr(_f32(poses)), float(K[0][0]),
                                         float(K[1][1]), float(K[0][2]), float(K[1][2]), ptr(ro), ptr(rd), stream()),
          'fastnerf_gen_rays_pixels')
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    require_gpu(rays_o, rays_d)
    sh = rays_o.shape
    ro, rd = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    oo, od = torch.empty_like(ro), torch.empty_like(rd)
    check(lib().fastnerf_ndc_rays(ro.shape[0], H, W, float(focal), float(near), ptr(ro), ptr(rd), ptr(oo), ptr(od),
                                  stream()), 'fastnerf_ndc_rays')
    return oo.reshape(sh), od.reshape(sh)


def pack_rays(rays_o, rays_d, near, far, ndc=False, H=0, W=0, focal=1.0):
    require_gpu(rays_o, rays_d)
    ro, rd = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    out = torch.empty(ro.shape[0], 11, device=ro.device, dtype=torch.float32)
    check(lib().fastnerf_pack_rays(ro.shape[0], ptr(ro), ptr(rd), float(near), float(far), int(bool(ndc)), int(H),
                                   int(W), float(focal), ptr(out), stream()), 'fastnerf_pack_rays')
    return out


def sample_coarse(rays11, S, lindisp=False, perturb=False, t_rand=None, seed=0):
    require_gpu(rays11, t_rand)
    n = rays11.shape[0]
    z = torch.empty(n, S, device=rays11.device, dtype=torch.float32)
    if t_rand is not None:
        t_rand = _f32(t_rand)
        assert t_rand.shape == (n, S)
    check(lib().fastnerf_sample_coarse(n, S, ptr(rays11), int(bool(lindisp)), int(bool(perturb) or t_rand is not None),
                                       ptr(t_rand), int(seed), ptr(z), stream()), 'fastnerf_sample_coarse')
    return z


def posenc(x, L):
    require_gpu(x)
    sh = x.shape
    xf = _f32(x).reshape(-1, 3)
    out = torch.empty(xf.shape[0], 3 + 6 * L, device=x.device, dtype=torch.float32)
    check(lib().fastnerf_posenc(xf.shape[0], L, ptr(xf), ptr(out), stream()), 'fastnerf_posenc')
    return out.reshape(list(sh[:-1]) + [3 + 6 * L])


def mlp_pack(params, packed_fwd=None, packed_bwd=None, kind=0):
    require_gpu(params)
    assert params.numel() == net_floats(kind, 0) and params.is_contiguous()
    if packed_fwd is None:
        packed_fwd = torch.empty(packed_floats(kind, 1), device=params.device, dtype=torch.float32)
    if packed_bwd is None:
        packed_bwd = torch.empty(packed_floats(kind, 2), device=params.device, dtype=torch.float32)
    assert packed_fwd.numel() == packed_floats(kind, 1) and packed_bwd.numel() == packed_floats(kind, 2), \
        'packed buffers were sized under a different math mode'
    # the two modes' buffers can have the same size: tag them so that a mix-up is an error, not garbage
    for t in (packed_fwd, packed_bwd):
        _tag_packed(t, _MATH)
    if _split(kind):
        check(lib().fastnerf_mlp_bf16_pack(int(kind), ptr(params), ptr(packed_fwd), ptr(packed_bwd), stream()),
              'fastnerf_mlp_bf16_pack')
        return packed_fwd, packed_bwd
    if _x6():
        check(lib().fastnerf_mlp_x6_pack(int(kind), ptr(params), ptr(packed_fwd), ptr(packed_bwd), stream()), 'fastnerf_mlp_x6_pack')
        return packed_fwd, packed_bwd
    check(lib().fastnerf_mlp_pack_ex(int(kind), ptr(params), ptr(packed_fwd), ptr(packed_bwd), stream()),
          'fastnerf_mlp_pack_ex')
    return packed_fwd, packed_bwd


def mlp_fwd(rays11, z, params, packed_fwd, act=None, raw=None, kind=0):
    """kind 0/1: points o + d*z; kind 2 (nerf++ background): inverted-sphere points of depth z, consumed
    far->near (raw[:, s] belongs to z[:, S-1-s])."""
    require_gpu(rays11, z, params, packed_fwd)
    n, S = z.shape
    if raw is None:
        raw = torch.empty(n, S, 4, device=z.device, dtype=torch.float32)
    if act is not None:
        assert act.numel() >= act_floats(n * S, kind)
    assert packed_fwd.numel() == packed_floats(kind, 1) and packed_tag(packed_fwd) == _MATH, \
        'packed weights were not produced by mlp_pack under the current math mode'
    if _split(kind):
        check(lib().fastnerf_mlp_bf16_fwd(int(kind), n, S, ptr(rays11), ptr(z), ptr(params), ptr(packed_fwd), ptr(raw),
                                          ptr(act), stream()), 'fastnerf_mlp_bf16_fwd')
        return raw
    if _x6():
        check(lib().fastnerf_mlp_x6_fwd(int(kind), n, S, ptr(rays11), ptr(z), ptr(params), ptr(packed_fwd), ptr(raw), ptr(act), 0,
                                        stream()), 'fastnerf_mlp_x6_fwd')
        return raw
    check(lib().fastnerf_mlp_fwd_ex(int(kind), n, S, ptr(rays11), ptr(z), ptr(params), ptr(packed_fwd), ptr(raw),
                                    ptr(act), stream()), 'fastnerf_mlp_fwd_ex')
    return raw


def mlp_bwd_partial_floats():
    return max(int(lib().fastnerf_mlp_bwd_partial_floats()), int(lib().fastnerf_mlp_bf16_partial_floats()))


def mlp_bwd(draw, act, params, packed_bwd, dact, partial, grads, kind=0):
    require_gpu(draw, act, params, packed_bwd, dact, partial, grads)
    n, S = draw.shape[0], draw.shape[1]
    assert dact.numel() >= dact_floats(n * S, kind) and grads.numel() == net_floats(kind, 0)
    assert packed_bwd.numel() == packed_floats(kind, 2) and packed_tag(packed_bwd) == _MATH, \
        'packed weights were not produced by mlp_pack under the current math mode'
    if _split(kind):
        check(lib().fastnerf_mlp_bf16_bwd(int(kind), n, S, ptr(draw), ptr(act), ptr(params), ptr(packed_bwd), ptr(dact),
                                          ptr(partial), ptr(grads), stream()), 'fastnerf_mlp_bf16_bwd')
        return grads
    if _x6():
        check(lib().fastnerf_mlp_x6_bwd(int(kind), n, S, ptr(draw), ptr(act), ptr(params), ptr(packed_bwd), ptr(dact), ptr(partial),
                                        ptr(grads), stream()), 'fastnerf_mlp_x6_bwd')
        return grads
    check(lib().fastnerf_mlp_bwd_ex(int(kind), n, S, ptr(draw), ptr(act), ptr(params), ptr(packed_bwd), ptr(dact),
                                    ptr(partial), ptr(grads), stream()), 'fastnerf_mlp_bwd_ex')
    return grads


def raw2outputs_fwd(raw, z, rays11, noise=None, white_bkgd=False):
    require_gpu(raw, z, rays11, noise)
    n, S = z.shape
    dev = z.device
    rgb = torch.empty(n, 3, device=dev, dtype=torch.float32)
    disp = torch.empty(n, device=dev, dtype=torch.float32)
    acc = torch.empty(n, device=dev, dtype=torch.float32)
    weights = torch.empty(n, S, device=dev, dtype=torch.float32)
    depth = torch.empty(n, device=dev, dtype=torch.float32)
    check(lib().fastnerf_raw2outputs_fwd(n, S, ptr(raw), ptr(z), ptr(rays11), ptr(noise), int(bool(white_bkgd)),
                                         ptr(rgb), ptr(disp), ptr(acc), ptr(weights), ptr(depth), stream()),
          'fastnerf_raw2outputs_fwd')
    return rgb, disp, acc, weights, depth


def render_rays_fwd(rays11, params_c, packed_c, params_f, packed_f, N_samples, N_importance, lindisp=False, perturb=False,
                    det=True, white_bkgd=False, t_rand=None, u=None, noise0=None, noise1=None, seed0=0, seed1=0, save=False, skip_dead_rgb=False, act_bufs=None):
    """One C-ABI call for the whole forward of render_rays (render.py:238-299).  Returns a dict of the tensors the
    step-by-step ops would have produced (same kernels, same results).  params_f / packed_f may be None when
    N_importance == 0."""
    require_gpu(rays11, params_c, packed_c, params_f, packed_f, t_rand, u, noise0, noise1)
    n = rays11.shape[0]
    dev = rays11.device
    f32 = dict(device=dev, dtype=torch.float32)
    split = _split(0)
    tag = _MATH
    assert packed_tag(packed_c) == tag and (packed_f is None or packed_tag(packed_f) == tag), \
        'packed weights were not produced by mlp_pack under the current math mode'
    if t_rand is not None:
        t_rand = _f32(t_rand)
        assert t_rand.shape == (n, N_samples)
    if u is not None:
        u = _f32(u)
        assert u.shape == (n, N_importance)
    # act_bufs = (buffer for the coarse pass, buffer for the fine pass): caller-owned scratch for the saved activations (11 GB at
    # the bench size) instead of fresh allocations -- for callers whose backward follows before the next forward
    def act_buf(k, count):
        if not save:
            return None
        if act_bufs is not None and act_bufs[k] is not None:
            assert act_bufs[k].numel() >= count and act_bufs[k].dtype == torch.float32 and act_bufs[k].device == dev
            return act_bufs[k]
        return torch.empty(count, **f32)
    o = {'z0': torch.empty(n, N_samples, **f32), 'raw0': torch.empty(n, N_samples, 4, **f32),
         'act0': act_buf(0, act_floats(n * N_samples)),
         'rgb0': torch.empty(n, 3, **f32), 'disp0': torch.empty(n, **f32), 'acc0': torch.empty(n, **f32),
         'w0': torch.empty(n, N_samples, **f32), 'depth0': torch.empty(n, **f32)}
    S1 = N_samples + N_importance
    if N_importance > 0:
        o.update({'z1': torch.empty(n, S1, **f32), 'z_samples': torch.empty(n, N_importance, **f32), 'z_std': torch.empty(n, **f32),
                  'raw1': torch.empty(n, S1, 4, **f32), 'act1': act_buf(1, act_floats(n * S1)),
                  'rgb1': torch.empty(n, 3, **f32), 'disp1': torch.empty(n, **f32), 'acc1': torch.empty(n, **f32),
                  'w1': torch.empty(n, S1, **f32), 'depth1': torch.empty(n, **f32)})
    g = o.get
    # skip_dead_rgb (FN_FWD_SKIP_DEAD_RGB): the inference launches may leave the colour logits of tiles without a live sample at
    # zero -- every other output is bit-identical; only callers that never expose `raw0` / `raw1` ask for it
    check(lib().fastnerf_render_rays_fwd_ex(
        mode_id(), n, int(N_samples), int(N_importance), ptr(rays11), int(bool(lindisp)),
        int(bool(perturb) or t_rand is not None), int(bool(det)), int(bool(white_bkgd)), ptr(t_rand), ptr(u), ptr(noise0), ptr(noise1),
        int(seed0), int(seed1), ptr(params_c), ptr(packed_c), ptr(params_f), ptr(packed_f),
        ptr(o['z0']), ptr(o['raw0']), ptr(o['act0']), ptr(o['rgb0']), ptr(o['disp0']), ptr(o['acc0']), ptr(o['w0']), ptr(o['depth0']),
        ptr(g('z1')), ptr(g('z_samples')), ptr(g('z_std')), ptr(g('raw1')), ptr(g('act1')), ptr(g('rgb1')), ptr(g('disp1')),
        ptr(g('acc1')), ptr(g('w1')), ptr(g('depth1')), 1 if skip_dead_rgb else 0, stream()), 'fastnerf_render_rays_fwd_ex')
    return o


def render_rays_bwd(rays11, white_bkgd, g_rgb, g_rgb0, noise0, noise1, z0, raw0, act0, z1, raw1, act1, params_c, packed_bwd_c,
                    params_f, packed_bwd_f, draw_ws, dact_ws, partial, grads_c, grads_f, N_samples, N_importance):
    """One C-ABI call for the backward of render_rays w.r.t. the parameters of the (distinct) coarse / fine nets."""
    require_gpu(rays11, g_rgb, g_rgb0, z0, raw0, act0, z1, raw1, act1, params_c, packed_bwd_c, params_f, packed_bwd_f, draw_ws,
                dact_ws, partial, grads_c, grads_f)
    n = rays11.shape[0]
    S1 = N_samples + N_importance
    tag = _MATH
    assert packed_tag(packed_bwd_c) == tag and (packed_bwd_f is None or packed_tag(packed_bwd_f) == tag), \
        'packed weights were not produced by mlp_pack under the current math mode'
    assert draw_ws.numel() >= n * S1 * 4 and dact_ws.numel() >= dact_floats(n * S1)
    check(lib().fastnerf_render_rays_bwd(
        mode_id(), n, int(N_samples), int(N_importance), ptr(rays11), int(bool(white_bkgd)), ptr(g_rgb), ptr(g_rgb0),
        ptr(noise0), ptr(noise1), ptr(z0), ptr(raw0), ptr(act0), ptr(z1), ptr(raw1), ptr(act1), ptr(params_c), ptr(packed_bwd_c),
        ptr(params_f), ptr(packed_bwd_f), ptr(draw_ws), ptr(dact_ws), ptr(partial), ptr(grads_c), ptr(grads_f), stream()),
        'fastnerf_render_rays_bwd')


def compact_live(draw):
    """Indices (ascending, int32) of the points whose d(loss)/d(raw) is not exactly zero, and the [live, total] counts --
    both stay on the device.  draw: [n, S, 4] or [P, 4]."""
    require_gpu(draw)
    P = draw.numel() // 4
    idx = torch.empty(P, device=draw.device, dtype=torch.int32)
    cnt = torch.empty(2, device=draw.device, dtype=torch.int32)
    ws = torch.empty(int(lib().fastnerf_compact_ws_ints(P)), device=draw.device, dtype=torch.int32)
    check(lib().fastnerf_compact_live(P, ptr(_f32(draw)), ptr(idx), ptr(cnt), ptr(ws), stream()), 'fastnerf_compact_live')
    return idx, cnt


def mlp_fwd_live(rays11, z, params, packed_fwd, act, live_idx, live_cnt, kind=0):
    """Training forward over a live list: saves the activations of points live_idx[0:live_cnt[0]] (current math mode)."""
    require_gpu(rays11, z, params, packed_fwd, act, live_idx, live_cnt)
    n, S = z.shape
    tag = _MATH
    assert act.numel() >= act_floats(n * S, kind) and live_idx.dtype == torch.int32 and live_cnt.dtype == torch.int32
    assert packed_fwd.numel() == packed_floats(kind, 1) and packed_tag(packed_fwd) == tag
    fn = lib().fastnerf_mlp_bf16_fwd_live if _split(kind) else (lib().fastnerf_mlp_x6_fwd_live if _x6() else lib().fastnerf_mlp_fwd_live_ex)
    check(fn(int(kind), n, S, ptr(rays11), ptr(z), ptr(params), ptr(packed_fwd), ptr(act), ptr(live_idx), ptr(live_cnt), stream()),
          'fastnerf_mlp_fwd_live')


def mlp_bwd_live(draw, act, params, packed_bwd, dact, partial, grads, live_idx, live_cnt, kind=0):
    require_gpu(draw, act, params, packed_bwd, dact, partial, grads, live_idx, live_cnt)
    n, S = draw.shape[0], draw.shape[1]
    tag = _MATH
    assert dact.numel() >= dact_floats(n * S, kind) and grads.numel() == net_floats(kind, 0)
    assert packed_bwd.numel() == packed_floats(kind, 2) and packed_tag(packed_bwd) == tag
    fn = lib().fastnerf_mlp_bf16_bwd_live if _split(kind) else (lib().fastnerf_mlp_x6_bwd_live if _x6() else lib().fastnerf_mlp_bwd_live_ex)
    check(fn(int(kind), n, S, ptr(draw), ptr(act), ptr(params), ptr(packed_bwd), ptr(dact), ptr(partial), ptr(grads), ptr(live_idx),
             ptr(live_cnt), stream()), 'fastnerf_mlp_bwd_live')
    return grads


def live_ws_ints(P):
    return 4 + int(P) + int(lib().fastnerf_compact_ws_ints(int(P)))


def render_rays_bwd_live(rays11, white_bkgd, g_rgb, g_rgb0, noise0, noise1, z0, raw0, z1, raw1, params_c, packed_c,
                         params_f, packed_f, draw_ws, act_ws, dact_ws, partial, live_ws, grads_c, grads_f, N_samples,
                         N_importance, counts=None):
    """One C-ABI call: backward of render_rays with exact zero-gradient point compaction, for a forward that saved
    nothing.  packed_c / packed_f: the (forward, backward) packed-weight pairs.  counts: optional int32[4] device
    tensor receiving (live, total) of the fine and of the coarse pass."""
    require_gpu(rays11, g_rgb, g_rgb0, z0, raw0, z1, raw1, params_c, params_f, draw_ws, act_ws, dact_ws, partial, live_ws,
                grads_c, grads_f, counts)
    n = rays11.shape[0]
    S1 = N_samples + N_importance
    tag = _MATH
    for pk in (packed_c, packed_f):
        assert pk is None or all(packed_tag(t) == tag for t in pk)
    assert draw_ws.numel() >= n * S1 * 4 and dact_ws.numel() >= dact_floats(n * S1) and act_ws.numel() >= act_floats(n * S1)
    assert live_ws.dtype == torch.int32 and live_ws.numel() >= live_ws_ints(n * S1)
    check(lib().fastnerf_render_rays_bwd_live(
        mode_id(), n, int(N_samples), int(N_importance), ptr(rays11), int(bool(white_bkgd)), ptr(g_rgb), ptr(g_rgb0), ptr(noise0), ptr(noise1),
        ptr(z0), ptr(raw0), ptr(z1), ptr(raw1), ptr(params_c), ptr(packed_c[0]), ptr(packed_c[1]),
        ptr(params_f), ptr(None if packed_f is None else packed_f[0]), ptr(None if packed_f is None else packed_f[1]),
        ptr(draw_ws), ptr(act_ws), ptr(dact_ws), ptr(partial), ptr(live_ws), ptr(grads_c), ptr(grads_f), ptr(counts), stream()),
        'fastnerf_render_rays_bwd_live')


def raw2outputs_bwd(raw, z, rays11, g_rgb, noise=None, white_bkgd=False, draw=None):
    require_gpu(raw, z, rays11, g_rgb, noise)
    n, S = z.shape
    if draw is None:
        draw = torch.empty(n, S, 4, device=z.device, dtype=torch.float32)
    check(lib().fastnerf_raw2outputs_bwd(n, S, ptr(raw), ptr(z), ptr(rays11), ptr(noise), int(bool(white_bkgd)),
                                         ptr(_f32(g_rgb)), ptr(draw), stream()), 'fastnerf_raw2outputs_bwd')
    return draw


def sample_pdf_merge(z, weights, Ni, det=False, u=None, seed=0, want_samples=True):
    require_gpu(z, weights, u)
    n, S = z.shape
    dev = z.device
    z_out = torch.empty(n, S + Ni, device=dev, dtype=torch.float32)
    z_samples = torch.empty(n, Ni, device=dev, dtype=torch.float32) if want_samples else None
    z_std = torch.empty(n, device=dev, dtype=torch.float32)
    if u is not None:
        u = _f32(u)
        assert u.shape == (n, Ni)
    check(lib().fastnerf_sample_pdf_merge(n, S, Ni, ptr(z), ptr(weights), int(bool(det)), ptr(u), int(seed),
                                          ptr(z_out), ptr(z_samples), ptr(z_std), stream()),
          'fastnerf_sample_pdf_merge')
    return z_out, z_samples, z_std


def sample_pdf(bins, weights, Ni, det=False, u=None, seed=0):
    require_gpu(bins, weights, u)
    n, M = bins.shape
    out = torch.empty(n, Ni, device=bins.device, dtype=torch.float32)
    if u is not None:
        u = _f32(u)
    check(lib().fastnerf_sample_pdf(n, M, Ni, ptr(_f32(bins)), ptr(_f32(weights)), int(bool(det)), ptr(u), int(seed),
                                    ptr(out), stream()), 'fastnerf_sample_pdf')
    return out


def leaf_table_reset(table):
    """Zero the per-(image, leaf) error table on the device (fastnerf_leaf_table_reset)."""
    require_gpu(table)
    assert table.dtype == torch.int32 and table.is_contiguous()
    check(lib().fastnerf_leaf_table_reset(ptr(table), table.numel(), stream()), 'fastnerf_leaf_table_reset')
    return table


def leaf_table_read(table):
    """The table as host floats (max |gt - pred| per leaf), waited for: what the split rule of tree.py:629-652 consumes."""
    require_gpu(table)
    assert table.dtype == torch.int32 and table.is_contiguous()
    out = torch.empty(table.shape, dtype=torch.float32)
    check(lib().fastnerf_leaf_table_read(ptr(table), out.data_ptr(), table.numel(), stream()), 'fastnerf_leaf_table_read')
    return out


def mse_leafmax(rgb, rgb0, target, grad_scale=1.0, want_grads=True, leaf_tag=None, max_leaves=0, table=None):
    require_gpu(rgb, rgb0, target, leaf_tag, table)
    rgb, target = _f32(rgb), _f32(target)
    rgb0 = None if rgb0 is None else _f32(rgb0)
    leaf_tag = None if leaf_tag is None else leaf_tag.contiguous()
    n = rgb.shape[0]
    dev = rgb.device
    g = torch.empty(n, 3, device=dev, dtype=torch.float32) if want_grads else None
    g0 = torch.empty(n, 3, device=dev, dtype=torch.float32) if (want_grads and rgb0 is not None) else None
    loss2 = torch.empty(2, device=dev, dtype=torch.float32)
    check(lib().fastnerf_mse_leafmax(n, ptr(rgb), ptr(rgb0), ptr(target), float(grad_scale), ptr(g), ptr(g0),
                                     ptr(loss2), ptr(leaf_tag), int(max_leaves), ptr(table), stream()),
          'fastnerf_mse_leafmax')
    return loss2, g, g0


def sigma_noise(n, S0, S1, std, seed, device):
    """(noise0 [n, S0], noise1 [n, S1] or None when S1 == 0): N(0, std^2) sigma noise of both passes of one render_rays call
    (render.py:162), one launch into one allocation (csrc/train.hip gauss_noise_kernel)."""
    tot0 = (n * S0 + 3) // 4 * 4          # (keeps the second view 16-byte aligned)
    buf = torch.empty(tot0 + n * S1, device=device, dtype=torch.float32)
    require_gpu(buf)
    check(lib().fastnerf_gauss_noise(buf.numel(), float(std), int(seed), ptr(buf), stream()), 'fastnerf_gauss_noise')
    return buf[:n * S0].view(n, S0), (buf[tot0:].view(n, S1) if S1 > 0 else None)


def adam_step(params, grads, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    require_gpu(params, grads, m, v)
    check(lib().fastnerf_adam_step(params.numel(), ptr(params), ptr(grads), ptr(m), ptr(v), float(lr), float(beta1),
                                   float(beta2), float(eps), int(step), stream()), 'fastnerf_adam_step')


# ---- nerf++-ours additions -------------------------------------------------------------------
def pp_intersect_sphere(rays11, check_inside=True):
    """ddp_train_nerf.py:54-69.  Raises (like the reference) when a camera is outside the unit sphere;
    check_inside=False skips the device->host read of the counter."""
    require_gpu(rays11)
    n = rays11.shape[0]
    fg_far = torch.empty(n, device=rays11.device, dtype=torch.float32)
    cnt = torch.zeros(1, device=rays11.device, dtype=torch.int32) if check_inside else None
    check(lib().fastnerf_pp_intersect_sphere(n, ptr(rays11), ptr(fg_far), ptr(cnt), stream()),
          'fastnerf_pp_intersect_sphere')
    if check_inside and int(cnt.item()) > 0:
        raise Exception('Not all your cameras are bounded by the unit sphere; please make sure the cameras are '
                        'normalized properly!')
    return fg_far


def pp_fg_depths(fg_far, S, near=1e-4, perturb=True, t_rand=None, seed=0):
    require_gpu(fg_far, t_rand)
    n = fg_far.shape[0]
    z = torch.empty(n, S, device=fg_far.device, dtype=torch.float32)
    if t_rand is not None:
        t_rand = _f32(t_rand)
    check(lib().fastnerf_pp_fg_depths(n, S, float(near), ptr(fg_far), int(bool(perturb) or t_rand is not None),
                                      ptr(t_rand), int(seed), ptr(z), stream()), 'fastnerf_pp_fg_depths')
    return z


def pp_sample_pdf_merge(z, weights, Ni, det=False, u=None, seed=0):
    require_gpu(z, weights, u)
    n, S = z.shape
    z_out = torch.empty(n, S + Ni, device=z.device, dtype=torch.float32)
    z_samples = torch.empty(n, Ni, device=z.device, dtype=torch.float32)
    if u is not None:
        u = _f32(u)
    check(lib().fastnerf_pp_sample_pdf_merge(n, S, Ni, ptr(z), ptr(weights), int(bool(det)), ptr(u), int(seed),
                                             ptr(z_out), ptr(z_samples), stream()), 'fastnerf_pp_sample_pdf_merge')
    return z_out, z_samples


def pp_perturb_samples(z, t_rand=None, seed=0):
    """perturb_samples (ddp_train_nerf.py:72-81) on [n,S] sorted depths."""
    require_gpu(z, t_rand)
    z = _f32(z)
    n, S = z.shape
    out = torch.empty_like(z)
    if t_rand is not None:
        t_rand = _f32(t_rand)
        assert t_rand.shape == z.shape
    check(lib().fastnerf_pp_perturb_samples(n, S, ptr(z), ptr(t_rand), int(seed), ptr(out), stream()), 'fastnerf_pp_perturb_samples')
    return out


def pp_sample_pdf(bins, weights, Ni, det=False, u=None, seed=0):
    """nerf++ sample_pdf (ddp_train_nerf.py:84-133): bins [n,M], weights [n,M-1] -> [n,Ni]."""
    require_gpu(bins, weights, u)
    n, M = bins.shape
    out = torch.empty(n, Ni, device=bins.device, dtype=torch.float32)
    if u is not None:
        u = _f32(u)
    check(lib().fastnerf_pp_sample_pdf(n, M, Ni, ptr(_f32(bins)), ptr(_f32(weights)), int(bool(det)), ptr(u), int(seed), ptr(out),
                                       stream()), 'fastnerf_pp_sample_pdf')
    return out


def pp_depth2pts_outside(ray_o, ray_d, depth):
    """depth2pts_outside (ddp_model.py:16-45): ray_o / ray_d [n,3], depth [n,S] -> (pts [n,S,4], depth_real [n,S])."""
    require_gpu(ray_o, ray_d, depth)
    depth = _f32(depth)
    n, S = depth.shape
    pts = torch.empty(n, S, 4, device=depth.device, dtype=torch.float32)
    dr = torch.empty(n, S, device=depth.device, dtype=torch.float32)
    check(lib().fastnerf_pp_depth2pts_outside(n, S, ptr(_f32(ray_o)), ptr(_f32(ray_d)), ptr(depth), ptr(pts), ptr(dr), stream()),
          'fastnerf_pp_depth2pts_outside')
    return pts, dr


def pp_composite_fwd(part, raw, z, rays11, fg_far=None):
    require_gpu(raw, z, rays11, fg_far)
    n, S = z.shape
    dev = z.device
    rgb = torch.empty(n, 3, device=dev, dtype=torch.float32)
    w = torch.empty(n, S, device=dev, dtype=torch.float32)
    depth = torch.empty(n, device=dev, dtype=torch.float32)
    lam = torch.empty(n, device=dev, dtype=torch.float32) if part == 0 else None
    check(lib().fastnerf_pp_composite_fwd(n, S, int(part), ptr(raw), ptr(z), ptr(rays11), ptr(fg_far), ptr(rgb), ptr(w),
                                          ptr(depth), ptr(lam), stream()), 'fastnerf_pp_composite_fwd')
    return rgb, w, depth, lam


def pp_composite_bwd(part, raw, z, rays11, g_rgb, fg_far=None, g_lambda=None):
    require_gpu(raw, z, rays11, g_rgb, fg_far, g_lambda)
    n, S = z.shape
    draw = torch.empty(n, S, 4, device=z.device, dtype=torch.float32)
    check(lib().fastnerf_pp_composite_bwd(n, S, int(part), ptr(raw), ptr(z), ptr(rays11), ptr(fg_far), ptr(_f32(g_rgb)),
                                          ptr(None if g_lambda is None else _f32(g_lambda)), ptr(draw), stream()),
          'fastnerf_pp_composite_bwd')
    return draw


def leaf_sumcount(rgb, target, leaf_tag, max_leaves, sums, counts):
    """Accumulate per-(image, leaf) fp64 sums of |gt-pred| (rays x channels) and ray counts."""
    require_gpu(rgb, target, leaf_tag, sums, counts)
    assert sums.dtype == torch.float64 and counts.dtype == torch.int32
    check(lib().fastnerf_leaf_sumcount(rgb.shape[0], ptr(_f32(rgb)), ptr(_f32(target)), ptr(leaf_tag), int(max_leaves),
                                       ptr(sums), ptr(counts), stream()), 'fastnerf_leaf_sumcount')


def pp_gen_rays(H, W, intrinsics, c2w, device='cuda'):
    """get_rays_single_image (nerf_sample_ray_split.py:10-34) -> rays_o, rays_d [H*W,3] on the device."""
    K = np.ascontiguousarray(np.asarray(intrinsics, dtype=np.float64).reshape(4, 4))
    M = np.ascontiguousarray(np.asarray(c2w, dtype=np.float64).reshape(4, 4))
    ro = torch.empty(H * W, 3, device=device, dtype=torch.float32)
    rd = torch.empty(H * W, 3, device=device, dtype=torch.float32)
    check(lib().fastnerf_pp_gen_rays(int(H), int(W), K.ctypes.data, M.ctypes.data, ptr(ro), ptr(rd), stream()),
          'fastnerf_pp_gen_rays')
    return ro, rd
