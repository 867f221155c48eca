"""PyTorch custom-op registration of the C-ABI entry points (`torch.ops.fastnerf.*`).

The product boundary is the C ABI (include/fastnerf.h, bound with ctypes in _lib.py); this module puts the SAME calls into
PyTorch's operator registry -- schemas, fake-tensor (shape) implementations, and autograd formulas where the reference
differentiates through the op -- so that they are visible to the dispatcher, `torch.compile` tracing and
`torch.library.opcheck` like any other PyTorch-ROCm custom op.  Nothing here computes: every implementation is one call into
ops.py (device pointers + the current HIP stream), and there is no CPU kernel registered -- a CPU tensor raises, as everywhere
in this package.

    torch.ops.fastnerf.gen_rays_pixels(pix, poses, fx, fy, cx, cy)          -> (rays_o, rays_d)   run_nerf_helpers.py:68-78
    torch.ops.fastnerf.pack_rays(rays_o, rays_d, near, far, ndc, H, W, focal) -> rays11             render.py:59-80
    torch.ops.fastnerf.sample_coarse(rays11, S, lindisp, perturb, t_rand, seed) -> z                 render.py:244-266
    torch.ops.fastnerf.posenc(x, L)                                            -> enc                run_nerf_helpers.py:15-63
    torch.ops.fastnerf.mlp_fwd(rays11, z, params, packed_fwd)                  -> raw                model.py:38-63 (+ PE)
    torch.ops.fastnerf.raw2outputs(raw, z, rays11, noise, white_bkgd)          -> (rgb, disp, acc, weights, depth)   render.py:149-192
                                                                                  differentiable w.r.t. raw through rgb
    torch.ops.fastnerf.sample_pdf_merge(z, weights, Ni, det, u, seed)          -> (z_all, z_samples, z_std)          run_nerf_helpers.py:112-155 + render.py:283
    torch.ops.fastnerf.mse_leafmax(rgb, rgb0, target, grad_scale, leaf_tag, max_leaves, table) -> (loss2, g_rgb, g_rgb0)   (table mutated)
    torch.ops.fastnerf.adam_step(params, grads, m, v, lr, step, beta1, beta2, eps)             -> ()  (params, m, v mutated)
    torch.ops.fastnerf.compact_live(draw)                                      -> (live_idx, counts)
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

_lib_def = torch.library.custom_op


@_lib_def('fastnerf::gen_rays_pixels', mutates_args=())
def gen_rays_pixels(pix: Tensor, poses: Tensor, fx: float, fy: float, cx: float, cy: float) -> Tuple[Tensor, Tensor]:
    K = [[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]
    return ops.gen_rays_pixels(pix, poses, K)


@gen_rays_pixels.register_fake
def _(pix, poses, fx, fy, cx, cy):
    return pix.new_empty((pix.shape[0], 3), dtype=torch.float32), pix.new_empty((pix.shape[0], 3), dtype=torch.float32)


@_lib_def('fastnerf::pack_rays', mutates_args=())
def pack_rays(rays_o: Tensor, rays_d: Tensor, near: float, far: float, ndc: bool, H: int, W: int, focal: float) -> Tensor:
    return ops.pack_rays(rays_o, rays_d, near, far, ndc=ndc, H=H, W=W, focal=focal)


@pack_rays.register_fake
def _(rays_o, rays_d, near, far, ndc, H, W, focal):
    return rays_o.new_empty((rays_o.numel() // 3, 11))


@_lib_def('fastnerf::sample_coarse', mutates_args=())
def sample_coarse(rays11: Tensor, S: int, lindisp: bool, perturb: bool, t_rand: Optional[Tensor], seed: int) -> Tensor:
    return ops.sample_coarse(rays11, S, lindisp=lindisp, perturb=perturb, t_rand=t_rand, seed=seed)


@sample_coarse.register_fake
def _(rays11, S, lindisp, perturb, t_rand, seed):
    return rays11.new_empty((rays11.shape[0], S))


@_lib_def('fastnerf::posenc', mutates_args=())
def posenc(x: Tensor, L: int) -> Tensor:
    return ops.posenc(x, L)


@posenc.register_fake
def _(x, L):
    return x.new_empty(tuple(x.shape[:-1]) + (3 + 6 * L,))


@_lib_def('fastnerf::mlp_fwd', mutates_args=())
def mlp_fwd(rays11: Tensor, z: Tensor, params: Tensor, packed_fwd: Tensor) -> Tensor:
    # (the tensor objects the dispatcher hands over carry no Python attributes: ops.mlp_fwd finds the math mode packed_fwd
    # was produced under in ops.mlp_pack's registry by storage address, and raises when it is not the current mode)
    return ops.mlp_fwd(rays11, z, params, packed_fwd)


@mlp_fwd.register_fake
def _(rays11, z, params, packed_fwd):
    return z.new_empty(tuple(z.shape) + (4,))


@_lib_def('fastnerf::raw2outputs', mutates_args=())
def raw2outputs(raw: Tensor, z: Tensor, rays11: Tensor, noise: Optional[Tensor], white_bkgd: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    return ops.raw2outputs_fwd(raw.contiguous(), z.contiguous(), rays11, noise, white_bkgd)


@raw2outputs.register_fake
def _(raw, z, rays11, noise, white_bkgd):
    n, S = z.shape
    return z.new_empty((n, 3)), z.new_empty((n,)), z.new_empty((n,)), z.new_empty((n, S)), z.new_empty((n,))


@_lib_def('fastnerf::raw2outputs_bwd', mutates_args=())
def raw2outputs_bwd(raw: Tensor, z: Tensor, rays11: Tensor, g_rgb: Tensor, noise: Optional[Tensor], white_bkgd: bool) -> Tensor:
    return ops.raw2outputs_bwd(raw.contiguous(), z.contiguous(), rays11, g_rgb.contiguous(), noise, white_bkgd)


@raw2outputs_bwd.register_fake
def _(raw, z, rays11, g_rgb, noise, white_bkgd):
    return raw.new_empty(raw.shape)


def _r2o_setup(ctx, inputs, output):
    raw, z, rays11, noise, white_bkgd = inputs
    ctx.save_for_backward(raw, z, rays11, noise if noise is not None else raw.new_empty(0))
    ctx.has_noise, ctx.white = noise is not None, white_bkgd


def _r2o_backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
    # the training loss reaches the network through the colour map only (run_nerf.py:482-490): that is the implemented formula
    raw, z, rays11, noise = ctx.saved_tensors
    for name, g in (('disp', g_disp), ('acc', g_acc), ('weights', g_w), ('depth', g_depth)):
        if g is not None and bool((g != 0).any()):
            raise NotImplementedError('torch.ops.fastnerf.raw2outputs is differentiable through the colour map only; got a non-zero '
                                      'gradient for ' + name)
    draw = torch.ops.fastnerf.raw2outputs_bwd(raw, z, rays11, g_rgb, noise if ctx.has_noise else None, ctx.white)
    return draw, None, None, None, None


raw2outputs.register_autograd(_r2o_backward, setup_context=_r2o_setup)


@_lib_def('fastnerf::sample_pdf_merge', mutates_args=())
def sample_pdf_merge(z: Tensor, weights: Tensor, Ni: int, det: bool, u: Optional[Tensor], seed: int) -> Tuple[Tensor, Tensor, Tensor]:
    return ops.sample_pdf_merge(z, weights, Ni, det=det, u=u, seed=seed)


@sample_pdf_merge.register_fake
def _(z, weights, Ni, det, u, seed):
    n, S = z.shape
    return z.new_empty((n, S + Ni)), z.new_empty((n, Ni)), z.new_empty((n,))


@_lib_def('fastnerf::mse_leafmax', mutates_args=('table',))
def mse_leafmax(rgb: Tensor, rgb0: Tensor, target: Tensor, grad_scale: float, leaf_tag: Optional[Tensor], max_leaves: int,
                table: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    return ops.mse_leafmax(rgb, rgb0, target, grad_scale=grad_scale, leaf_tag=leaf_tag, max_leaves=max_leaves, table=table)


@mse_leafmax.register_fake
def _(rgb, rgb0, target, grad_scale, leaf_tag, max_leaves, table):
    return rgb.new_empty((2,)), rgb.new_empty(rgb.shape), rgb.new_empty(rgb.shape)


@_lib_def('fastnerf::adam_step', mutates_args=('params', 'm', 'v'))
def adam_step(params: Tensor, grads: Tensor, m: Tensor, v: Tensor, lr: float, step: int, beta1: float, beta2: float, eps: float) -> None:
    ops.adam_step(params, grads, m, v, lr, step, beta1, beta2, eps)


@_lib_def('fastnerf::compact_live', mutates_args=())
def compact_live(draw: Tensor) -> Tuple[Tensor, Tensor]:
    return ops.compact_live(draw)


@compact_live.register_fake
def _(draw):
    return draw.new_empty((draw.numel() // 4,), dtype=torch.int32), draw.new_empty((2,), dtype=torch.int32)


OPS = ['gen_rays_pixels', 'pack_rays', 'sample_coarse', 'posenc', 'mlp_fwd', 'raw2outputs', 'raw2outputs_bwd', 'sample_pdf_merge',
       'mse_leafmax', 'adam_step', 'compact_live']
