"""Volumetric renderer -- mirror of nerf-ours/render.py on the HIP kernels.

Same call surface: render (render.py:26-91), batchify_rays (:12-24), render_rays (:195-305),
raw2outputs (:149-192), render_path (:94-146).  `render_rays` runs the fused device pipeline

    sample_coarse -> mlp_fwd(coarse) -> raw2outputs -> sample_pdf_merge -> mlp_fwd(fine) -> raw2outputs

and, when autograd is recording, registers ONE autograd node whose backward runs
raw2outputs_bwd + mlp_bwd for both nets and hands the flat gradients to the NeRF parameters,
so the reference's `loss.backward(); optimizer.step()` loop works unchanged.
"""
import os

import numpy as np
import torch

from . import ops
from .model import NeRF, param_slices
from .run_nerf_helpers import compute_ssim, get_rays, to8b

_SEED_GEN = torch.Generator()


def _next_seed():
    """Key of a device-side Philox stream, drawn from torch's global CPU RNG so that torch.manual_seed governs it.  In a sharded run
    every rank draws the SAME value (parallel.sync_seed keeps the host streams in step, which the redundantly drawn pixel picks need) and
    folds its rank in: rows r::world of different ranks must not carry identical jitter patterns.  Rank 0 / a single process: unchanged."""
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    from . import parallel
    rk = parallel.rank()
    if rk:
        seed ^= (rk * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)
    return seed or 1      # (0 means "deterministic" to the samplers: never hand it out as a key)


def _burn_seeds(k):
    """A rank whose shard of a batch is EMPTY skips the step's kernels but must consume the host RNG exactly like the ranks that run it:
    the redundantly drawn pixel picks of the epoch loop rely on every rank's torch CPU stream staying in step between parallel.sync_seed()
    calls."""
    for _ in range(int(k)):
        _next_seed()


def _pytest_rand(shape, device):
    np.random.seed(0)
    return torch.Tensor(np.random.rand(*shape)).to(device)


class _Workspace:
    """Scratch for the backward pass, one set per (device, stream): kernels of one stream run in order, so reuse
    across calls on that stream is safe, and two streams running backward concurrently never share a buffer."""
    _cache = {}

    @classmethod
    def _key(cls, name, device):
        return (name, str(device), torch.cuda.current_stream(device).cuda_stream)

    @classmethod
    def partial(cls, device):
        key = cls._key('partial', device)
        if key not in cls._cache:
            cls._cache[key] = torch.empty(ops.mlp_bwd_partial_floats(), device=device, dtype=torch.float32)
        return cls._cache[key]

    @classmethod
    def get(cls, name, device, count, dtype=torch.float32):
        key = cls._key(name, device)
        t = cls._cache.get(key)
        if t is None or t.numel() < count:
            cls._cache[key] = t = torch.empty(count, device=device, dtype=dtype)
        return t

    @classmethod
    def dact(cls, device, floats):
        return cls.get('dact', device, floats)


# ---- exact zero-gradient point compaction of the training backward (csrc/train.hip, render.cpp) ----
# FASTNERF_COMPACT = auto (default) | 1 | 0.  The compacted backward recomputes the forward of the live points
# (forward work 1 + f, backward work f for a live fraction f), the plain one saves every activation in the first
# forward (1, 1): compaction wins below f ~ 0.76 (split-bf16) / 0.69 (exact fp32).  `auto` starts compacted and follows
# the measured fraction.  Both math modes have the live-list kernels.
_COMPACT = os.environ.get('FASTNERF_COMPACT', 'auto')
assert _COMPACT in ('auto', '0', '1'), 'FASTNERF_COMPACT must be auto, 0 or 1'


def set_compact(mode):
    global _COMPACT
    assert mode in ('auto', '0', '1')
    _COMPACT = mode


def get_compact():
    return _COMPACT


class LivePolicy:
    """Decides per step whether the backward runs compacted; fed with the (live, total) counters of compacted steps
    through pinned-memory copies that are only read once their event has completed (never stalls the stream)."""
    MAX_LAG = 6
    EVERY, PROBE = 4, 256   # (a measurement is a 16-byte asynchronous copy: cheap enough to follow fast changes early in training)
    # break-even live fractions from the measured kernel times (forward without saving + f x (saving forward + backward)
    # against saving forward + backward): 0.76 in the split-bf16 mode, 0.69 in the exact-fp32 mode; hysteresis around them
    THRESHOLDS = {'bf16x3': (0.78, 0.70), 'fp32': (0.70, 0.62), 'bf16x6': (0.70, 0.62)}

    def __init__(self):
        self.frac = None          # last measured live fraction (both passes together)
        self.on = True
        self.step = 0
        self._pending = None      # (pinned host tensor, event)

    def available(self, net_c, net_f, N_importance):
        return N_importance == 0 or (net_f is not None and net_f is not net_c)

    def use_live(self, net_c, net_f, N_importance):
        if _COMPACT == '0' or not self.available(net_c, net_f, N_importance):
            return False
        if _COMPACT == '1':
            return True
        self.poll()
        if self.on:
            return True
        return self.step % self.PROBE == 0     # an occasional compacted step keeps the measurement alive

    def poll(self):
        # The host enqueues steps faster than the GPU runs them, so a measurement can stay "pending" for as many steps as the
        # queue is deep.  Decisions are never taken more than MAX_LAG steps blind: waiting for that event then only holds
        # the HOST back -- the GPU still has MAX_LAG steps of work queued behind it.
        if self._pending is not None and self.step - self._pending[2] >= self.MAX_LAG:
            self._pending[1].synchronize()
        if self._pending is not None and self._pending[1].query():
            c = self._pending[0].tolist()
            self._pending = None
            tot = c[1] + c[3]
            if tot > 0:
                self.frac = (c[0] + c[2]) / tot
                off, on = self.THRESHOLDS[ops.get_math()]
                if self.on and self.frac > off:
                    self.on = False
                elif not self.on and self.frac < on:
                    self.on = True

    def after_live_step(self, counts):
        if self._pending is None and (self.step % self.EVERY == 0 or not self.on):
            host = torch.empty(4, dtype=torch.int32).pin_memory()
            host.copy_(counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (host, ev, self.step)

    def tick(self):
        self.step += 1


def _forward_core(rays11, net_c, net_f, N_samples, N_importance, lindisp, perturb, white_bkgd, t_rand, u, noise0,
                  noise1, save, packed_c=None, packed_f=None, skip_dead_rgb=False, act_ws=False):
    """The fused forward.  Returns (outputs dict, saved-for-backward dict)."""
    pc = packed_c if packed_c is not None else net_c.packed()
    fine = pf = None
    if N_importance > 0:
        fine = net_f if net_f is not None else net_c
        pf = packed_f if packed_f is not None else (fine.packed() if fine is not net_c else pc)
    # act_ws (the fused Trainer: its backward runs before its next forward): the saved activations live in this stream's
    # persistent scratch instead of 11 GB of fresh allocations per step
    act_bufs = None
    if save and act_ws:
        n_ = rays11.shape[0]
        act_bufs = (_Workspace.get('act0', rays11.device, ops.act_floats(n_ * N_samples)),
                    _Workspace.get('act1', rays11.device, ops.act_floats(n_ * (N_samples + N_importance))) if N_importance > 0 else None)
    # one C-ABI call enqueues sampler -> MLP -> compositing [-> sample_pdf + merge -> MLP -> compositing]
    o = ops.render_rays_fwd(rays11, net_c.flat, pc[0], None if fine is None else fine.flat, None if pf is None else pf[0],
                            N_samples, N_importance, lindisp=lindisp, perturb=perturb, det=(perturb == 0.),
                            white_bkgd=white_bkgd, t_rand=t_rand, u=u, noise0=noise0, noise1=noise1,
                            seed0=_next_seed() if (perturb and t_rand is None) else 0,
                            seed1=_next_seed() if (N_importance > 0 and perturb and u is None) else 0, save=save,
                            skip_dead_rgb=bool(skip_dead_rgb and not save and net_c.use_viewdirs), act_bufs=act_bufs)
    out = {}
    saved = {'rays11': rays11, 'z0': o['z0'], 'raw0': o['raw0'], 'act0': o['act0'], 'noise0': noise0, 'white': white_bkgd,
             'net_c': net_c, 'net_f': None, 'pc': pc, 'live': not save}
    if N_importance > 0:
        out.update(rgb_map=o['rgb1'], disp_map=o['disp1'], acc_map=o['acc1'], raw=o['raw1'], rgb0=o['rgb0'], disp0=o['disp0'],
                   acc0=o['acc0'], z_std=o['z_std'], weights=o['w1'], z_vals=o['z1'], depth_map=o['depth1'],
                   z_samples=o['z_samples'], weights0=o['w0'], z0=o['z0'])
        saved.update(z1=o['z1'], raw1=o['raw1'], act1=o['act1'], noise1=noise1, net_f=fine, pf=pf)
    else:
        out.update(rgb_map=o['rgb0'], disp_map=o['disp0'], acc_map=o['acc0'], raw=o['raw0'], weights=o['w0'], z_vals=o['z0'],
                   depth_map=o['depth0'])
    return out, saved


def _backward_core(saved, g_rgb, g_rgb0, out_c=None, out_f=None, counts=None):
    """Writes d(loss)/d(params) of the coarse (and fine) net, given d(loss)/d(rgb maps), into
    out_c / out_f (flat, parameter order; default: the nets' flat_grad buffers).  A forward that saved no activations
    (saved['live']) is followed by the compacted backward; `counts` (int32[4], device) then receives the live / total
    point counts of the fine and the coarse pass."""
    rays11 = saved['rays11']
    dev = rays11.device
    partial = _Workspace.partial(dev)
    net_c, net_f = saved['net_c'], saved['net_f']
    out_c = out_c if out_c is not None else net_c.flat_grad
    if net_f is not None and net_f is not net_c:
        out_f = out_f if out_f is not None else net_f.flat_grad
    n, S0 = saved['z0'].shape
    if saved.get('live'):
        assert net_f is None or net_f is not net_c, 'the compacted backward needs two distinct nets (or one pass)'
        Ni = 0 if net_f is None else saved['z1'].shape[1] - S0
        P = n * (S0 + Ni)
        g_a = g_rgb if g_rgb is not None else torch.zeros(n, 3, device=dev)
        g_b = (g_rgb0 if g_rgb0 is not None else torch.zeros(n, 3, device=dev)) if Ni > 0 else None
        ws = _Workspace.dact(dev, ops.dact_floats(P) + P * 4)
        draw_ws = ws[ops.dact_floats(P):]
        act_ws = _Workspace.get('act', dev, ops.act_floats(P))
        live_ws = _Workspace.get('live', dev, ops.live_ws_ints(P), torch.int32)
        ops.render_rays_bwd_live(rays11, saved['white'], g_a, g_b, saved['noise0'], saved.get('noise1'), saved['z0'], saved['raw0'],
                                 saved.get('z1'), saved.get('raw1'), net_c.flat, saved['pc'], None if Ni == 0 else net_f.flat,
                                 None if Ni == 0 else saved['pf'], draw_ws, act_ws, ws, partial, live_ws, out_c,
                                 out_f if Ni > 0 else None, S0, Ni, counts=counts)
        return
    if net_f is None or net_f is not net_c:
        # one C-ABI call: compositing backward + MLP backward for the fine and the coarse pass
        Ni = 0 if net_f is None else saved['z1'].shape[1] - S0
        g_a = g_rgb if g_rgb is not None else torch.zeros(n, 3, device=dev)
        g_b = (g_rgb0 if g_rgb0 is not None else torch.zeros(n, 3, device=dev)) if Ni > 0 else None
        ws = _Workspace.dact(dev, ops.dact_floats(n * (S0 + Ni)) + n * (S0 + Ni) * 4)
        draw_ws = ws[ops.dact_floats(n * (S0 + Ni)):]
        ops.render_rays_bwd(rays11, saved['white'], g_a, g_b, saved['noise0'], saved.get('noise1'), saved['z0'], saved['raw0'],
                            saved['act0'], saved.get('z1'), saved.get('raw1'), saved.get('act1'), net_c.flat, saved['pc'][1],
                            None if Ni == 0 else net_f.flat, None if Ni == 0 else saved['pf'][1], draw_ws, ws, partial, out_c,
                            out_f if Ni > 0 else None, S0, Ni)
        return
    # one shared network for both passes (N_importance > 0 without network_fine): accumulate the two gradients
    n, S1 = saved['z1'].shape
    if g_rgb is None:
        g_rgb = torch.zeros(n, 3, device=dev)
    draw1 = ops.raw2outputs_bwd(saved['raw1'], saved['z1'], rays11, g_rgb, saved['noise1'], saved['white'])
    dact = _Workspace.dact(dev, ops.dact_floats(n * S1))
    gtmp = torch.empty_like(out_c)
    ops.mlp_bwd(draw1, saved['act1'], net_f.flat, saved['pf'][1], dact, partial, gtmp)
    g_c = g_rgb0 if g_rgb0 is not None else torch.zeros(n, 3, device=dev)
    draw0 = ops.raw2outputs_bwd(saved['raw0'], saved['z0'], rays11, g_c, saved['noise0'], saved['white'])
    dact = _Workspace.dact(dev, ops.dact_floats(n * S0))
    ops.mlp_bwd(draw0, saved['act0'], net_c.flat, saved['pc'][1], dact, partial, out_c)
    out_c.add_(gtmp)


_POLICY = LivePolicy()   # the autograd route's policy (the fused Trainer keeps its own)


def _grad_views(flat):
    return [flat[off:off + int(np.prod(shape))].view(shape) for _, off, shape in param_slices()]


class _RenderRaysFn(torch.autograd.Function):
    """One autograd node for the whole fused render_rays; inputs are the NeRF parameters so that
    autograd routes the flat gradients to them."""

    @staticmethod
    def forward(ctx, cfg, *params):
        live = _POLICY.use_live(cfg['net_c'], cfg['net_f'], cfg['N_importance'])
        out, saved = _forward_core(save=not live, **cfg)
        ctx.saved = saved
        ctx.n_params = len(params)
        keys = ['rgb_map', 'disp_map', 'acc_map', 'raw'] + (['rgb0', 'disp0', 'acc0', 'z_std'] if 'rgb0' in out else [])
        ctx.keys = keys
        outs = tuple(out[k] for k in keys)
        ctx.mark_non_differentiable(*[o for k, o in zip(keys, outs) if k not in ('rgb_map', 'rgb0')])
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        g = dict(zip(ctx.keys, gouts))
        saved = ctx.saved
        # fresh buffers: autograd accumulates the returned tensors into the parameters' .grad
        # (which may alias the nets' flat_grad), so the kernels must not write there directly
        two = saved['net_f'] is not None and saved['net_f'] is not saved['net_c']
        out_c = torch.empty_like(saved['net_c'].flat)
        out_f = torch.empty_like(saved['net_f'].flat) if two else None
        counts = _Workspace.get('counts', out_c.device, 4, torch.int32) if saved.get('live') else None
        _backward_core(saved, g.get('rgb_map'), g.get('rgb0'), out_c, out_f, counts=counts)
        if counts is not None:
            _POLICY.after_live_step(counts)
        _POLICY.tick()
        grads = saved['net_c'].param_grads_from(out_c) + (saved['net_f'].param_grads_from(out_f) if two else [])
        assert len(grads) == ctx.n_params
        return (None,) + tuple(grads)


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """render.py:149-192 -> (rgb_map, disp_map, acc_map, weights, depth_map).  Forward only."""
    n, S = z_vals.shape
    rays11 = torch.zeros(n, 11, device=z_vals.device, dtype=torch.float32)
    rays11[:, 3:6] = rays_d
    noise = None
    if raw_noise_std > 0.:
        noise = torch.randn(n, S, device=z_vals.device) * raw_noise_std
        if pytest:
            noise = _pytest_rand((n, S), z_vals.device) * raw_noise_std
    return ops.raw2outputs_fwd(raw.contiguous().float(), z_vals.contiguous().float(), rays11, noise, white_bkgd)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False):
    """render.py:195-305.  `network_query_fn` is accepted for signature compatibility; positional
    encoding + MLP run fused inside the HIP kernels, so `network_fn`/`network_fine` must be
    fastnerf NeRF modules (anything else raises -- there is no fallback path)."""
    net_c = getattr(network_fn, 'module', network_fn)
    net_f = getattr(network_fine, 'module', network_fine) if network_fine is not None else None
    if not isinstance(net_c, NeRF) or (net_f is not None and not isinstance(net_f, NeRF)):
        raise TypeError('render_rays needs fastnerf NeRF modules (the HIP path has no generic fallback)')
    if ray_batch.shape[-1] not in (8, 11):
        raise ValueError('ray batches are [N,8] (o, d, near, far) or [N,11] (+ view directions), render.py:216-219')
    if ray_batch.shape[-1] == 11 and not net_c.use_viewdirs:
        raise ValueError('a ray batch with view directions needs networks built with use_viewdirs=True')
    if ray_batch.shape[-1] == 8 and net_c.use_viewdirs:
        raise ValueError('networks built with use_viewdirs=True need ray batches with view directions [N,11]')
    ops.require_gpu(ray_batch)
    rays11 = ray_batch.contiguous().float()
    if rays11.shape[-1] == 8:      # no view directions: the kernels' direction slots stay zero (their weights are zero too)
        rays11 = torch.cat([rays11, torch.zeros(rays11.shape[0], 3, device=rays11.device)], -1)
    n = rays11.shape[0]
    dev = rays11.device
    t_rand = u = noise0 = noise1 = None
    if perturb > 0. and pytest:
        t_rand = _pytest_rand((n, N_samples), dev)
        if N_importance > 0:
            u = _pytest_rand((n, N_importance), dev)
    if raw_noise_std > 0.:
        if pytest:
            noise0 = _pytest_rand((n, N_samples), dev) * raw_noise_std
            noise1 = _pytest_rand((n, N_samples + N_importance), dev) * raw_noise_std
        else:      # one Philox launch for both passes (the injected-tensor path above stays for pytest=True)
            noise0, noise1 = ops.sigma_noise(n, N_samples, N_samples + N_importance if N_importance > 0 else 0, raw_noise_std, _next_seed(), dev)
    cfg = dict(rays11=rays11, net_c=net_c, net_f=net_f, N_samples=N_samples, N_importance=N_importance,
               lindisp=lindisp, perturb=perturb, white_bkgd=white_bkgd, t_rand=t_rand, u=u, noise0=noise0,
               noise1=noise1,
               # nobody sees the colour logits of this call: tiles without a live sample may skip them (FN_FWD_SKIP_DEAD_RGB)
               skip_dead_rgb=not retraw)
    params = list(net_c.parameters()) + (list(net_f.parameters()) if (net_f is not None and net_f is not net_c) else [])
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        outs = _RenderRaysFn.apply(cfg, *params)
        keys = ['rgb_map', 'disp_map', 'acc_map', 'raw'] + (['rgb0', 'disp0', 'acc0', 'z_std'] if N_importance > 0 else [])
        out = dict(zip(keys, outs))
    else:
        out, _ = _forward_core(save=False, **cfg)
    ret = {'rgb_map': out['rgb_map'], 'disp_map': out['disp_map'], 'acc_map': out['acc_map']}
    if retraw:
        ret['raw'] = out['raw']
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'], ret['z_std'] = out['rgb0'], out['disp0'], out['acc0'], out['z_std']
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """render.py:12-24."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """render.py:26-91 -> [rgb_map, disp_map, acc_map, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    ops.require_gpu(rays_o, rays_d)
    view_o, view_d = rays_o, rays_d
    if use_viewdirs and c2w_staticcam is not None:   # (render.py:59-66: the static camera only exists inside this branch)
        rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
    sh = rays_d.shape
    rays11 = ops.pack_rays(rays_o, rays_d, near, far, ndc=ndc, H=H, W=W, focal=float(K[0][0]))
    if not use_viewdirs:
        rays11 = rays11[:, :8].contiguous()           # [N,8]: o, d, near, far (render.py:74-78)
    elif c2w_staticcam is not None:
        # viewdirs come from the moving camera, origins/directions from the static one
        rays11[:, 8:11] = ops.pack_rays(view_o, view_d, near, far)[:, 8:11]
    all_ret = batchify_rays(rays11, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """render.py:94-146 without the LPIPS dependency (external, optional in the reference env):
    renders every pose, reports PSNR and SSIM when ground truth is given, writes PNGs when imageio exists."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps, psnrs, ssims = [], [], [], []
    with torch.no_grad():
        for i, c2w in enumerate(render_poses):
            rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=torch.as_tensor(c2w)[:3, :4], **render_kwargs)
            rgbs.append(rgb.cpu().numpy())
            disps.append(disp.cpu().numpy())
            if gt_imgs is not None and render_factor == 0:
                gt = gt_imgs[i].cpu().numpy() if torch.is_tensor(gt_imgs[i]) else np.asarray(gt_imgs[i])
                psnrs.append(-10. * np.log10(np.mean(np.square(rgbs[-1] - gt))))
                ssims.append(float(compute_ssim(torch.as_tensor(gt).float(), rgb.cpu())))
            if savedir is not None:
                try:
                    import imageio
                    imageio.imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
                except ImportError:
                    np.save(os.path.join(savedir, '{:03d}.npy'.format(i)), to8b(rgbs[-1]))
    if psnrs and savedir is not None:
        with open(os.path.join(savedir, 'results.txt'), 'w') as f:
            f.write('mean PSNR: {}\nmean SSIM: {}\n'.format(np.mean(psnrs), np.mean(ssims)))
    render_path.last_psnrs = psnrs
    render_path.last_ssims = ssims
    return np.stack(rgbs, 0), np.stack(disps, 0)
