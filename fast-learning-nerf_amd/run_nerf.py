"""Training driver surface -- mirror of nerf-ours/run_nerf.py for the hot path.

create_nerf (run_nerf.py:67-153) returns the same 6-tuple and the same render_kwargs keys;
batchify / run_network (:40-64) keep their signatures.  `Trainer` is the fused,
autograd-free optimisation step (render + 2xMSE + backward + all-reduce + Adam + LR decay,
run_nerf.py:479-508) that the benchmark and `train()` use; the autograd-compatible route
(`render(...)`, `loss.backward()`, `optimizer.step()`) stays available through render.py.
Dataset readers / CLI parsing are out of scope; `train()` takes
in-memory images + poses.
"""
import ctypes
import math
import os
import time

import numpy as np
import torch

from . import _lib, ops, parallel
from .model import NeRF, noview_slices
from .render import LivePolicy, _Workspace, _backward_core, _burn_seeds, _forward_core, _next_seed, render, render_path  # noqa: F401
from .run_nerf_helpers import get_embedder, img2mse, mse2psnr
from .tree import QuadTreeManager


def batchify(fn, chunk):
    """run_nerf.py:40-47."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn=None, embeddirs_fn=None, netchunk=1024 * 64):
    """run_nerf.py:50-64: PE + MLP on explicit points [N,S,3] with per-ray viewdirs [N,3].
    Runs the fused HIP forward by expressing every point as a ray with o=pt, d=0."""
    net = getattr(fn, 'module', fn)
    if not isinstance(net, NeRF):
        raise TypeError('run_network needs a fastnerf NeRF module')
    ops.require_gpu(inputs, viewdirs)
    if (viewdirs is not None) != net.use_viewdirs:
        raise ValueError('viewdirs must be given exactly when the network was built with use_viewdirs=True')
    sh = inputs.shape
    pts = inputs.reshape(-1, 3).float()
    P = pts.shape[0]
    rays11 = torch.zeros(P, 11, device=pts.device, dtype=torch.float32)
    rays11[:, 0:3] = pts
    if viewdirs is not None:
        rays11[:, 8:11] = viewdirs[:, None].expand(sh).reshape(-1, 3)
    z = torch.zeros(P, 1, device=pts.device, dtype=torch.float32)
    raw = ops.mlp_fwd(rays11, z, net.flat, net.packed()[0])
    return raw.reshape(list(sh[:-1]) + [4])


class _Args:
    """Defaults of argument_parser.py:4-123 for the flags the hot path reads."""
    netdepth = 8; netwidth = 256; netdepth_fine = 8; netwidth_fine = 256
    N_rand = 32 * 32 * 4; lrate = 5e-4; lrate_decay = 250; chunk = 1024 * 32; netchunk = 1024 * 64
    no_batching = True; no_reload = False; ft_path = None
    N_samples = 64; N_importance = 0; perturb = 1.; use_viewdirs = True; i_embed = 0
    multires = 10; multires_views = 4; raw_noise_std = 0.
    dataset_type = 'blender'; white_bkgd = False; no_ndc = False; lindisp = False
    n_epoch = 12; init_level = 3; rays_downscale = 1; subdivide_every = 1; subdivide_thres = 0.015
    randSamp_perc = 0.5; basedir = './logs'; expname = 'fastnerf'

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def make_args(**kw):
    return _Args(**kw)


def create_nerf(args, device='cuda'):
    """run_nerf.py:67-153 -> (render_kwargs_train, render_kwargs_test, start_epoch, start_iter,
    grad_vars, optimizer).  Both nets live in ONE flat parameter / gradient buffer (coarse first,
    as in `grad_vars`), which is what the fused Trainer and the RCCL all-reduce operate on."""
    if args.multires != 10 or (args.use_viewdirs and args.multires_views != 4) or args.i_embed != 0:
        raise NotImplementedError('HIP path implements multires=10, multires_views=4, i_embed=0')
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    embeddirs_fn, input_ch_views = None, 0                    # run_nerf.py:73-76
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    two = args.N_importance > 0
    n_nets = 2 if two else 1
    dev = torch.device(device)
    output_ch = 5 if two else 4
    N = ops.NET_PARAMS if args.use_viewdirs else noview_slices(output_ch)[1]   # parameters per net
    flat_all = torch.empty(n_nets * N, device=dev, dtype=torch.float32)
    grad_all = torch.zeros(n_nets * N, device=dev, dtype=torch.float32)
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=[4],
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, device=dev, flat=flat_all[:N],
                 flat_grad=grad_all[:N])
    grad_vars = list(model.parameters())
    model_fine = None
    if two:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=[4], input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, device=dev,
                          flat=flat_all[N:], flat_grad=grad_all[N:])
        grad_vars += list(model_fine.parameters())
    network_query_fn = lambda inputs, viewdirs, network_fn: run_network(inputs, viewdirs, network_fn,
                                                                        embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                                                        netchunk=args.netchunk)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start_epoch, start_iter = 0, 0
    # checkpoints: same file layout as run_nerf.py:109-127 / 532-539
    ckpts = []
    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        if os.path.isdir(d):
            ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f]
    create_nerf.last_ckpt_path = None
    if len(ckpts) > 0 and not args.no_reload:
        ckpt = torch.load(ckpts[-1], map_location=dev, weights_only=False)
        create_nerf.last_ckpt_path = ckpts[-1]
        start_epoch, start_iter = ckpt['global_epoch'], ckpt['global_iter']
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    create_nerf.flat_all, create_nerf.grad_all = flat_all, grad_all
    model._flat_all, model._grad_all = flat_all, grad_all
    return render_kwargs_train, render_kwargs_test, start_epoch, start_iter, grad_vars, optimizer


class _StepOut:
    """The tensors of one fused step as a read-only mapping with the keys of render_rays' output dict (plus the extras the
    fused Trainer has always returned); views into the step's output block are built on first access."""

    def __init__(self, block, regions, names):
        self._block, self._regions, self._names, self._cache = block, regions, names, {}

    def __getitem__(self, key):
        t = self._cache.get(key)
        if t is None:
            off, shape = self._regions[self._names[key]]
            n = 1
            for d in shape:
                n *= d
            t = self._cache[key] = self._block[off:off + n].view(shape)
        return t

    def __contains__(self, key):
        return key in self._names

    def get(self, key, default=None):
        return self[key] if key in self._names else default

    def keys(self):
        return self._names.keys()

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)


def _step_regions(n, S0, Ni):
    """name -> (offset, shape) of the per-step tensors inside one fp32 block (every region starts on a 256-byte boundary),
    and the block's size."""
    S1 = S0 + Ni
    shapes = [('rays11', (n, 11)), ('z0', (n, S0)), ('raw0', (n, S0, 4)), ('rgb0', (n, 3)), ('disp0', (n,)), ('acc0', (n,)),
              ('w0', (n, S0)), ('depth0', (n,)), ('g_rgb', (n, 3)), ('loss2', (2,))]
    if Ni > 0:
        shapes += [('z1', (n, S1)), ('z_samples', (n, Ni)), ('z_std', (n,)), ('raw1', (n, S1, 4)), ('rgb1', (n, 3)),
                   ('disp1', (n,)), ('acc1', (n,)), ('w1', (n, S1)), ('depth1', (n,)), ('g_rgb0', (n, 3))]
    regions, off = {}, 0
    for name, shape in shapes:
        cnt = 1
        for d in shape:
            cnt *= d
        regions[name] = (off, shape)
        off += (cnt + 63) // 64 * 64
    return regions, off


_OUT_NAMES_2 = {'rgb_map': 'rgb1', 'disp_map': 'disp1', 'acc_map': 'acc1', 'raw': 'raw1', 'rgb0': 'rgb0', 'disp0': 'disp0',
                'acc0': 'acc0', 'z_std': 'z_std', 'weights': 'w1', 'z_vals': 'z1', 'depth_map': 'depth1',
                'z_samples': 'z_samples', 'weights0': 'w0', 'z0': 'z0'}
_OUT_NAMES_1 = {'rgb_map': 'rgb0', 'disp_map': 'disp0', 'acc_map': 'acc0', 'raw': 'raw0', 'weights': 'w0', 'z_vals': 'z0',
                'depth_map': 'depth0'}


class Trainer:
    """Fused optimisation step over rays sharded across ranks (one process per GPU).

    step(): render (perturbed, hierarchical) -> loss = mse(fine) + mse(coarse) -> analytic backward
    -> RCCL all-reduce(SUM) of the flat gradient (grads are pre-scaled by n_local/N_global, so the
    sum is the gradient of the global-batch mean) -> Adam over both nets in one launch -> LR decay
    with the reference's pre-increment rule (run_nerf.py:498-508)."""

    def __init__(self, render_kwargs_train, H, W, K, near, far, lrate=5e-4, lrate_decay=250, decay=True,
                 beta1=0.9, beta2=0.999, eps=1e-8):
        kw = render_kwargs_train
        self.net_c = kw['network_fn']
        self.net_f = kw['network_fine']
        self.N_samples, self.N_importance = kw['N_samples'], kw['N_importance']
        self.perturb, self.white_bkgd = kw['perturb'], kw['white_bkgd']
        self.raw_noise_std = kw.get('raw_noise_std', 0.)
        self.ndc, self.lindisp = kw.get('ndc', True), kw.get('lindisp', False)
        self.use_viewdirs = self.net_c.use_viewdirs
        self.H, self.W, self.K, self.near, self.far = H, W, K, near, far
        self.flat = self.net_c._flat_all
        self.grad = self.net_c._grad_all
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.lrate, self.lrate_decay, self.decay = lrate, lrate_decay, decay
        self.lr = lrate
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.adam_t = 0
        self.global_iter = 0
        self.world = parallel.world_size()
        self.live = LivePolicy()   # exact zero-gradient point compaction of the backward (render.py)
        # first pass of a compacted step: tiles without a live sample skip their colour branch (FN_FWD_SKIP_DEAD_RGB; the fused
        # step never exposes raw logits, every other output and the gradients are bit-identical -- tests/test_gpu_compact.py)
        self.skip_dead_rgb = True
        self.live_counts = torch.zeros(4, device=self.flat.device, dtype=torch.int32)
        self.last_step_live = False
        # one C-ABI call per step (fastnerf_train_step) instead of ~10 calls and ~25 allocations: two distinct nets with view
        # directions (or a single pass).  FASTNERF_FUSED_STEP=0 keeps the call-by-call sequencing (same kernels, same bits).
        self.fused = (os.environ.get('FASTNERF_FUSED_STEP', '1') != '0' and self.use_viewdirs
                      and (self.N_importance == 0 or (self.net_f is not None and self.net_f is not self.net_c)))
        self.overlap_allreduce = os.environ.get('FASTNERF_OVERLAP_ALLREDUCE', '1') != '0'
        self._sa = None          # fn_step_args of the fused path
        self._sa_key = None
        self.repack()

    def repack(self):
        self.pc = self.net_c.packed(refresh=True)
        self.pf = self.net_f.packed(refresh=True) if self.net_f is not None else None

    def forward_backward(self, rays_o, rays_d, target, leaf_tag=None, table=None, max_leaves=0, t_rand=None, u=None,
                         n_global=None):
        n = rays_o.shape[0]
        dev = rays_o.device
        rays11 = ops.pack_rays(rays_o, rays_d, self.near, self.far, ndc=self.ndc, H=self.H, W=self.W,
                               focal=float(self.K[0][0]))
        if not self.use_viewdirs:
            rays11[:, 8:11] = 0.
        noise0 = noise1 = None
        if self.raw_noise_std > 0.:
            noise0, noise1 = ops.sigma_noise(n, self.N_samples, self.N_samples + self.N_importance, self.raw_noise_std, _next_seed(), dev)
        live = self.live.use_live(self.net_c, self.net_f, self.N_importance)
        out, saved = _forward_core(rays11, self.net_c, self.net_f, self.N_samples, self.N_importance, self.lindisp,
                                   self.perturb, self.white_bkgd, t_rand, u, noise0, noise1, save=not live,
                                   packed_c=self.pc, packed_f=self.pf, skip_dead_rgb=live and self.skip_dead_rgb, act_ws=True)
        scale = 1.0 if n_global is None else float(n) / float(n_global)
        loss2, g, g0 = ops.mse_leafmax(out['rgb_map'], out.get('rgb0'), target, grad_scale=scale, leaf_tag=leaf_tag,
                                       max_leaves=max_leaves, table=table)
        _backward_core(saved, g, g0, counts=self.live_counts if live else None)
        self.net_c.collect_grads()          # (no-ops with view directions: the kernels write the parameters' gradients)
        if self.net_f is not None and self.net_f is not self.net_c:
            self.net_f.collect_grads()
        if live:
            self.live.after_live_step(self.live_counts)
        self.live.tick()
        self.last_step_live = live
        return loss2, out

    # ---- fused route: one fastnerf_train_step call per phase group ------------------------------------------------
    def _fused_prepare(self, rays_o, rays_d, target, leaf_tag, table, max_leaves, t_rand, u, n_global):
        """Fill fn_step_args for this batch; returns (args, out mapping, loss2, live)."""
        n = rays_o.shape[0]
        dev = rays_o.device
        S0, Ni = self.N_samples, self.N_importance
        S1 = S0 + Ni
        P0, P1 = n * S0, n * S1
        ops.require_gpu(rays_o, rays_d, target, leaf_tag, table, t_rand, u)
        tag = ops.get_math()
        if ops.packed_tag(self.pc[0]) != tag:
            self.repack()           # the math mode changed under this trainer
            self._sa_key = None
        live = self.live.use_live(self.net_c, self.net_f, Ni)
        key = (n, str(dev), torch.cuda.current_stream(dev).cuda_stream, tag)
        a = self._sa
        if a is None or self._sa_key != key:
            a = self._sa = _lib.StepArgs()
            self._sa_key = key
            self._regions, self._block_floats = _step_regions(n, S0, Ni)
            a.n, a.net_floats = n, ops.NET_PARAMS
            a.math_mode = ops.mode_id()
            a.N_samples, a.N_importance = S0, Ni
            a.params, a.grads, a.adam_m, a.adam_v = self.flat.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
            a.packed_fwd_c, a.packed_bwd_c = self.pc[0].data_ptr(), self.pc[1].data_ptr()
            if Ni > 0:
                a.packed_fwd_f, a.packed_bwd_f = self.pf[0].data_ptr(), self.pf[1].data_ptr()
            # scratch of this (device, stream), shared with the call-by-call route (render._Workspace)
            ws = _Workspace.dact(dev, ops.dact_floats(P1) + P1 * 4)
            a.dact_ws = ws.data_ptr()
            a.draw_ws = ws.data_ptr() + 4 * ops.dact_floats(P1)
            a.partial_ws = _Workspace.partial(dev).data_ptr()
            a.counts = self.live_counts.data_ptr()
            self._keep = [ws]
        # scalar settings are read from the trainer on EVERY step, like the call-by-call route does (a caller may change
        # trainer.perturb / white_bkgd / near / far between steps): a handful of ctypes stores
        a.lindisp, a.perturb = int(bool(self.lindisp)), int(bool(self.perturb))
        a.white_bkgd, a.ndc, a.H, a.W = int(bool(self.white_bkgd)), int(bool(self.ndc)), int(self.H), int(self.W)
        a.focal, a.near_plane, a.far_plane = float(self.K[0][0]), float(self.near), float(self.far)
        a.beta1, a.beta2, a.eps = self.beta1, self.beta2, self.eps
        # the big buffers of the route this step takes (allocated on first use: a run that never falls back to the plain
        # backward never pays for its 11 GB of saved activations)
        if live:
            if not a.act_ws:
                t1 = _Workspace.get('act', dev, ops.act_floats(P1))
                t2 = _Workspace.get('live', dev, ops.live_ws_ints(P1), torch.int32)
                a.act_ws, a.live_ws = t1.data_ptr(), t2.data_ptr()
                self._keep += [t1, t2]
        elif not a.act0:
            t1 = _Workspace.get('act0', dev, ops.act_floats(P0))
            a.act0 = t1.data_ptr()
            self._keep.append(t1)
            if Ni > 0:
                t2 = _Workspace.get('act1', dev, ops.act_floats(P1))
                a.act1 = t2.data_ptr()
                self._keep.append(t2)
        block = torch.empty(self._block_floats, device=dev, dtype=torch.float32)
        base = block.data_ptr()
        for name, (off, _) in self._regions.items():
            setattr(a, name, base + 4 * off)
        ro, rd, tg = ops._f32(rays_o).reshape(-1, 3), ops._f32(rays_d).reshape(-1, 3), ops._f32(target)
        a.rays_o, a.rays_d, a.target = ro.data_ptr(), rd.data_ptr(), tg.data_ptr()
        hold = [ro, rd, tg]
        if t_rand is not None:
            t_rand = ops._f32(t_rand)
            assert t_rand.shape == (n, S0)
            hold.append(t_rand)
        if u is not None:
            u = ops._f32(u)
            assert u.shape == (n, Ni)
            hold.append(u)
        a.t_rand = None if t_rand is None else t_rand.data_ptr()
        a.u = None if u is None else u.data_ptr()
        noise0 = noise1 = None
        if self.raw_noise_std > 0.:
            noise0, noise1 = ops.sigma_noise(n, S0, S1 if Ni > 0 else 0, self.raw_noise_std, _next_seed(), dev)
            hold += [noise0, noise1]
        a.noise0 = None if noise0 is None else noise0.data_ptr()
        a.noise1 = None if noise1 is None else noise1.data_ptr()
        # (the same draws, in the same order, as the call-by-call route: render._forward_core)
        a.seed0 = _next_seed() if (self.perturb and t_rand is None) else 0
        a.seed1 = _next_seed() if (Ni > 0 and self.perturb and u is None) else 0
        if leaf_tag is not None:
            leaf_tag = leaf_tag.contiguous()
            hold.append(leaf_tag)
        a.leaf_tag = None if leaf_tag is None else leaf_tag.data_ptr()
        a.table = None if table is None else table.data_ptr()
        a.max_leaves = int(max_leaves)
        a.grad_scale = 1.0 if n_global is None else float(n) / float(n_global)
        a.live = int(live)
        a.fwd_flags = 1 if (live and self.skip_dead_rgb and noise0 is None) else 0
        self._hold = hold       # inputs stay referenced until the next step is prepared (the launches are asynchronous)
        out = _StepOut(block, self._regions, _OUT_NAMES_2 if Ni > 0 else _OUT_NAMES_1)
        loss2 = block[self._regions['loss2'][0]:self._regions['loss2'][0] + 2]
        return a, out, loss2, live

    def _fused_call(self, a, phases):
        _lib.check(_lib.lib().fastnerf_train_step(ctypes.byref(a), int(phases), _lib.stream()), 'fastnerf_train_step')

    def _after_backward(self, live):
        if live:
            self.live.after_live_step(self.live_counts)
        self.live.tick()
        self.last_step_live = live

    def step(self, rays_o, rays_d, target, leaf_tag=None, table=None, max_leaves=0, t_rand=None, u=None,
             n_global=None, decay=None):
        n = rays_o.shape[0]
        Nn = ops.NET_PARAMS
        two = self.N_importance > 0 and self.net_f is not None and self.net_f is not self.net_c
        overlap = self.world > 1 and two and self.overlap_allreduce and self.grad.numel() == 2 * Nn
        fused = self.fused and n > 0
        self.adam_t += 1
        if fused:
            a, out, loss2, live = self._fused_prepare(rays_o, rays_d, target, leaf_tag, table, max_leaves, t_rand, u, n_global)
            a.lr, a.adam_t = float(self.lr), int(self.adam_t)
            if self.world == 1:
                self._fused_call(a, _lib.STEP_FORWARD | _lib.STEP_BWD_FINE | _lib.STEP_BWD_COARSE | _lib.STEP_UPDATE)
                self._after_backward(live)
            else:
                # data parallel: the fine net's gradient (the second half of the flat buffer) is final after the fine pass and is
                # all-reduced while the coarse pass's backward runs (the collective has its own stream; `wait` orders ours behind it)
                if overlap:
                    self._fused_call(a, _lib.STEP_FORWARD | _lib.STEP_BWD_FINE)
                    w1 = parallel.all_reduce_sum_async(self.grad[Nn:])
                    self._fused_call(a, _lib.STEP_BWD_COARSE)
                    w0 = parallel.all_reduce_sum_async(self.grad[:Nn])
                    parallel.wait_all(w1, w0)
                else:
                    self._fused_call(a, _lib.STEP_FORWARD | _lib.STEP_BWD_FINE | _lib.STEP_BWD_COARSE)
                    parallel.all_reduce_sum(self.grad)
                self._after_backward(live)
                self._fused_call(a, _lib.STEP_UPDATE)
        else:
            if n == 0:
                # a rank whose shard of a (tail) batch is empty still joins the collective with a zero gradient
                self.grad.zero_()
                _burn_seeds((1 if self.raw_noise_std > 0. else 0) + (1 if (self.perturb and t_rand is None) else 0)
                            + (1 if (self.N_importance > 0 and self.perturb and u is None) else 0))
                loss2, out = torch.zeros(2, device=self.grad.device), {}
            else:
                loss2, out = self.forward_backward(rays_o, rays_d, target, leaf_tag, table, max_leaves, t_rand, u, n_global)
            if self.world > 1:
                if overlap:   # same two collectives as the fused route, so that every rank issues the same sequence
                    parallel.wait_all(parallel.all_reduce_sum_async(self.grad[Nn:]), parallel.all_reduce_sum_async(self.grad[:Nn]))
                else:
                    parallel.all_reduce_sum(self.grad)
            ops.adam_step(self.flat, self.grad, self.m, self.v, self.lr, self.adam_t, self.beta1, self.beta2, self.eps)
            self.repack()
        if (self.decay if decay is None else decay):
            self.lr = self.lrate * (0.1 ** (self.global_iter / (self.lrate_decay * 1000)))
            self.global_iter += 1
        return loss2, out

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'adam_t': self.adam_t, 'global_iter': self.global_iter, 'lr': self.lr}

    # ---- interchange with torch.optim.Adam (the optimizer_state_dict of the reference's .tar, run_nerf.py:121,537) ----
    def _param_list(self):
        ps = list(self.net_c.parameters())
        if self.net_f is not None and self.net_f is not self.net_c:
            ps += list(self.net_f.parameters())
        return ps

    def torch_optimizer_state_dict(self):
        """State in torch.optim.Adam's format over grad_vars (coarse then fine parameters, parameters() order = the
        flat buffer's order): loadable by the reference's `optimizer.load_state_dict`."""
        state, off = {}, 0
        for i, p in enumerate(self._param_list()):
            k = p.numel()
            state[i] = {'step': torch.tensor(float(self.adam_t)),
                        'exp_avg': self.m[off:off + k].view(p.shape).clone(),
                        'exp_avg_sq': self.v[off:off + k].view(p.shape).clone()}
            off += k
        assert off == self.flat.numel()
        group = {'lr': self.lr, 'betas': (self.beta1, self.beta2), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(state)))}
        return {'state': state if self.adam_t > 0 else {}, 'param_groups': [group]}

    def load_torch_optimizer(self, opt):
        """Take over the moments / step count / lr of a torch.optim.Adam (or its state_dict) over the same
        parameters, e.g. the optimizer create_nerf restored from a reference checkpoint."""
        sd = opt.state_dict() if hasattr(opt, 'state_dict') else opt
        st = sd['state']
        self.lr = float(sd['param_groups'][0]['lr'])
        if len(st) == 0:
            self.m.zero_(); self.v.zero_(); self.adam_t = 0
            return
        # a parameter that never received a gradient has NO entry in torch.optim.Adam's state (e.g. the unused views_linears.0 of
        # a reference checkpoint trained without view directions, model.py:60-61): zero moments, step count from the others
        off, steps = 0, set()
        for i, p in enumerate(self._param_list()):
            k = p.numel()
            e = st.get(i)
            if e is None:
                self.m[off:off + k].zero_()
                self.v[off:off + k].zero_()
            else:
                self.m[off:off + k].copy_(torch.as_tensor(e['exp_avg']).reshape(-1))
                self.v[off:off + k].copy_(torch.as_tensor(e['exp_avg_sq']).reshape(-1))
                steps.add(int(float(e['step'])))
            off += k
        assert off == self.flat.numel() and len(steps) == 1, 'optimizer state does not match the parameter list'
        self.adam_t = steps.pop()

    def load_state_dict(self, sd):
        self.m.copy_(sd['m']); self.v.copy_(sd['v'])
        self.adam_t, self.global_iter, self.lr = sd['adam_t'], sd['global_iter'], sd['lr']


def reference_state_dict(net):
    """state_dict with the `module.` prefix of the nn.DataParallel wrapper the reference saves and strictly loads
    (run_nerf.py:82,90,124-126,535-536)."""
    return {'module.' + k: v for k, v in net.state_dict().items()}


def tree_pkl_path(args, epoch):
    """run_nerf.py:339,542."""
    return os.path.join(args.basedir, args.expname, 'treeDivide_{:04d}.pkl'.format(epoch))


def save_checkpoint(args, epoch_id, trainer, kw_train, mgr):
    """run_nerf.py:532-544: `{epoch:03d}.tar` (global_epoch, global_iter, the two DataParallel state_dicts, the Adam
    state in torch.optim's format) + `treeDivide_{epoch:04d}.pkl`.  Both files load in the reference."""
    d = os.path.join(args.basedir, args.expname)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, '{:03d}.tar'.format(epoch_id))
    torch.save({'global_epoch': epoch_id, 'global_iter': trainer.global_iter,
                'network_fn_state_dict': reference_state_dict(kw_train['network_fn']),
                'network_fine_state_dict': (None if kw_train['network_fine'] is None    # N_importance = 0: run_nerf.py:536 fails
                                            else reference_state_dict(kw_train['network_fine'])),   # on None.state_dict(); None is kept
                'optimizer_state_dict': trainer.torch_optimizer_state_dict()}, path)
    mgr.save_trees(tree_pkl_path(args, epoch_id))
    return path


def load_dataset(args):
    """The data-loading branch of the reference's train() (run_nerf.py:160-241) for the two dataset types on this path:
    -> dict(images [n,H,W,3], poses [n,3,4], render_poses, hwf [H,W,focal], K, i_train, i_val, i_test, near, far).
    `args` needs dataset_type, datadir and the matching options (blender: half_res, testskip, white_bkgd; llff: factor,
    spherify, llffhold, no_ndc); render_test swaps the render poses for the test poses like the reference."""
    K = None
    if args.dataset_type == 'llff':
        from .load_llff import load_llff_data
        images, poses, bds, render_poses, i_test = load_llff_data(args.datadir, getattr(args, 'factor', 8), recenter=True,
                                                                  bd_factor=.75, spherify=getattr(args, 'spherify', False))
        hwf = poses[0, :3, -1]
        poses = poses[:, :3, :4]
        i_test = i_test if isinstance(i_test, list) else [i_test]
        if getattr(args, 'llffhold', 8) > 0:
            i_test = np.arange(images.shape[0])[::getattr(args, 'llffhold', 8)]
        i_val = i_test
        i_train = np.array([i for i in np.arange(int(images.shape[0])) if (i not in i_test and i not in i_val)])
        if args.no_ndc:
            near, far = np.ndarray.min(bds) * .9, np.ndarray.max(bds) * 1.
        else:
            near, far = 0., 1.
    elif args.dataset_type == 'blender':
        from .load_blender import load_blender_data
        images, poses, render_poses, hwf, i_split = load_blender_data(args.datadir, getattr(args, 'half_res', False),
                                                                      getattr(args, 'testskip', 8))
        i_train, i_val, i_test = i_split
        near, far = 2., 6.
        if args.white_bkgd:
            images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:])
        else:
            images = images[..., :3]
        poses = poses[:, :3, :4]
    else:
        raise ValueError('Unknown dataset type {!r} (this build reads blender and llff)'.format(args.dataset_type))
    H, W, focal = hwf
    H, W = int(H), int(W)
    if K is None:
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    if getattr(args, 'render_test', False):
        render_poses = np.array(poses[i_test])
    return dict(images=images, poses=poses, render_poses=render_poses, hwf=[H, W, focal], K=K, i_train=i_train, i_val=i_val,
                i_test=i_test, near=near, far=far)


def train(images, poses, H, W, focal, args, near=2., far=6., device='cuda', log=print, max_iters_per_epoch=None,
          compat_rng=False):
    """Epoch loop of run_nerf.py:train() (:337-546) on in-memory data: center-crop warm-up, per-epoch
    quadtree ray generation, fused steps with the on-device leaf-loss table, tree adjustment every
    `subdivide_every` epochs, checkpoints in the reference's file format.  images [n,H,W,3] (CPU or
    GPU tensor), poses [n,3,4]."""
    dev = torch.device(device)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    kw_train, kw_test, start_epoch, global_iter, grad_vars, optimizer = create_nerf(args, device=dev)
    bds_dict = {'near': near, 'far': far}   # run_nerf.py:267-272: the bounds travel in both kwargs dicts
    kw_train.update(bds_dict)
    kw_test.update(bds_dict)
    trainer = Trainer(kw_train, H, W, K, near, far, lrate=args.lrate, lrate_decay=args.lrate_decay)
    trainer.global_iter = global_iter
    if len(optimizer.state_dict()['state']) > 0:   # create_nerf restored a checkpoint (ours or the reference's)
        trainer.load_torch_optimizer(optimizer)
    images = torch.as_tensor(images, dtype=torch.float32)
    poses = torch.as_tensor(poses, dtype=torch.float32)[:, :3, :4]
    mgr = QuadTreeManager(H, W, K, images, poses, mseThres=0.0, max_depth=args.init_level, device=dev)
    # like the model, the subdivided trees of the checkpointed epoch are reloaded (run_nerf.py:338-345)
    if os.path.exists(tree_pkl_path(args, start_epoch)):
        mgr.load_trees(tree_pkl_path(args, start_epoch), cur_level=start_epoch)
        log("load '" + tree_pkl_path(args, start_epoch) + "'")
    N_rand = args.N_rand
    rank, world = parallel.rank(), parallel.world_size()
    history = []

    def run_batches(rays_o, rays_d, tgt, tags, decay, table, max_leaves, local_of=None):
        """local_of = N: the tensors hold only THIS rank's rows of an epoch of N rows (gen_rays_device(shard=...)), batch after batch."""
        n_total = rays_o.shape[0] if local_of is None else local_of
        it = 0
        lo = 0
        for b0 in range(0, n_total, N_rand):
            b1 = min(b0 + N_rand, n_total)
            if local_of is None:
                sl = slice(b0 + rank, b1, world) if world > 1 else slice(b0, b1)
            else:
                cnt = parallel.shard_count(b1 - b0, N_rand, rank, world)
                sl = slice(lo, lo + cnt)
                lo += cnt
            # (a rank's rows r::world of the global batch are a strided view: the kernels take contiguous buffers)
            loss2, _ = trainer.step(rays_o[sl].contiguous(), rays_d[sl].contiguous(), tgt[sl].contiguous(),
                                    leaf_tag=None if tags is None else tags[sl].contiguous(),
                                    table=table, max_leaves=max_leaves, n_global=(b1 - b0) if world > 1 else None,
                                    decay=decay)
            it += 1
            if max_iters_per_epoch is not None and it >= max_iters_per_epoch:
                break
        return loss2, it

    parallel.sync_seed()   # every rank draws the same pixel lists (rows rank::world of ONE global batch)
    if start_epoch == 0:
        # center-crop warm-up (run_nerf.py:367-423): one coordinate set shared by all images, no LR decay
        dH, dW = H // 4, W // 4
        rows = torch.arange(H // 2 - dH, H // 2 + dH)
        cols = torch.arange(W // 2 - dW, W // 2 + dW)
        coords = torch.stack(torch.meshgrid(rows, cols, indexing='ij'), -1).reshape(-1, 2)
        randNum = min(int(N_rand * 500 / mgr.n_images), coords.shape[0])
        sel = coords[np.random.choice(coords.shape[0], size=[randNum], replace=False)]
        pix = torch.cat([torch.cat([torch.full((randNum, 1), i, dtype=torch.int64), sel], 1)
                         for i in range(mgr.n_images)], 0)
        ro, rd, tgt = mgr.gather(pix)
        loss2, it = run_batches(ro, rd, tgt, None, False, None, 0)
        log('warm-up: {} iters, fine/coarse mse {}'.format(it, loss2.tolist()))

    for epoch_id in range(start_epoch + 1, args.n_epoch + 1):
        t0 = time.time()
        parallel.sync_seed()
        last = epoch_id == args.n_epoch
        if last:
            mgr.epoch_size = mgr.n_images * mgr.h * mgr.w
        # data parallel with the device generator: every rank generates ONLY the rows it steps (rows rank :: world of every batch; the seed is
        # the same on every rank after sync_seed, and the row -> ray map is a bijection of the row index: the union is the one epoch)
        sharded_gen = world > 1 and not compat_rng and dev.type == 'cuda'
        ro, rd, tgt = mgr.gen_rays_v3_multiThread(down_scale=1, prob=False, randSamp_proc=args.randSamp_perc,
                                                  last_epoch=last, compat_rng=compat_rng,
                                                  shard=(rank, world, N_rand) if sharded_gen else None)
        tags = mgr.result_leaf_tag
        max_leaves = mgr.max_leaves()
        table = torch.zeros(mgr.n_images * max_leaves, device=dev, dtype=torch.int32)
        loss2, it = run_batches(ro, rd, tgt, tags, True, table, max_leaves, local_of=mgr.epoch_rows if sharded_gen else None)
        psnr = mse2psnr(loss2[:1].cpu())
        history.append((epoch_id, it, float(loss2[0]), float(psnr[0]), time.time() - t0))
        log('epoch {}: {} iters, fine mse {:.5f} psnr {:.2f}, {:.1f}s'.format(*history[-1]) +
            ('' if trainer.live.frac is None else ' (live samples {:.2f}, backward {})'.format(
                trainer.live.frac, 'compacted' if trainer.live.on else 'plain')))
        if args.subdivide_every > 0 and epoch_id % args.subdivide_every == 0 and epoch_id < args.n_epoch - 1:
            if world > 1:
                parallel.all_reduce_max_int(table)
            mgr.adjust_tree_from_table(table.view(mgr.n_images, max_leaves), thres=args.subdivide_thres)
        if rank == 0 and getattr(args, 'save_ckpt', False):
            save_checkpoint(args, epoch_id, trainer, kw_train, mgr)
    return kw_train, kw_test, trainer, mgr, history
