"""nerf++-ours hot path on the HIP kernels: mirror of
nerf++-ours/{nerf_network.py MLPNet, ddp_model.py NerfNet / NerfNetWithAutoExpo,
ddp_train_nerf.py intersect_sphere / perturb_samples / sample_pdf / train_step}.

The foreground MLPNet has exactly the NeRF MLP's shapes; the background MLPNet takes the 84-channel
encoding of the 4-D inverted-sphere point.  Both run through the same fused fp32-MFMA kernels
(`kind` 1 / 2 of fastnerf_mlp_*_ex); sigma = |.| and rgb = sigmoid(.) are applied by the
compositing kernels, as the reference's network does internally."""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import ops
from .render import _Workspace, _burn_seeds, _next_seed

TINY_NUMBER = 1e-6
HUGE_NUMBER = 1e10


def mlpnet_slices(kind):
    """[(name, offset, shape)] in MLPNet.parameters() order (nerf_network.py:70-120)."""
    ic = 84 if kind == 2 else 63
    out, off = [], 0
    dims = [ic] + [256] * 4 + [256 + ic] + [256] * 2
    for i in range(8):
        out.append((f'base_layers.{i}.0.weight', off, (256, dims[i]))); off += 256 * dims[i]
        out.append((f'base_layers.{i}.0.bias', off, (256,))); off += 256
    for name, shp in (('sigma_layers.0', (1, 256)), ('base_remap_layers.0', (256, 256)), ('rgb_layers.0', (128, 283)),
                      ('rgb_layers.2', (3, 128))):
        out.append((name + '.weight', off, shp)); off += shp[0] * shp[1]
        out.append((name + '.bias', off, (shp[0],))); off += shp[0]
    assert off == ops.net_floats(kind, 0)
    return out


class MLPNet(nn.Module):
    """Parameters are views into a flat buffer in parameters() order; names match the reference."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_viewdirs=27, skips=[4], use_viewdirs=True, device='cuda',
                 flat=None, flat_grad=None):
        super().__init__()
        if not (D == 8 and W == 256 and input_ch in (63, 84) and input_ch_viewdirs == 27 and list(skips) == [4]
                and use_viewdirs):
            raise NotImplementedError('HIP MLPNet implements D=8, W=256, input_ch in {63, 84}, viewdirs 27, skips=[4]')
        self.kind = 2 if input_ch == 84 else 1
        self.input_ch, self.input_ch_viewdirs = input_ch, input_ch_viewdirs
        # same module construction order as the reference => same init under torch.manual_seed
        base, dim = [], input_ch
        for i in range(D):
            base.append(nn.Sequential(nn.Linear(dim, W), nn.ReLU()))
            dim = W
            if i in skips and i != D - 1:
                dim += input_ch
        self.base_layers = nn.ModuleList(base)
        self.sigma_layers = nn.Sequential(nn.Linear(dim, 1))
        self.base_remap_layers = nn.Sequential(nn.Linear(dim, 256))
        self.rgb_layers = nn.Sequential(nn.Linear(256 + input_ch_viewdirs, W // 2), nn.ReLU(), nn.Linear(W // 2, 3),
                                        nn.Sigmoid())
        dev = torch.device(device)
        n = ops.net_floats(self.kind, 0)
        self.flat = flat if flat is not None else torch.empty(n, device=dev, dtype=torch.float32)
        self.flat_grad = flat_grad if flat_grad is not None else torch.zeros(n, device=dev, dtype=torch.float32)
        mods = dict(self.named_modules())
        for name, off, shape in mlpnet_slices(self.kind):
            mod_name, leaf = name.rsplit('.', 1)
            mod = mods[mod_name]
            k = int(np.prod(shape))
            view = self.flat[off:off + k].view(shape)
            with torch.no_grad():
                view.copy_(getattr(mod, leaf).detach().to(dev))
            p = nn.Parameter(view)
            p.grad = self.flat_grad[off:off + k].view(shape)
            setattr(mod, leaf, p)
        self._packed = None

    def packed(self, refresh=True):
        if self._packed is None or self._packed_mode != ops.get_math():
            self._packed = (torch.empty(ops.packed_floats(self.kind, 1), device=self.flat.device),
                            torch.empty(ops.packed_floats(self.kind, 2), device=self.flat.device))
            self._packed_mode = ops.get_math()
            refresh = True
        if refresh:
            ops.mlp_pack(self.flat, *self._packed, kind=self.kind)
        return self._packed

    def forward(self, input):
        """Reference signature on already-embedded inputs (nerf_network.py:122-142); convenience
        path only -- NerfNet.forward runs the fused kernels on rays."""
        pts = input[..., :self.input_ch]
        base = self.base_layers[0](pts)
        for i in range(len(self.base_layers) - 1):
            if i == 4:
                base = torch.cat((pts, base), dim=-1)
            base = self.base_layers[i + 1](base)
        sigma = torch.abs(self.sigma_layers(base))
        remap = self.base_remap_layers(base)
        rgb = self.rgb_layers(torch.cat((remap, input[..., -self.input_ch_viewdirs:]), dim=-1))
        return OrderedDict([('rgb', rgb), ('sigma', sigma.squeeze(-1))])


def _nerfnet_forward(net, rays11, fg_far, fg_z, bg_z, save, workspace=False):
    """workspace=True (CascadeTrainer: the backward of a level follows its forward before anything else runs on the stream): the
    saved activations live in the per-(device, stream) scratch of render._Workspace instead of two fresh multi-GB allocations per
    level and step -- inside a long process the caching allocator otherwise ends up freeing and re-allocating them (bench.py's nerf++
    leg measured 26 instead of 19 ms per batch behind the other legs)."""
    dev = rays11.device
    n, Sf = fg_z.shape
    Sb = bg_z.shape[1]
    pf, pb = net.fg_net.packed(), net.bg_net.packed()
    act_f = act_b = None
    if save and workspace:
        act_f = _Workspace.get('pp_act_fg', dev, ops.act_floats(n * Sf, 1))
        act_b = _Workspace.get('pp_act_bg', dev, ops.act_floats(n * Sb, 2))
    elif save:
        act_f = torch.empty(ops.act_floats(n * Sf, 1), device=dev)
        act_b = torch.empty(ops.act_floats(n * Sb, 2), device=dev)
    raw_f = ops.mlp_fwd(rays11, fg_z, net.fg_net.flat, pf[0], act=act_f, kind=1)
    fg_rgb, fg_w, fg_depth, lam = ops.pp_composite_fwd(0, raw_f, fg_z, rays11, fg_far)
    raw_b = ops.mlp_fwd(rays11, bg_z, net.bg_net.flat, pb[0], act=act_b, kind=2)
    bg_rgb, bg_w, bg_depth, _ = ops.pp_composite_fwd(1, raw_b, bg_z, rays11)
    saved = dict(rays11=rays11, fg_far=fg_far, fg_z=fg_z, bg_z=bg_z, raw_f=raw_f, raw_b=raw_b, act_f=act_f, act_b=act_b,
                 lam=lam, bg_rgb=bg_rgb, pf=pf, pb=pb)
    return (fg_rgb, fg_w, fg_depth, lam, bg_rgb, bg_w, bg_depth), saved


def _nerfnet_backward(net, saved, g_rgb, out_f=None, out_b=None):
    """d(loss)/d(params) of fg / bg nets from d(loss)/d(rgb) [n,3] (rgb = fg + lambda * bg)."""
    dev = g_rgb.device
    out_f = out_f if out_f is not None else net.fg_net.flat_grad
    out_b = out_b if out_b is not None else net.bg_net.flat_grad
    lam, bg_rgb = saved['lam'], saved['bg_rgb']
    g_lam = (g_rgb * bg_rgb).sum(-1).contiguous()
    g_bg = (lam[:, None] * g_rgb).contiguous()
    partial = _Workspace.partial(dev)
    n, Sf = saved['fg_z'].shape
    Sb = saved['bg_z'].shape[1]
    draw_f = ops.pp_composite_bwd(0, saved['raw_f'], saved['fg_z'], saved['rays11'], g_rgb, saved['fg_far'], g_lam)
    dact = _Workspace.dact(dev, max(ops.dact_floats(n * Sf, 1), ops.dact_floats(n * Sb, 2)))
    ops.mlp_bwd(draw_f, saved['act_f'], net.fg_net.flat, saved['pf'][1], dact, partial, out_f, kind=1)
    draw_b = ops.pp_composite_bwd(1, saved['raw_b'], saved['bg_z'], saved['rays11'], g_bg)
    ops.mlp_bwd(draw_b, saved['act_b'], net.bg_net.flat, saved['pb'][1], dact, partial, out_b, kind=2)


class _NerfNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, rays11, fg_far, fg_z, bg_z, *params):
        outs, saved = _nerfnet_forward(net, rays11, fg_far, fg_z, bg_z, save=True)
        fg_rgb, fg_w, fg_depth, lam, bg_rgb, bg_w, bg_depth = outs
        rgb = fg_rgb + lam[:, None] * bg_rgb
        ctx.net, ctx.saved, ctx.n_params = net, saved, len(params)
        res = (rgb, fg_w, bg_w, fg_rgb, fg_depth, lam[:, None] * bg_rgb, lam * bg_depth, lam)
        ctx.mark_non_differentiable(*res[1:])
        return res

    @staticmethod
    def backward(ctx, g_rgb, *_):
        net = ctx.net
        out_f, out_b = torch.empty_like(net.fg_net.flat), torch.empty_like(net.bg_net.flat)
        _nerfnet_backward(net, ctx.saved, g_rgb.contiguous(), out_f, out_b)
        grads = [out_f[o:o + int(np.prod(s))].view(s) for _, o, s in mlpnet_slices(1)] + \
                [out_b[o:o + int(np.prod(s))].view(s) for _, o, s in mlpnet_slices(2)]
        assert len(grads) == ctx.n_params
        return (None, None, None, None, None) + tuple(grads)


class NerfNet(nn.Module):
    """ddp_model.py:49-143.  fg and bg parameters share one flat buffer [fg | bg] (1,202,440 floats)."""

    def __init__(self, args=None, device='cuda'):
        super().__init__()
        dev = torch.device(device)
        nf, nb = ops.net_floats(1, 0), ops.net_floats(2, 0)
        self.flat = torch.empty(nf + nb, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(nf + nb, device=dev, dtype=torch.float32)
        self.fg_net = MLPNet(input_ch=63, device=dev, flat=self.flat[:nf], flat_grad=self.flat_grad[:nf])
        self.bg_net = MLPNet(input_ch=84, device=dev, flat=self.flat[nf:], flat_grad=self.flat_grad[nf:])

    def forward(self, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals):
        ops.require_gpu(ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals)
        rays11 = ops.pack_rays(ray_o, ray_d, 0.0, 0.0)
        fg_far, fg_z, bg_z = (t.contiguous().float() for t in (fg_z_max, fg_z_vals, bg_z_vals))
        params = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            res = _NerfNetFn.apply(self, rays11, fg_far, fg_z, bg_z, *params)
        else:
            (fg_rgb, fg_w, fg_depth, lam, bg_rgb, bg_w, bg_depth), _ = _nerfnet_forward(self, rays11, fg_far, fg_z, bg_z,
                                                                                       save=False)
            res = (fg_rgb + lam[:, None] * bg_rgb, fg_w, bg_w, fg_rgb, fg_depth, lam[:, None] * bg_rgb, lam * bg_depth, lam)
        keys = ('rgb', 'fg_weights', 'bg_weights', 'fg_rgb', 'fg_depth', 'bg_rgb', 'bg_depth', 'bg_lambda')
        return OrderedDict(zip(keys, res))


class NerfNetWithAutoExpo(nn.Module):
    """ddp_model.py:157-188 with optim_autoexpo=False (the configs on this path never enable it)."""

    def __init__(self, args=None, optim_autoexpo=False, img_names=None, device='cuda'):
        super().__init__()
        if optim_autoexpo:
            raise NotImplementedError('optim_autoexpo is not used by the tanks-and-temples configs')
        self.nerf_net = NerfNet(args, device=device)

    def forward(self, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals, img_name=None):
        return self.nerf_net(ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals)

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        """Accepts the reference's checkpoints: its nets are saved through nn.DataParallel (`module.` prefix, ddp_train_nerf.py:154).
        Further keywords of nn.Module.load_state_dict (`assign=`) are passed on."""
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, strict=strict, **kwargs)

    def reference_state_dict(self):
        """state_dict under the names the reference saves and strictly loads (`module.nerf_net.fg_net. ...`)."""
        return OrderedDict(('module.' + k, v) for k, v in self.state_dict().items())


def intersect_sphere(ray_o, ray_d):
    """ddp_train_nerf.py:54-69."""
    return ops.pp_intersect_sphere(ops.pack_rays(ray_o, ray_d, 0.0, 0.0))


def perturb_samples(z_vals):
    """ddp_train_nerf.py:72-81: stratified jitter of sorted depths [..., S] inside their mid-point intervals (the uniform
    draws come from a Philox stream keyed from torch's CPU generator, like every device-side draw of this package)."""
    ops.require_gpu(z_vals)
    sh = z_vals.shape
    return ops.pp_perturb_samples(z_vals.reshape(-1, sh[-1]), seed=_next_seed()).reshape(sh)


def sample_pdf(bins, weights, N_samples, det=False):
    """ddp_train_nerf.py:84-133: bins [..., M+1], weights [..., M] -> [..., N_samples] (train_step and render_single_image use
    the fused sampler + sort-merge ops.pp_sample_pdf_merge; this is the stand-alone form for other callers)."""
    ops.require_gpu(bins, weights)
    sh = list(weights.shape[:-1])
    if bins.shape[-1] != weights.shape[-1] + 1:
        raise ValueError('sample_pdf: bins must have one entry more than weights')
    out = ops.pp_sample_pdf(bins.reshape(-1, bins.shape[-1]), weights.reshape(-1, weights.shape[-1]), N_samples, det=det,
                            seed=0 if det else _next_seed())
    return out.reshape(sh + [N_samples])


def depth2pts_outside(ray_o, ray_d, depth):
    """ddp_model.py:16-45: ray_o / ray_d [..., 3], depth [...] (inverse distance to the sphere origin, in [0, 1]) ->
    (pts [..., 4], depth_real [...])."""
    ops.require_gpu(ray_o, ray_d, depth)
    sh = depth.shape
    ro = ray_o.expand(list(sh) + [3]).reshape(-1, 3)
    rd = ray_d.expand(list(sh) + [3]).reshape(-1, 3)
    pts, dr = ops.pp_depth2pts_outside(ro, rd, depth.reshape(-1, 1))
    return pts.reshape(list(sh) + [4]), dr.reshape(sh)


def render_single_image(models, ray_sampler, chunk_size):
    """ddp_test_nerf.py:126-227: one whole image through every cascade level with deterministic depths (uniform
    foreground steps from `min_depth` to the unit sphere, uniform inverse-depth background steps, `sample_pdf(det=True)`
    + sort-merge for the finer levels), in chunks of `chunk_size` rays.  `models` = {'cascade_level', 'cascade_samples',
    'net_<m>'}; `ray_sampler` has H, W and get_all() -> {'ray_o', 'ray_d', 'min_depth', ...} ([H*W, .] tensors).
    Returns one OrderedDict per level with every output of NerfNet.forward except the weights, reshaped to [H, W, -1]
    on the host, like the reference."""
    ray_batch = ray_sampler.get_all()
    dev = models['net_0'].nerf_net.flat.device if hasattr(models['net_0'], 'nerf_net') else models['net_0'].flat.device
    ro_all, rd_all, md_all = (torch.as_tensor(ray_batch[k]).to(dev).float() for k in ('ray_o', 'ray_d', 'min_depth'))
    levels = models['cascade_level']
    merged = [OrderedDict() for _ in range(levels)]
    with torch.no_grad():
        for s in range(0, ro_all.shape[0], chunk_size):
            ray_o, ray_d, fg_near = ro_all[s:s + chunk_size], rd_all[s:s + chunk_size], md_all[s:s + chunk_size]
            fg_far = intersect_sphere(ray_o, ray_d)
            ret = fg_z = bg_z = None
            for m in range(levels):
                net = models['net_{}'.format(m)]
                N = models['cascade_samples'][m]
                if m == 0:
                    step = (fg_far - fg_near) / (N - 1)
                    fg_z = torch.stack([fg_near + i * step for i in range(N)], dim=-1)          # :150-153
                    bg_z = torch.linspace(0., 1., N, device=dev).expand(ray_o.shape[0], N).contiguous()
                else:
                    fg_z, _ = ops.pp_sample_pdf_merge(fg_z, ret['fg_weights'].contiguous(), N, det=True)   # :164-178
                    bg_z, _ = ops.pp_sample_pdf_merge(bg_z, ret['bg_weights'].contiguous(), N, det=True)
                ret = net(ray_o, ray_d, fg_far, fg_z, bg_z)
                for key, val in ret.items():
                    if key not in ('fg_weights', 'bg_weights') and torch.is_tensor(val):
                        merged[m].setdefault(key, []).append(val.cpu())
    for m in range(levels):
        for key in merged[m]:
            merged[m][key] = torch.cat(merged[m][key], dim=0).reshape(ray_sampler.H, ray_sampler.W, -1)
    return merged


class CascadeTrainer:
    """Fused train_step batch of ddp_train_nerf.py:347-404 over `cascade_samples` levels, each level
    with its own net + Adam (no LR decay in this fork), gradients all-reduced when world_size > 1.
    `perturb=False` (an extension: the reference always jitters) takes the unperturbed depths and the deterministic inverse-CDF
    positions, so that a sharded run can be compared with the single-rank run on the union of the shards."""

    def __init__(self, nets, cascade_samples=(64, 128), lrate=5e-4, min_depth=1e-4, perturb=True):
        self.nets = [getattr(n, 'nerf_net', n) for n in nets]
        self.cascade_samples = list(cascade_samples)
        self.lrate, self.min_depth, self.perturb = lrate, min_depth, bool(perturb)
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.m = [torch.zeros_like(n.flat) for n in self.nets]
        self.v = [torch.zeros_like(n.flat) for n in self.nets]
        self.t = [0] * len(self.nets)

    def step(self, ray_o, ray_d, target, rand=None, leaf_tag=None, table=None, max_leaves=0, n_global=None,
             update=True, sumcount=None):
        """One batch through every cascade level.  `table`: the MAX table of nerf-ours' rule (int32 bit patterns);
        `sumcount=(sums fp64, counts int32)`: the nerf++ fork's MEAN rule (nerf++-ours/tree.py:609-632) -- the LAST level's colours
        of this batch are accumulated per (image, leaf) on the device, no host copy.  n_global: rays of the whole batch when this
        rank holds a shard of it (the gradient is scaled to the global-batch mean before the all-reduce); a rank whose shard is
        empty still joins every collective."""
        from . import parallel
        n = ray_o.shape[0]
        dist = parallel.world_size() > 1
        if n == 0 and not dist:        # an empty batch of a single process is a no-op: no Adam step on a zero gradient (moments would decay,
            self.last_depths = []      # t advance and the parameters drift by momentum)
            return torch.zeros(len(self.nets), device=self.nets[0].flat.device), target.new_zeros((0, 3))
        if n == 0:                     # an empty SHARD of a non-empty global batch: join every collective, take the common update
            _burn_seeds(2 * len(self.nets) if self.perturb else 0)   # (the draws of the loop below: two per level)
            for m, net in enumerate(self.nets):
                net.flat_grad.zero_()
                if dist:
                    parallel.all_reduce_sum(net.flat_grad)
                if update:
                    self.t[m] += 1
                    ops.adam_step(net.flat, net.flat_grad, self.m[m], self.v[m], self.lrate, self.t[m])
            self.last_depths = []
            return torch.zeros(len(self.nets), device=self.nets[0].flat.device), target.new_zeros((0, 3))
        rays11 = ops.pack_rays(ray_o, ray_d, 0.0, 0.0)
        fg_far = ops.pp_intersect_sphere(rays11)
        losses, ret = [], None
        fg_z = bg_z = None
        self.last_depths = []      # (fg_z, bg_z) of every level: what a checker needs to re-evaluate the batch
        for m, net in enumerate(self.nets):
            N = self.cascade_samples[m]
            r = rand[m] if rand is not None else {}
            if m == 0:
                fg_z = ops.pp_fg_depths(fg_far, N, self.min_depth, self.perturb, r.get('fg_t'), _next_seed() if self.perturb else 0)
                bg_z = ops.sample_coarse(ops.pack_rays(ray_o, ray_d, 0.0, 1.0), N, perturb=self.perturb, t_rand=r.get('bg_t'),
                                         seed=_next_seed() if self.perturb else 0)
            else:
                det = not self.perturb
                fg_z, _ = ops.pp_sample_pdf_merge(fg_z, ret[1], N, u=r.get('fg_u'), det=det and r.get('fg_u') is None,
                                                  seed=0 if det else _next_seed())
                bg_z, _ = ops.pp_sample_pdf_merge(bg_z, ret[5], N, u=r.get('bg_u'), det=det and r.get('bg_u') is None,
                                                  seed=0 if det else _next_seed())
            self.last_depths.append((fg_z, bg_z))
            outs, saved = _nerfnet_forward(net, rays11, fg_far, fg_z, bg_z, save=True, workspace=True)
            fg_rgb, fg_w, fg_depth, lam, bg_rgb, bg_w, bg_depth = outs
            rgb = fg_rgb + lam[:, None] * bg_rgb
            last = m == len(self.nets) - 1
            scale = 1.0 if n_global is None else float(n) / float(n_global)
            loss2, g, _ = ops.mse_leafmax(rgb, None, target, grad_scale=scale, leaf_tag=leaf_tag if (last and table is not None) else None,
                                          max_leaves=max_leaves, table=table if last else None)
            if last and sumcount is not None:
                ops.leaf_sumcount(rgb, target, leaf_tag, max_leaves, sumcount[0], sumcount[1])
            _nerfnet_backward(net, saved, g)
            if dist:
                parallel.all_reduce_sum(net.flat_grad)
            if update:
                self.t[m] += 1
                ops.adam_step(net.flat, net.flat_grad, self.m[m], self.v[m], self.lrate, self.t[m])
            losses.append(loss2[0])
            ret = outs
        return torch.stack(losses), rgb

    # ---- exchange with torch.optim.Adam (the reference's `optim_<m>` entries of model_*.pth, ddp_train_nerf.py:306-314) ----
    def torch_optimizer_state_dict(self, m):
        """Level m's Adam state in torch.optim.Adam's format over net.parameters() (= the flat buffer's order)."""
        state, off = {}, 0
        for i, p in enumerate(self.nets[m].parameters()):
            k = p.numel()
            state[i] = {'step': torch.tensor(float(self.t[m])), 'exp_avg': self.m[m][off:off + k].view(p.shape).clone(),
                        'exp_avg_sq': self.v[m][off:off + k].view(p.shape).clone()}
            off += k
        assert off == self.nets[m].flat.numel()
        group = {'lr': self.lrate, 'betas': (self.beta1, self.beta2), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(state)))}
        return {'state': state if self.t[m] > 0 else {}, 'param_groups': [group]}

    def load_torch_optimizer(self, m, opt):
        """Take over level m's moments / step count from a torch.optim.Adam (or its state_dict) over the same parameters."""
        sd = opt.state_dict() if hasattr(opt, 'state_dict') else opt
        st = sd['state']
        if len(st) == 0:
            self.m[m].zero_(); self.v[m].zero_(); self.t[m] = 0
            return
        off, steps = 0, set()
        for i, p in enumerate(self.nets[m].parameters()):
            k = p.numel()
            e = st.get(i)
            if e is None:
                self.m[m][off:off + k].zero_(); self.v[m][off:off + k].zero_()
            else:
                self.m[m][off:off + k].copy_(torch.as_tensor(e['exp_avg']).reshape(-1))
                self.v[m][off:off + k].copy_(torch.as_tensor(e['exp_avg_sq']).reshape(-1))
                steps.add(int(float(e['step'])))
            off += k
        assert off == self.nets[m].flat.numel() and len(steps) == 1, 'optimizer state does not match the parameter list'
        self.t[m] = steps.pop()


class QuadTreeManager:
    """nerf++-ours/tree.py fork of the manager: ctor takes ray samplers (objects with H, W, img [H*W,3],
    rays_o / rays_d [H*W,3]); picks are 30%..50% variance-weighted (prob=True, rand=args.randSamp_perc);
    split criterion is the MEAN leaf loss.  Backed by the same native tree as nerf-ours."""

    def __init__(self, ray_samplers, mseThres=0.1, max_depth=5, device='cuda', sharp_imgs=None):
        from .tree import QuadTreeManager as Base
        n = len(ray_samplers)
        H, W = ray_samplers[0].H, ray_samplers[0].W
        images = torch.as_tensor(np.stack([np.asarray(rs.img).reshape(H, W, 3) for rs in ray_samplers], 0))
        poses = torch.eye(4)[None, :3, :4].repeat(n, 1, 1)   # unused: rays are supplied by the samplers
        self._b = Base(H, W, np.eye(3), images, poses, mseThres, max_depth, device=device, criterion='mean',
                       sharp_imgs=sharp_imgs)
        dev = torch.device(device)
        self.origins = torch.stack([torch.as_tensor(np.asarray(rs.rays_o), dtype=torch.float32) for rs in ray_samplers],
                                   0).reshape(n, H, W, 3)
        self.dirs = torch.stack([torch.as_tensor(np.asarray(rs.rays_d), dtype=torch.float32) for rs in ray_samplers],
                                0).reshape(n, H, W, 3)
        self.images = images
        self._dev = dev
        self._dev_data = None

    def __getattr__(self, name):
        return getattr(self.__dict__['_b'], name)

    def __setattr__(self, name, value):
        # the reference's loop assigns manager state directly (treeManager.epoch_size = ..., ddp_train_nerf.py:282)
        if name in ('epoch_size', 'cur_level', 'quadTrees', 'childrens') and '_b' in self.__dict__:
            setattr(self.__dict__['_b'], name, value)
        else:
            object.__setattr__(self, name, value)

    def gen_rays_v3_multiThread(self, down_scale=16, prob=True, rand=0.7, debug=False, last_epoch=False, compat_rng=True):
        """compat_rng=True: the reference's numpy / torch call order (seeded picks identical, host loop over leaves);
        False: the same distribution drawn for all leaves at once on the device."""
        b = self._b
        if prob and b.processor is None:
            from .image_process import ImageProcessor
            b.processor = ImageProcessor([b.images[i].cpu().numpy() for i in range(b.n_images)], scale=0, sharp_imgs=b._sharp_in)
        pix = b.gen_pixels(down_scale, last_epoch, compat_rng, prob=prob, rand=rand)
        b.result_leaf_tag = b._tags_i32.to(self._dev).contiguous() if self._dev.type == 'cuda' else b._tags_i32
        if self._dev_data is None:
            self._dev_data = (self.origins.to(self._dev), self.dirs.to(self._dev), self.images.to(self._dev))
        o, d, im = self._dev_data
        p = pix.to(self._dev)
        i, r, c = p[:, 0], p[:, 1], p[:, 2]
        return o[i, r, c].contiguous(), d[i, r, c].contiguous(), im[i, r, c].contiguous()

    def adjust_tree_multiThread(self, rgb_gt, rgb_pred, thres=0.001, debug=False):
        return self._b.adjust_tree_multiThread(rgb_gt, rgb_pred, thres, debug)


# ---------------------------------------------------------------------------------------------------------------------
# the caller surface of nerf++-ours/ddp_train_nerf.py for config 5: create_nerf (:136-184), train_step (:327-424), the epoch loop
# with the quadtree fork in it (:187-324).  Same names, arguments, return values and checkpoint files (`model_{epoch:04d}.pth`
# holding net_<m> / optim_<m> state_dicts, net keys under DataParallel's `module.` prefix); what runs underneath is NerfNet on
# the HIP kernels through its autograd node and torch.optim.Adam on the flat-buffer parameter views.  (The fused, autograd-free
# engine for the same batch is CascadeTrainer.)
# ---------------------------------------------------------------------------------------------------------------------
def _ckpt_iter(path):
    import os
    stem = os.path.basename(path)[:-4]
    return int(stem[stem.rfind('_') + 1:])


def _is_ckpt_name(fname):
    """`model_<int>.pth`: what this loop and the reference's write (ddp_train_nerf.py:306).  (The reference's path2iter raises on
    any other *.pth in the experiment directory; here a stray file is skipped instead of turning a resume into a crash.)"""
    import re
    return re.fullmatch(r'model_\d+\.pth', fname) is not None


def create_nerf(rank, args, device='cuda'):
    """ddp_train_nerf.py:136-184 -> (start, models): models = OrderedDict(cascade_level, cascade_samples, net_<m>, optim_<m>); the
    newest `*.pth` of basedir/expname (or args.ckpt_path) is reloaded unless args.no_reload.  Networks are initialised under
    torch.manual_seed(777) like the reference, so that every process starts from the same weights."""
    import os
    if getattr(args, 'optim_autoexpo', False):
        raise NotImplementedError('optim_autoexpo is dead code in the reference loop (train_step passes img_name=None)')
    torch.manual_seed(777)
    models = OrderedDict()
    models['cascade_level'] = args.cascade_level
    models['cascade_samples'] = [int(x.strip()) for x in args.cascade_samples.split(',')]
    for m in range(models['cascade_level']):
        net = NerfNetWithAutoExpo(args, optim_autoexpo=False, device=device)
        models['net_{}'.format(m)] = net
        models['optim_{}'.format(m)] = torch.optim.Adam(net.parameters(), lr=args.lrate)
    start = 0
    ckpt_path = getattr(args, 'ckpt_path', None)
    d = os.path.join(args.basedir, args.expname)
    if ckpt_path is not None and os.path.isfile(ckpt_path):
        ckpts = [ckpt_path]
    else:
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if _is_ckpt_name(f)] if os.path.isdir(d) else []
    ckpts = sorted(ckpts, key=_ckpt_iter)
    if len(ckpts) > 0 and not getattr(args, 'no_reload', False):
        start = _ckpt_iter(ckpts[-1])
        to_load = torch.load(ckpts[-1], map_location=device, weights_only=False)
        for m in range(models['cascade_level']):
            for name in ('net_{}'.format(m), 'optim_{}'.format(m)):
                models[name].load_state_dict(to_load[name])
    return start, models


def save_models(models, path):
    """ddp_train_nerf.py:306-314: net_<m> / optim_<m> state_dicts, net keys as the reference's DataParallel wrapper names them."""
    to_save = OrderedDict()
    for m in range(models['cascade_level']):
        to_save['net_{}'.format(m)] = models['net_{}'.format(m)].reference_state_dict()
        to_save['optim_{}'.format(m)] = models['optim_{}'.format(m)].state_dict()
    torch.save(to_save, path)


def train_step(models, rays_o, rays_d, target_rgb, args):
    """ddp_train_nerf.py:327-424: one pass over the epoch's rays in batches of args.batch_size; per batch every cascade level
    draws its depths (level 0: perturbed uniform fg steps up to the unit sphere + perturbed uniform inverse-depth bg steps;
    finer levels: sample_pdf on the previous level's inner weights, merged and sorted), renders, and takes an Adam step on
    img2mse.  Returns the last level's colours of every ray [n, 3] on the host (what adjust_tree_multiThread consumes)."""
    dev = models['net_0'].nerf_net.flat.device
    n_total = rays_o.shape[0]
    collect = []
    train_step.last_log = log = OrderedDict()
    for b0 in range(0, n_total, args.batch_size):
        ray_o = rays_o[b0:b0 + args.batch_size].to(dev).float().contiguous()
        ray_d = rays_d[b0:b0 + args.batch_size].to(dev).float().contiguous()
        rgb_gt = target_rgb[b0:b0 + args.batch_size].to(dev).float()
        fg_far = intersect_sphere(ray_o, ray_d)
        fg_near = 1e-4 * torch.ones_like(ray_d[..., 0])
        ret = fg_depth = bg_depth = None
        for m in range(models['cascade_level']):
            net, optim = models['net_{}'.format(m)], models['optim_{}'.format(m)]
            N = models['cascade_samples'][m]
            if m == 0:
                step = (fg_far - fg_near) / (N - 1)
                fg_depth = perturb_samples(torch.stack([fg_near + i * step for i in range(N)], dim=-1))
                bg_depth = perturb_samples(torch.linspace(0., 1., N, device=dev).expand(ray_d.shape[0], N).contiguous())
            else:
                fg_new = sample_pdf(.5 * (fg_depth[..., 1:] + fg_depth[..., :-1]), ret['fg_weights'].detach()[..., 1:-1], N)
                fg_depth, _ = torch.sort(torch.cat((fg_depth, fg_new), dim=-1))
                bg_new = sample_pdf(.5 * (bg_depth[..., 1:] + bg_depth[..., :-1]), ret['bg_weights'].detach()[..., 1:-1], N)
                bg_depth, _ = torch.sort(torch.cat((bg_depth, bg_new), dim=-1))
            optim.zero_grad()
            ret = net(ray_o, ray_d, fg_far, fg_depth, bg_depth, img_name=None)
            loss = torch.mean((ret['rgb'] - rgb_gt) ** 2)
            loss.backward()
            optim.step()
            log['level_{}/loss'.format(m)] = loss.detach()
        collect.append(ret['rgb'].detach().cpu())
    return torch.cat(collect, 0)


def ddp_train_nerf(args, ray_samplers, log=print, device='cuda', stop_after=None, fused=None):
    """The epoch loop of ddp_train_nerf.py:187-324 on in-memory ray samplers (the directory reader is data_loader_split.py):
    quadtree manager with the MEAN split criterion, per epoch variance-weighted picks (prob=True, rand=args.randSamp_perc;
    the last epoch uniform over every pixel), train_step, subdivision every args.subdivide_every epochs except the last two,
    `model_{epoch:04d}.pth` after every epoch.  Returns (models, treeManager, per-epoch records).

    fused (default: args.fused, else True): the data-parallel engine -- CascadeTrainer on the HIP kernels without autograd, every
    batch sharded rows r::world over the ranks of torch.distributed (one process per GPU), per-level gradient all-reduce, the MEAN
    rule's per-(image, leaf) SUM / COUNT tables accumulated on the device and all-reduced once per subdivide epoch (no host copy
    per batch).  fused=False: the reference-shaped single-process route (train_step above: autograd + torch.optim.Adam)."""
    if fused is None:
        fused = bool(getattr(args, 'fused', True))
    if fused:
        return _ddp_train_fused(args, ray_samplers, log, device, stop_after)
    import os
    os.makedirs(os.path.join(args.basedir, args.expname), exist_ok=True)
    start, models = create_nerf(0, args, device=device)
    tree = QuadTreeManager(ray_samplers, mseThres=0.0, max_depth=args.init_level, device=device)
    records = []
    for epoch_id in range(start + 1, args.n_epoch + 1):
        if stop_after is not None and epoch_id > stop_after:
            break
        last = epoch_id == args.n_epoch
        if last:
            tree.epoch_size = tree.n_images * tree.h * tree.w
            rays_o, rays_d, target = tree.gen_rays_v3_multiThread(down_scale=args.rays_downscale, prob=False, last_epoch=True)
        else:
            rays_o, rays_d, target = tree.gen_rays_v3_multiThread(down_scale=args.rays_downscale, prob=True, rand=args.randSamp_perc,
                                                                  last_epoch=False)
        leaves_before = sum(tree.num_leaves(i) for i in range(tree.n_images))
        pred = train_step(models, rays_o, rays_d, target, args)
        if epoch_id % args.subdivide_every == 0 and epoch_id < args.n_epoch - 1:
            tree.adjust_tree_multiThread(target.cpu(), pred, thres=args.subdivide_thres)
        save_models(models, os.path.join(args.basedir, args.expname, 'model_{:04d}.pth'.format(epoch_id)))
        mse = float(torch.mean((pred - target.cpu()) ** 2))
        records.append({'epoch': epoch_id, 'rays': int(rays_o.shape[0]), 'mse': mse, 'leaves_before': leaves_before,
                        'leaves_after': sum(tree.num_leaves(i) for i in range(tree.n_images)), 'cur_level': tree.cur_level})
        log('epoch {}: {} rays, mse {:.5f}, leaves {} -> {}'.format(epoch_id, rays_o.shape[0], mse, leaves_before,
                                                                    records[-1]['leaves_after']))
    return models, tree, records


def _ddp_train_fused(args, ray_samplers, log, device, stop_after):
    """Config 5 as a data-parallel loop (ddp_train_nerf.py:187-324, tree.py:609-632).  Every rank holds the same
    nets (torch.manual_seed(777) in create_nerf), the same trees and -- seeds broadcast from rank 0 before every epoch's picks --
    the same epoch of rays; of each batch b0 .. b0 + batch_size it renders and back-propagates rows b0 + r :: world.  Exchanges:
    one gradient all-reduce per cascade level and batch; one all-reduce(SUM) of the per-(image, leaf) fp64 error sums and int32
    counts per subdivide epoch (exact: the tables are bit-identical to a single rank's, so every rank splits the same leaves);
    one fp64 scalar per epoch for the logged mse.  Rank 0 writes `model_{epoch:04d}.pth` in the reference's layout."""
    import os
    from . import parallel
    rk, world = parallel.rank(), parallel.world_size()
    d = os.path.join(args.basedir, args.expname)
    if rk == 0:
        os.makedirs(d, exist_ok=True)
    parallel.barrier()
    start, models = create_nerf(rk, args, device=device)
    L = models['cascade_level']
    trainer = CascadeTrainer([models['net_{}'.format(m)] for m in range(L)], models['cascade_samples'], lrate=args.lrate,
                             perturb=bool(getattr(args, 'perturb', 1)))
    for m in range(L):
        trainer.load_torch_optimizer(m, models['optim_{}'.format(m)])   # (a resumed run continues from the file's moments)
    tree = QuadTreeManager(ray_samplers, mseThres=0.0, max_depth=args.init_level, device=device)
    dev = torch.device(device)
    records = []
    for epoch_id in range(start + 1, args.n_epoch + 1):
        if stop_after is not None and epoch_id > stop_after:
            break
        parallel.sync_seed()          # the epoch's pixel picks are drawn redundantly on every rank: same seed, same picks
        last = epoch_id == args.n_epoch
        if last:
            tree.epoch_size = tree.n_images * tree.h * tree.w
            rays_o, rays_d, target = tree.gen_rays_v3_multiThread(down_scale=args.rays_downscale, prob=False, last_epoch=True)
        else:
            rays_o, rays_d, target = tree.gen_rays_v3_multiThread(down_scale=args.rays_downscale, prob=True, rand=args.randSamp_perc,
                                                                  last_epoch=False)
        tags = tree.result_leaf_tag
        ml = tree.max_leaves()
        sums = torch.zeros(tree.n_images * ml, device=dev, dtype=torch.float64)
        counts = torch.zeros(tree.n_images * ml, device=dev, dtype=torch.int32)
        sq = torch.zeros(1, device=dev, dtype=torch.float64)
        n_total = rays_o.shape[0]
        leaves_before = sum(tree.num_leaves(i) for i in range(tree.n_images))
        for b0 in range(0, n_total, args.batch_size):
            b1 = min(b0 + args.batch_size, n_total)
            rows = slice(b0 + rk, b1, world)
            tg = target[rows].float().contiguous()
            _, rgb = trainer.step(rays_o[rows].float().contiguous(), rays_d[rows].float().contiguous(), tg,
                                  leaf_tag=tags[rows].contiguous(), max_leaves=ml, n_global=b1 - b0, sumcount=(sums, counts))
            if rgb.shape[0]:
                sq += ((rgb - tg).double() ** 2).sum()
        if epoch_id % args.subdivide_every == 0 and epoch_id < args.n_epoch - 1:
            parallel.all_reduce_leaf_sumcount(sums, counts)
            tree.adjust_tree_from_sumcount(sums, counts, args.subdivide_thres)
        parallel.all_reduce_sum(sq)
        mse = float(sq.item()) / (3.0 * max(1, n_total))
        if rk == 0:
            for m in range(L):
                models['optim_{}'.format(m)].load_state_dict(trainer.torch_optimizer_state_dict(m))
            save_models(models, os.path.join(d, 'model_{:04d}.pth'.format(epoch_id)))
        parallel.barrier()
        records.append({'epoch': epoch_id, 'rays': int(n_total), 'mse': mse, 'leaves_before': leaves_before,
                        'leaves_after': sum(tree.num_leaves(i) for i in range(tree.n_images)), 'cur_level': tree.cur_level})
        if rk == 0:
            log('epoch {}: {} rays ({} per rank and batch), mse {:.5f}, leaves {} -> {}'.format(
                epoch_id, n_total, -(-args.batch_size // world), mse, leaves_before, records[-1]['leaves_after']))
    ddp_train_nerf.last_trainer = trainer
    return models, tree, records
