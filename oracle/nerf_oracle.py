"""CPU oracle for the NeRF training inner loop -- TEST INFRASTRUCTURE ONLY.

This module is a from-scratch CPU (PyTorch fp32, CPU tensors) restatement of the
algorithm of the reference hot path (nerf-ours).  It exists only so that
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg can check /
time the HIP path against it.  Nothing in the product package
(`fast-learning-nerf_amd/`) imports it; the product path has no CPU fallback.

Parity status: PINNED.  `oracle/make_golden.py` imports the reference itself
(from /root/reference, build container only) and records golden vectors under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function here against
those vectors (the reference has no tests of its own for this path).

Each function cites the reference lines (relative to /root/reference/nerf-ours/)
whose arithmetic it restates.  Operation ORDER follows the reference where fp32
rounding is order dependent (e.g. `near*(1-t)+far*t`, `lower+(upper-lower)*u`).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

F32 = torch.float32


# --------------------------------------------------------------------------
# cameras / rays
# --------------------------------------------------------------------------
def pose_spherical(theta_deg, phi_deg, radius):
    """Camera-to-world for a camera on a sphere (load_blender.py:10-34)."""
    th = theta_deg / 180.0 * np.pi
    ph = phi_deg / 180.0 * np.pi
    tr = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=F32)
    rp = torch.tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0],
                       [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], dtype=F32)
    rt = torch.tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                       [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], dtype=F32)
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=F32)
    return flip @ (rt @ (rp @ tr))


def intrinsics(H, W, focal):
    """K as run_nerf.py:238-242."""
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float64)


def get_rays(H, W, K, c2w):
    """Pinhole rays for every pixel (run_nerf_helpers.py:68-78).

    Pixel centres are the integers 0..W-1 / 0..H-1 (no +0.5); output is
    indexed [row, col, 3].  rays_d[c] = sum_k dir[k] * c2w[c, k].
    """
    c2w = torch.as_tensor(c2w, dtype=F32)
    col = torch.linspace(0, W - 1, W).view(1, W).expand(H, W)
    row = torch.linspace(0, H - 1, H).view(H, 1).expand(H, W)
    dx = (col - K[0][2]) / K[0][0]
    dy = -(row - K[1][2]) / K[1][1]
    dz = -torch.ones_like(dx)
    dirs = torch.stack([dx, dy, dz], -1)                       # [H,W,3]
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)    # same product/sum order
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    """numpy twin of get_rays (run_nerf_helpers.py:81-88)."""
    col, row = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(col - K[0][2]) / K[0][0], -(row - K[1][2]) / K[1][1], -np.ones_like(col)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Forward-facing NDC warp (run_nerf_helpers.py:91-108)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    sx = -1.0 / (W / (2.0 * focal))
    sy = -1.0 / (H / (2.0 * focal))
    o0 = sx * rays_o[..., 0] / rays_o[..., 2]
    o1 = sy * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------
# positional encoding
# --------------------------------------------------------------------------
def posenc(x, n_freqs):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]
    (run_nerf_helpers.py:15-63; log-sampled bands are exactly 2^k in fp32)."""
    out = [x]
    for k in range(n_freqs):
        f = torch.tensor(2.0 ** k, dtype=F32)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def posenc_dim(n_freqs, d=3):
    return d + 2 * n_freqs * d


# --------------------------------------------------------------------------
# MLP (model.py:8-63), functional over a state-dict of tensors
# --------------------------------------------------------------------------
PARAM_SHAPES = None


def nerf_param_shapes(D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,)):
    """Ordered (name, shape) list in `model.parameters()` order (model.py:20-34)."""
    shapes = []
    for i in range(D):
        if i == 0:
            fan_in = input_ch
        elif (i - 1) in skips:
            fan_in = W + input_ch
        else:
            fan_in = W
        shapes.append((f'pts_linears.{i}.weight', (W, fan_in)))
        shapes.append((f'pts_linears.{i}.bias', (W,)))
    shapes.append(('views_linears.0.weight', (W // 2, input_ch_views + W)))
    shapes.append(('views_linears.0.bias', (W // 2,)))
    shapes.append(('feature_linear.weight', (W, W)))
    shapes.append(('feature_linear.bias', (W,)))
    shapes.append(('alpha_linear.weight', (1, W)))
    shapes.append(('alpha_linear.bias', (1,)))
    shapes.append(('rgb_linear.weight', (3, W // 2)))
    shapes.append(('rgb_linear.bias', (3,)))
    return shapes


def init_nerf_params(gen, **kw):
    """Default nn.Linear init: U(+-1/sqrt(fan_in)) for weight and bias
    (kaiming_uniform a=sqrt(5); Appendix A of SURVEY).  Uses its own generator,
    it is NOT stream-compatible with torch.manual_seed + nn.Linear."""
    sd = OrderedDict()
    fan = None
    for name, shape in nerf_param_shapes(**kw):
        if name.endswith('weight'):
            fan = shape[1]
        bound = 1.0 / math.sqrt(fan)
        sd[name] = (torch.rand(shape, generator=gen, dtype=F32) * 2 - 1) * bound
    return sd


def nerf_forward(sd, x, D=8, input_ch=63, input_ch_views=27, skips=(4,)):
    """NeRF.forward with use_viewdirs=True (model.py:38-63).  Output [.., 4] =
    (rgb logits x3, sigma); no output activation."""
    pts, views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    h = pts
    for i in range(D):
        h = torch.nn.functional.linear(h, sd[f'pts_linears.{i}.weight'], sd[f'pts_linears.{i}.bias'])
        h = torch.relu(h)
        if i in skips:
            h = torch.cat([pts, h], -1)
    if 'output_linear.weight' in sd:   # use_viewdirs=False (model.py:35-36, 60-61): [.., output_ch] straight from the trunk
        return torch.nn.functional.linear(h, sd['output_linear.weight'], sd['output_linear.bias'])
    alpha = torch.nn.functional.linear(h, sd['alpha_linear.weight'], sd['alpha_linear.bias'])
    feat = torch.nn.functional.linear(h, sd['feature_linear.weight'], sd['feature_linear.bias'])
    h = torch.cat([feat, views], -1)
    h = torch.relu(torch.nn.functional.linear(h, sd['views_linears.0.weight'], sd['views_linears.0.bias']))
    rgb = torch.nn.functional.linear(h, sd['rgb_linear.weight'], sd['rgb_linear.bias'])
    return torch.cat([rgb, alpha], -1)


def run_network(sd, pts, viewdirs, multires=10, multires_views=4):
    """PE + MLP over [N,S,3] points (run_nerf.py:50-64); netchunk does not
    change results and is not restated."""
    flat = pts.reshape(-1, 3)
    emb = posenc(flat, multires)
    if viewdirs is None:               # run_nerf.py:55-59: no direction encoding without use_viewdirs
        out = nerf_forward(sd, emb, input_ch=posenc_dim(multires), input_ch_views=0)
        return out.reshape(list(pts.shape[:-1]) + [out.shape[-1]])
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    emb = torch.cat([emb, posenc(dirs, multires_views)], -1)
    out = nerf_forward(sd, emb, input_ch=posenc_dim(multires), input_ch_views=posenc_dim(multires_views))
    return out.reshape(list(pts.shape[:-1]) + [4])


# --------------------------------------------------------------------------
# sampling + compositing
# --------------------------------------------------------------------------
def coarse_z(near, far, n_samples, lindisp=False, t_rand=None):
    """Coarse depths (render.py:244-266).  near/far: [N,1].  t_rand: injected
    U[0,1) jitter [N,S] or None (perturb == 0)."""
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand(near.shape[0], n_samples)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def raw2outputs(raw, z_vals, rays_d, noise=None, white_bkgd=False):
    """Alpha compositing (render.py:149-192).  `noise` is the already scaled
    sigma noise [N,S] (raw_noise_std * N(0,1)) or None."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sigma = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * dists)
    trans = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf(bins, weights, n_samples, u=None):
    """Inverse-CDF sampling (run_nerf_helpers.py:112-155).  bins [N,M],
    weights [N,M-1]; u = injected uniforms [N,n_samples] or None for the
    deterministic linspace(0,1,n)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = torch.linspace(0.0, 1.0, steps=n_samples).expand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo = torch.gather(cdf, -1, below)
    cdf_hi = torch.gather(cdf, -1, above)
    bin_lo = torch.gather(bins, -1, below)
    bin_hi = torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def render_rays(ray_batch, sd_coarse, sd_fine, N_samples, N_importance=0, lindisp=False,
                white_bkgd=False, t_rand=None, u=None, noise0=None, noise1=None, retraw=False,
                multires=10, multires_views=4):
    """One chunk of volumetric rendering (render.py:195-305).  Random draws are
    injected: t_rand [N,Sc] (None <=> perturb==0), u [N,Ni] (None <=> det),
    noise0/noise1 = scaled sigma noise for the coarse / fine pass."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None   # render.py:218
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z = coarse_z(near, far, N_samples, lindisp, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw = run_network(sd_coarse, pts, viewdirs, multires, multires_views)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z, rays_d, noise0, white_bkgd)
    ret = {}
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'] = rgb_map, disp_map, acc_map
        ret['weights0'] = weights
        ret['z0'] = z
        z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, u).detach()
        z, _ = torch.sort(torch.cat([z, z_samples], -1), -1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        raw = run_network(sd_fine if sd_fine is not None else sd_coarse, pts, viewdirs, multires, multires_views)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z, rays_d, noise1, white_bkgd)
        ret['z_std'] = torch.std(z_samples, dim=-1, unbiased=False)
        ret['z_samples'] = z_samples
    ret['rgb_map'], ret['disp_map'], ret['acc_map'] = rgb_map, disp_map, acc_map
    ret['z_vals'] = z
    ret['weights'] = weights
    ret['depth_map'] = depth_map
    if retraw:
        ret['raw'] = raw
    return ret


def make_ray_batch(rays_o, rays_d, near, far, H=None, W=None, focal=None, ndc=False, use_viewdirs=True):
    """Pack [N,11] = o(3) d(3) near far viewdir(3) (render.py:59-80); [N,8] without view directions.
    viewdirs are normalised BEFORE the ndc warp."""
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1.0, rays_o, rays_d)
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    nr = near * torch.ones_like(rays_d[..., :1])
    fr = far * torch.ones_like(rays_d[..., :1])
    if not use_viewdirs:
        return torch.cat([rays_o, rays_d, nr, fr], -1)
    return torch.cat([rays_o, rays_d, nr, fr, viewdirs.reshape(-1, 3).float()], -1)


# --------------------------------------------------------------------------
# loss / optimiser
# --------------------------------------------------------------------------
def img2mse(x, y):
    """run_nerf_helpers.py:9."""
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    """run_nerf_helpers.py:10."""
    return -10.0 * torch.log(x) / torch.log(torch.tensor([10.0]))


class Adam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8), no weight decay,
    as constructed at run_nerf.py:99 -- restated explicitly."""

    def __init__(self, params, lr=5e-4, b1=0.9, b2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
                p.addcdiv_(m, denom, value=-step_size)


def lr_schedule(lrate, lrate_decay, global_iter):
    """run_nerf.py:498-502: uses the PRE-increment global_iter."""
    return lrate * (0.1 ** (global_iter / (lrate_decay * 1000)))


def train_step(sd_coarse, sd_fine, opt, ray_batch, target, N_samples, N_importance,
               white_bkgd=True, t_rand=None, u=None, lindisp=False, noise0=None, noise1=None):
    """One optimisation step (run_nerf.py:479-494): loss = mse(fine)+mse(coarse),
    autograd backward, Adam.  Returns (loss, loss_coarse, rgb_fine, grads)."""
    params = list(sd_coarse.values()) + (list(sd_fine.values()) if sd_fine is not None else [])
    for p in params:
        p.requires_grad_(True)
        p.grad = None
    ret = render_rays(ray_batch, sd_coarse, sd_fine, N_samples, N_importance, lindisp, white_bkgd,
                      t_rand, u, noise0, noise1)
    img_loss = img2mse(ret['rgb_map'], target)
    loss = img_loss
    img_loss0 = None
    if 'rgb0' in ret:
        img_loss0 = img2mse(ret['rgb0'], target)
        loss = loss + img_loss0
    grads = torch.autograd.grad(loss, params)
    for p in params:
        p.requires_grad_(False)
    opt.step(grads)
    return img_loss.detach(), (img_loss0.detach() if img_loss0 is not None else None), ret['rgb_map'].detach(), grads


# --------------------------------------------------------------------------
# per-(image, leaf) loss reduction that feeds the quadtree
# --------------------------------------------------------------------------
def leaf_loss_sumcount(rgb_gt, rgb_pred, leaf_tag, n_images, max_leaves):
    """Per-(image, leaf) sum of |gt - pred| over rays and channels + ray count: the inputs of the nerf++ fork's MEAN split rule
    (nerf++-ours/tree.py:609-632: `torch.mean(torch.abs(gt - pred))` over a leaf's rays).  A ray's term is rounded to a multiple of
    2^-30 as the device kernel does (csrc/train.hip: exact, order- and shard-independent fp64 sums).  -> (sums f64, counts i32)."""
    e = torch.abs(rgb_gt.float() - rgb_pred.float()).double().sum(-1)
    e = torch.round(e * 1073741824.0) / 1073741824.0
    flat = leaf_tag[:, 0].long() * max_leaves + leaf_tag[:, 1].long()
    sums = torch.zeros(n_images * max_leaves, dtype=torch.float64).index_add_(0, flat, e)
    counts = torch.zeros(n_images * max_leaves, dtype=torch.int32).index_add_(0, flat, torch.ones_like(flat, dtype=torch.int32))
    return sums, counts


def leaf_loss_max(rgb_gt, rgb_pred, leaf_tag, n_images, max_leaves):
    """Segmented max of |gt - pred| over rays and channels per (image, leaf)
    (tree.py:538 + 632-642 restated as one table).  leaf_tag [N,2] int64."""
    table = torch.zeros(n_images, max_leaves, dtype=F32)
    err = torch.abs(rgb_gt - rgb_pred).max(dim=-1).values
    flat = leaf_tag[:, 0] * max_leaves + leaf_tag[:, 1]
    table.view(-1).scatter_reduce_(0, flat, err, reduce='amax', include_self=True)
    return table
