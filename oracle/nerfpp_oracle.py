"""CPU oracle for the nerf++-ours additions -- TEST INFRASTRUCTURE ONLY.

PyTorch-CPU fp32 restatement of nerf++-ours/{ddp_train_nerf.py, ddp_model.py, nerf_network.py,
nerf_sample_ray_split.py}; citations are relative to /root/reference/nerf++-ours/.  Parity status:
PINNED by tests/golden/g10_*.npz recorded from the reference by oracle/make_golden.py.
"""
from collections import OrderedDict

import numpy as np
import torch

TINY_NUMBER = 1e-6   # utils.py:8
HUGE_NUMBER = 1e10   # utils.py:7
F32 = torch.float32


def get_rays_single_image(H, W, intrinsics, c2w):
    """nerf_sample_ray_split.py:10-34: OpenCV convention, pixel centres at +0.5, d = R K^-1 [u,v,1]."""
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    u = u.reshape(-1).astype(dtype=np.float32) + 0.5
    v = v.reshape(-1).astype(dtype=np.float32) + 0.5
    pixels = np.stack((u, v, np.ones_like(u)), axis=0)
    rays_d = np.dot(np.linalg.inv(intrinsics[:3, :3]), pixels)
    rays_d = np.dot(c2w[:3, :3], rays_d).transpose((1, 0))
    rays_o = np.tile(c2w[:3, 3].reshape((1, 3)), (rays_d.shape[0], 1))
    depth = np.linalg.inv(c2w)[2, 3] * np.ones((rays_o.shape[0],), dtype=np.float32)
    return rays_o, rays_d, depth   # float64 when c2w / intrinsics are (the reference does not cast here)


def intersect_sphere(ray_o, ray_d):
    """ddp_train_nerf.py:54-69: depth at which the ray leaves the unit sphere."""
    d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
    p = ray_o + d1.unsqueeze(-1) * ray_d
    ray_d_cos = 1.0 / torch.norm(ray_d, dim=-1)
    p_norm_sq = torch.sum(p * p, dim=-1)
    if (p_norm_sq >= 1.0).any():
        raise Exception('Not all your cameras are bounded by the unit sphere; please make sure the cameras are '
                        'normalized properly!')
    return d1 + torch.sqrt(1.0 - p_norm_sq) * ray_d_cos


def perturb_samples(z_vals, t_rand):
    """ddp_train_nerf.py:72-81 with the uniform draw injected."""
    mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], dim=-1)
    lower = torch.cat([z_vals[..., 0:1], mids], dim=-1)
    return lower + (upper - lower) * t_rand


def fg_depths(fg_far, near, N_samples):
    """ddp_train_nerf.py:357-360: near + i*step, i = 0..N-1 (stacked, not linspace)."""
    step = (fg_far - near) / (N_samples - 1)
    return torch.stack([near + i * step for i in range(N_samples)], dim=-1)


def bg_depths(n_rays, N_samples):
    """ddp_train_nerf.py:364-365."""
    return torch.linspace(0.0, 1.0, N_samples).view(1, N_samples).expand(n_rays, N_samples)


def sample_pdf(bins, weights, N_samples, u=None):
    """ddp_train_nerf.py:84-133.  bins [..,M+1], weights [..,M]; u injected or linspace (det)."""
    weights = weights + TINY_NUMBER
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., 0:1]), cdf], dim=-1)
    M = weights.shape[-1]
    if u is None:
        u = torch.linspace(0.0, 1.0, N_samples).view(1, N_samples).expand(list(weights.shape[:-1]) + [N_samples])
    above = torch.sum(u.unsqueeze(-1) >= cdf[..., :M].unsqueeze(-2), dim=-1).long()
    below = torch.clamp(above - 1, min=0)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < TINY_NUMBER, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    return b0 + t * (b1 - b0 + TINY_NUMBER)


def depth2pts_outside(ray_o, ray_d, depth):
    """ddp_model.py:16-45: inverted-sphere background point (x', y', z', 1/r)."""
    d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
    p_mid = ray_o + d1.unsqueeze(-1) * ray_d
    p_mid_norm = torch.norm(p_mid, dim=-1)
    ray_d_cos = 1.0 / torch.norm(ray_d, dim=-1)
    d2 = torch.sqrt(1.0 - p_mid_norm * p_mid_norm) * ray_d_cos
    p_sphere = ray_o + (d1 + d2).unsqueeze(-1) * ray_d
    rot_axis = torch.cross(ray_o, p_sphere, dim=-1)
    rot_axis = rot_axis / torch.norm(rot_axis, dim=-1, keepdim=True)
    phi = torch.asin(p_mid_norm)
    theta = torch.asin(p_mid_norm * depth)
    rot_angle = (phi - theta).unsqueeze(-1)
    p_new = p_sphere * torch.cos(rot_angle) + torch.cross(rot_axis, p_sphere, dim=-1) * torch.sin(rot_angle) + \
        rot_axis * torch.sum(rot_axis * p_sphere, dim=-1, keepdim=True) * (1.0 - torch.cos(rot_angle))
    p_new = p_new / torch.norm(p_new, dim=-1, keepdim=True)
    pts = torch.cat((p_new, depth.unsqueeze(-1)), dim=-1)
    depth_real = 1.0 / (depth + TINY_NUMBER) * torch.cos(theta) * ray_d_cos + d1
    return pts, depth_real


def embed(x, n_freqs):
    """nerf_network.py:11-60: [x, sin(f0 x), cos(f0 x), ...] for any input dim; f_k = 2^k."""
    out = [x]
    for k in range(n_freqs):
        f = float(2.0 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, dim=-1)


def mlpnet_param_shapes(input_ch, input_ch_viewdirs=27, D=8, W=256, skips=(4,)):
    """(name, shape) in MLPNet.parameters() order (nerf_network.py:70-120)."""
    out = []
    dim = input_ch
    for i in range(D):
        out.append((f'base_layers.{i}.0.weight', (W, dim)))
        out.append((f'base_layers.{i}.0.bias', (W,)))
        dim = W
        if i in skips and i != D - 1:
            dim += input_ch
    out += [('sigma_layers.0.weight', (1, dim)), ('sigma_layers.0.bias', (1,)),
            ('base_remap_layers.0.weight', (256, dim)), ('base_remap_layers.0.bias', (256,)),
            ('rgb_layers.0.weight', (W // 2, 256 + input_ch_viewdirs)), ('rgb_layers.0.bias', (W // 2,)),
            ('rgb_layers.2.weight', (3, W // 2)), ('rgb_layers.2.bias', (3,))]
    return out


def mlpnet_forward(sd, x, input_ch, input_ch_viewdirs=27, D=8, skips=(4,)):
    """MLPNet.forward (nerf_network.py:122-142) -> (rgb [..,3] after sigmoid, sigma [..] after abs)."""
    lin = torch.nn.functional.linear
    pts = x[..., :input_ch]
    base = torch.relu(lin(pts, sd['base_layers.0.0.weight'], sd['base_layers.0.0.bias']))
    for i in range(D - 1):
        if i in skips:
            base = torch.cat((pts, base), dim=-1)
        base = torch.relu(lin(base, sd[f'base_layers.{i + 1}.0.weight'], sd[f'base_layers.{i + 1}.0.bias']))
    sigma = torch.abs(lin(base, sd['sigma_layers.0.weight'], sd['sigma_layers.0.bias'])).squeeze(-1)
    remap = lin(base, sd['base_remap_layers.0.weight'], sd['base_remap_layers.0.bias'])
    views = x[..., -input_ch_viewdirs:]
    h = torch.relu(lin(torch.cat((remap, views), dim=-1), sd['rgb_layers.0.weight'], sd['rgb_layers.0.bias']))
    rgb = torch.sigmoid(lin(h, sd['rgb_layers.2.weight'], sd['rgb_layers.2.bias']))
    return rgb, sigma


def nerfnet_forward(sd_fg, sd_bg, ray_o, ray_d, fg_z_max, fg_z_vals, bg_z_vals, L=10, Lv=4):
    """NerfNet.forward (ddp_model.py:74-143)."""
    ray_d_norm = torch.norm(ray_d, dim=-1, keepdim=True)
    viewdirs = ray_d / ray_d_norm
    n = ray_d.shape[0]
    S = fg_z_vals.shape[-1]
    fg_pts = ray_o.unsqueeze(-2) + fg_z_vals.unsqueeze(-1) * ray_d.unsqueeze(-2)
    vd = viewdirs.unsqueeze(-2).expand(n, S, 3)
    fg_rgb, fg_sigma = mlpnet_forward(sd_fg, torch.cat((embed(fg_pts, L), embed(vd, Lv)), dim=-1), 3 + 6 * L)
    fg_dists = fg_z_vals[..., 1:] - fg_z_vals[..., :-1]
    fg_dists = ray_d_norm * torch.cat((fg_dists, fg_z_max.unsqueeze(-1) - fg_z_vals[..., -1:]), dim=-1)
    fg_alpha = 1.0 - torch.exp(-fg_sigma * fg_dists)
    T = torch.cumprod(1.0 - fg_alpha + TINY_NUMBER, dim=-1)
    bg_lambda = T[..., -1]
    T = torch.cat((torch.ones_like(T[..., 0:1]), T[..., :-1]), dim=-1)
    fg_weights = fg_alpha * T
    fg_rgb_map = torch.sum(fg_weights.unsqueeze(-1) * fg_rgb, dim=-2)
    fg_depth_map = torch.sum(fg_weights * fg_z_vals, dim=-1)

    Sb = bg_z_vals.shape[-1]
    bo = ray_o.unsqueeze(-2).expand(n, Sb, 3)
    bd = ray_d.unsqueeze(-2).expand(n, Sb, 3)
    bg_pts, _ = depth2pts_outside(bo, bd, bg_z_vals)
    vdb = viewdirs.unsqueeze(-2).expand(n, Sb, 3)
    inp = torch.cat((embed(bg_pts, L), embed(vdb, Lv)), dim=-1)
    inp = torch.flip(inp, dims=[-2])
    bz = torch.flip(bg_z_vals, dims=[-1])
    bg_dists = bz[..., :-1] - bz[..., 1:]
    bg_dists = torch.cat((bg_dists, HUGE_NUMBER * torch.ones_like(bg_dists[..., 0:1])), dim=-1)
    bg_rgb, bg_sigma = mlpnet_forward(sd_bg, inp, 4 + 8 * L)
    bg_alpha = 1.0 - torch.exp(-bg_sigma * bg_dists)
    T = torch.cumprod(1.0 - bg_alpha + TINY_NUMBER, dim=-1)[..., :-1]
    T = torch.cat((torch.ones_like(T[..., 0:1]), T), dim=-1)
    bg_weights = bg_alpha * T
    bg_rgb_map = torch.sum(bg_weights.unsqueeze(-1) * bg_rgb, dim=-2)
    bg_depth_map = torch.sum(bg_weights * bz, dim=-1)
    bg_rgb_map = bg_lambda.unsqueeze(-1) * bg_rgb_map
    bg_depth_map = bg_lambda * bg_depth_map
    return OrderedDict([('rgb', fg_rgb_map + bg_rgb_map), ('fg_weights', fg_weights), ('bg_weights', bg_weights),
                        ('fg_rgb', fg_rgb_map), ('fg_depth', fg_depth_map), ('bg_rgb', bg_rgb_map),
                        ('bg_depth', bg_depth_map), ('bg_lambda', bg_lambda)])


def cascade_step(sd_levels, ray_o, ray_d, target, cascade_samples, rand, min_depth=1e-4):
    """Forward/backward of one batch of train_step (ddp_train_nerf.py:347-404) for every cascade
    level, with all random draws injected: rand[m] = dict(fg_t, bg_t) for m=0 and dict(fg_u, bg_u)
    for m>0.  Returns per level (loss, grads [fg params..., bg params...], ret)."""
    outs = []
    fg_far = intersect_sphere(ray_o, ray_d)
    near = min_depth * torch.ones_like(ray_d[..., 0])
    ret = None
    for m, (sd_fg, sd_bg) in enumerate(sd_levels):
        N = cascade_samples[m]
        if m == 0:
            fg_depth = perturb_samples(fg_depths(fg_far, near, N), rand[m]['fg_t'])
            bg_depth = perturb_samples(bg_depths(ray_o.shape[0], N), rand[m]['bg_t'])
        else:
            fw = ret['fg_weights'].detach()[..., 1:-1]
            fmid = 0.5 * (fg_depth[..., 1:] + fg_depth[..., :-1])
            fg_depth, _ = torch.sort(torch.cat((fg_depth, sample_pdf(fmid, fw, N, rand[m]['fg_u'])), dim=-1))
            bw = ret['bg_weights'].detach()[..., 1:-1]
            bmid = 0.5 * (bg_depth[..., 1:] + bg_depth[..., :-1])
            bg_depth, _ = torch.sort(torch.cat((bg_depth, sample_pdf(bmid, bw, N, rand[m]['bg_u'])), dim=-1))
        params = list(sd_fg.values()) + list(sd_bg.values())
        for p in params:
            p.requires_grad_(True)
        ret = nerfnet_forward(sd_fg, sd_bg, ray_o, ray_d, fg_far, fg_depth, bg_depth)
        loss = torch.mean((ret['rgb'] - target) ** 2)
        grads = torch.autograd.grad(loss, params)
        for p in params:
            p.requires_grad_(False)
        outs.append((loss.detach(), grads, {k: v.detach() for k, v in ret.items()}, fg_depth.clone(), bg_depth.clone()))
    return outs
