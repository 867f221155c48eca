"""Synthetic Lego-like data (no dataset ships with the reference and there is no network):
cameras on a sphere (`pose_spherical`, load_blender.py:29-34) looking at an analytic scene of
coloured Gaussian density blobs over a white background.  Used by bench / tests / examples to
produce identical training targets for the HIP path and the CPU oracle."""
import os

import numpy as np
import torch


def pose_spherical(theta_deg, phi_deg, radius):
    th, ph = theta_deg / 180.0 * np.pi, phi_deg / 180.0 * np.pi
    tr = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=np.float64)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    return torch.tensor(flip @ (rt @ (rp @ tr)), dtype=torch.float32)


BLOBS = [  # centre, sigma, peak density, colour
    ((0.0, 0.0, 0.0), 0.45, 12.0, (0.9, 0.2, 0.1)),
    ((0.6, 0.3, 0.2), 0.30, 16.0, (0.1, 0.7, 0.2)),
    ((-0.5, -0.4, 0.3), 0.35, 14.0, (0.15, 0.25, 0.9)),
]


# FASTNERF_SCENE_CUTOFF=c (experiments): density exactly zero beyond c standard deviations of a blob's centre -- solid bodies
# with empty space around them instead of Gaussian tails that never vanish (tools/live_trajectory.py)
CUTOFF = float(os.environ.get('FASTNERF_SCENE_CUTOFF', '0') or 0)


def scene(pts):
    """pts [..,3] float64 -> (sigma [..], rgb [..,3])."""
    sig = torch.zeros(pts.shape[:-1], dtype=pts.dtype)
    col = torch.zeros(pts.shape, dtype=pts.dtype)
    for c, s, d, rgb in BLOBS:
        r2 = ((pts - torch.tensor(c, dtype=pts.dtype)) ** 2).sum(-1)
        w = d * torch.exp(-r2 / (2 * s * s))
        if CUTOFF > 0:
            w = torch.where(r2 <= (CUTOFF * s) ** 2, w, torch.zeros_like(w))
        sig = sig + w
        col = col + w[..., None] * torch.tensor(rgb, dtype=pts.dtype)
    return sig, col / (sig[..., None] + 1e-12)


def render_images(H, W, focal, poses, near=2.0, far=6.0, n_quad=256):
    """Reference images [n,H,W,3] by fp64 quadrature of the analytic scene (white background)."""
    out = []
    t = torch.linspace(near, far, n_quad, dtype=torch.float64)
    dt = (far - near) / (n_quad - 1)
    cols, rows = torch.meshgrid(torch.arange(W, dtype=torch.float64), torch.arange(H, dtype=torch.float64), indexing='xy')
    dirs = torch.stack([(cols - 0.5 * W) / focal, -(rows - 0.5 * H) / focal, -torch.ones_like(cols)], -1)
    for c2w in poses:
        c2w = c2w.double()
        rd = (dirs[..., None, :] * c2w[:3, :3]).sum(-1)
        ro = c2w[:3, 3]
        pts = ro + rd[..., None, :] * t[:, None]
        sig, col = scene(pts)
        alpha = 1 - torch.exp(-sig * dt * rd.norm(dim=-1, keepdim=True))
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-10], -1), -1)[..., :-1]
        w = alpha * T
        rgb = (w[..., None] * col).sum(-2) + (1 - w.sum(-1, keepdim=True))
        out.append(rgb.float())
    return torch.stack(out, 0)


def render_rays(rays_o, rays_d, near=2.0, far=6.0, n_quad=128, cutoff=None):
    """Colours [N,3] of the analytic scene along arbitrary rays, by the same quadrature as render_images, on the
    rays' device (fp32 is plenty for training targets).  Used by bench.py's trained-scene legs.  cutoff = c > 0: the density is
    exactly zero beyond c standard deviations of each blob's centre (solid bodies in empty space instead of Gaussian tails)."""
    dev, dt_ = rays_o.device, rays_o.dtype
    cutoff = CUTOFF if cutoff is None else cutoff
    t = torch.linspace(near, far, n_quad, device=dev, dtype=dt_)
    step = (far - near) / (n_quad - 1)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * t[None, :, None]
    sig = torch.zeros(pts.shape[:-1], device=dev, dtype=dt_)
    col = torch.zeros(pts.shape, device=dev, dtype=dt_)
    for c, s, d, rgb in BLOBS:
        r2 = ((pts - torch.tensor(c, device=dev, dtype=dt_)) ** 2).sum(-1)
        w = d * torch.exp(-r2 / (2 * s * s))
        if cutoff > 0:
            w = torch.where(r2 <= (cutoff * s) ** 2, w, torch.zeros_like(w))
        sig = sig + w
        col = col + w[..., None] * torch.tensor(rgb, device=dev, dtype=dt_)
    col = col / (sig[..., None] + 1e-12)
    alpha = 1 - torch.exp(-sig * step * rays_d.norm(dim=-1, keepdim=True))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1 - alpha + 1e-10], -1), -1)[..., :-1]
    w = alpha * T
    return (w[..., None] * col).sum(-2) + (1 - w.sum(-1, keepdim=True))


def make_dataset(n_images=8, H=32, W=32, fov=0.6911112070083618, radius=4.0, phi=-30.0, device=None):
    """Cameras on a circle around the analytic scene and their images.  device=None: fp64 quadrature on the host (what the
    tests and goldens use); device='cuda': the same scene rendered on the GPU with render_rays (fp32, 256 steps) -- for
    full-resolution probes, where the host version takes minutes."""
    focal = 0.5 * W / np.tan(0.5 * fov)
    poses = torch.stack([pose_spherical(-180.0 + 360.0 * k / n_images, phi, radius)[:3, :4] for k in range(n_images)], 0)
    if device is None:
        return render_images(H, W, focal, poses), poses, focal
    from . import ops
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    imgs = []
    for i in range(n_images):
        ro, rd = ops.gen_rays(H, W, K, poses[i].to(device))
        flat_o, flat_d = ro.reshape(-1, 3), rd.reshape(-1, 3)
        rows = [render_rays(flat_o[s:s + 65536], flat_d[s:s + 65536], n_quad=256) for s in range(0, H * W, 65536)]
        imgs.append(torch.cat(rows, 0).reshape(H, W, 3).cpu())
    return torch.stack(imgs, 0), poses, focal
