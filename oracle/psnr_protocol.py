"""Test infrastructure (CPU): the paired PSNR-at-equal-iterations protocol behind tests/golden/g22_psnr_cpu_ensemble.npz.

north_star: "PSNR within 0.1 dB at equal iteration count".  Free-running trajectories of the same batches are chaotic, and
the level a run reaches after 200 iterations depends far more on the INITIAL WEIGHTS (27.8 ... 30.5 dB over ten seeds, one seed in six
never leaves the empty-scene solution) than on the arithmetic, so the comparison is PAIRED: for every initialisation seed s the CPU
oracle and the GPU start from the same weights, see the same batches and the same injected t_rand / u, and the statistic is the mean
over seeds of PSNR_gpu(s) - PSNR_cpu(s).

Everything both sides need is generated here ON THE CPU from seeds (no GPU ray generation: the build container that records the CPU
ensemble has none), so the inputs are bit-identical on both sides:
  cameras   100 x pose_spherical(-180 + 3.6 k, -30, 4), 800 x 800, focal of bench.py;
  batches   ITERS x RAYS uniformly drawn pixels of all cameras, torch CPU generator seed 2; targets = the analytic scene;
  jitter    t_rand [ITERS, RAYS, 64], u [ITERS, RAYS, 128] from the same generator;
  held out  HELD_OUT pixels from seed 7, rendered with perturb = 0;
  weights   init_weights(s): oracle init_nerf_params under torch.Generator().manual_seed(5000 + s), coarse then fine."""
import numpy as np
import torch

from . import nerf_oracle as O

H = W = 800
FOCAL = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
N_SAMPLES, N_IMPORTANCE = 64, 128
ITERS, RAYS, HELD_OUT, WINDOW = 200, 256, 1024, 20
N_CAMERAS = 100


def cameras():
    return torch.stack([O.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(N_CAMERAS)], 0)


def rays_of(pix, poses):
    """Pinhole rays of pixels [n, 3] = (camera, row, column), the convention of get_rays (run_nerf_helpers.py:68-78)."""
    i, r, c = pix[:, 0].long(), pix[:, 1].float(), pix[:, 2].float()
    dirs = torch.stack([(c - 0.5 * W) / FOCAL, -(r - 0.5 * H) / FOCAL, -torch.ones_like(c)], -1)
    rot = poses[i][:, :3, :3]
    return poses[i][:, :3, 3].contiguous(), (dirs[:, None, :] * rot).sum(-1)


def draw_pixels(gen, n):
    return torch.stack([torch.randint(0, N_CAMERAS, (n,), generator=gen), torch.randint(0, H, (n,), generator=gen),
                        torch.randint(0, W, (n,), generator=gen)], 1)


LONG_ITERS = 1000        # the longer horizon (G23): the same protocol with 1000 iterations, batch seed 3


def inputs(scene_fn, iters=None, batch_seed=2):
    """scene_fn(rays_o, rays_d) -> colours (fastnerf.synthetic.render_rays with cutoff 0, on CPU tensors)."""
    ITERS = globals()['ITERS'] if iters is None else int(iters)
    poses = cameras()
    g = torch.Generator().manual_seed(batch_seed)
    ro, rd, tgt = [], [], []
    for _ in range(ITERS):
        o, d = rays_of(draw_pixels(g, RAYS), poses)
        ro.append(o); rd.append(d); tgt.append(scene_fn(o, d))
    t_rand = torch.rand(ITERS, RAYS, N_SAMPLES, generator=g)
    u = torch.rand(ITERS, RAYS, N_IMPORTANCE, generator=g)
    ho_o, ho_d = rays_of(draw_pixels(torch.Generator().manual_seed(7), HELD_OUT), poses)
    return {'ro': torch.stack(ro), 'rd': torch.stack(rd), 'tgt': torch.stack(tgt), 't_rand': t_rand, 'u': u,
            'ho_ro': ho_o, 'ho_rd': ho_d, 'ho_tgt': scene_fn(ho_o, ho_d)}


def init_weights(seed):
    gen = torch.Generator().manual_seed(5000 + seed)
    return O.init_nerf_params(gen), O.init_nerf_params(gen)


def psnr(mse):
    return float(-10.0 * np.log10(mse))


def jitter_weights(sdc, sdf, seed, member):
    """The NULL member of the paired design: the same initialisation times (1 + 1e-6 N(0, 1)), i.e. a perturbation of
    the size of fp32 rounding.  Two CPU runs that differ by it bound what "another arithmetic of fp32 width" can look like."""
    g = torch.Generator().manual_seed(7000 + 10 * seed + member)
    with torch.no_grad():
        for sd in (sdc, sdf):
            for w in sd.values():
                w.mul_(1.0 + 1e-6 * torch.randn(w.shape, generator=g))


def held_out_psnr(sdc, sdf, data):
    with torch.no_grad():
        rb = O.make_ray_batch(data['ho_ro'], data['ho_rd'], 2.0, 6.0)
        se = 0.0
        for s in range(0, HELD_OUT, 256):
            ret = O.render_rays(rb[s:s + 256], sdc, sdf, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)
            se += float(((ret['rgb_map'] - data['ho_tgt'][s:s + 256]) ** 2).sum())
    return psnr(se / (HELD_OUT * 3))


def cpu_run(seed, data, member=0, return_weights=False):
    """One free run of the CPU oracle over all iterations `data` holds -> (train PSNR over the last WINDOW iterations, held-out PSNR,
    first loss) [+ the final (coarse, fine) state dicts].  member > 0: the jittered initialisation of jitter_weights."""
    ITERS = data['ro'].shape[0]
    sdc, sdf = init_weights(seed)
    if member:
        jitter_weights(sdc, sdf, seed, member)
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    losses = []
    for it in range(ITERS):
        opt.lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4       # pre-increment rule (run_nerf.py:498-508)
        rb = O.make_ray_batch(data['ro'][it], data['rd'][it], 2.0, 6.0)
        l1, _, _, _ = O.train_step(sdc, sdf, opt, rb, data['tgt'][it], N_SAMPLES, N_IMPORTANCE, True, t_rand=data['t_rand'][it], u=data['u'][it])
        losses.append(float(l1))
    out = (psnr(np.mean(losses[-WINDOW:])), held_out_psnr(sdc, sdf, data), losses[0])
    return out + ((sdc, sdf),) if return_weights else out
