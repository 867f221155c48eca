#!/usr/bin/env python3
"""Golden vectors for the nerf++-ours additions (G10), recorded from the REFERENCE itself
(build container only; same stubbing approach as make_golden.py).  Data-only fixtures ->
tests/golden/g10_*.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import install_stubs, OUT  # noqa: E402

REF = '/root/reference/nerf++-ours'


class Args:
    netdepth = 8; netwidth = 256; max_freq_log2 = 10; max_freq_log2_viewdirs = 4; use_viewdirs = True
    batch_size = 64; lrate = 5e-4; lambda_autoexpo = 1.0; optim_autoexpo = False


def main():
    assert os.path.isdir(REF)
    install_stubs()
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, REF)
    import ddp_train_nerf as D
    import ddp_model as M
    import nerf_sample_ray_split as RS
    g = torch.Generator().manual_seed(4321)
    os.makedirs(OUT, exist_ok=True)

    # ---- rays, sphere intersection, inverted-sphere points, sampler ---------------------------
    intr = np.array([[300.0, 0, 40.0, 0], [0, 300.0, 30.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    c2w = np.eye(4); c2w[:3, 3] = [0.1, -0.2, 0.3]
    ro_s, rd_s, dep_s = RS.get_rays_single_image(6, 8, intr, c2w)
    n = 40
    ray_o = (torch.rand(n, 3, generator=g) - 0.5) * 0.8
    ray_d = torch.randn(n, 3, generator=g)
    fg_far = D.intersect_sphere(ray_o, ray_d)
    depth = torch.rand(n, 16, generator=g) * 0.98 + 0.01
    pts, depth_real = M.depth2pts_outside(ray_o[:, None].expand(n, 16, 3), ray_d[:, None].expand(n, 16, 3), depth)
    bins = torch.sort(torch.rand(n, 63, generator=g), -1).values
    w = torch.rand(n, 62, generator=g) ** 3
    s_det = D.sample_pdf(bins, w, 128, det=True)
    torch.manual_seed(11); u = torch.rand(n, 128)
    torch.manual_seed(11); s_u = D.sample_pdf(bins, w, 128, det=False)
    np.savez(os.path.join(OUT, 'g10_pp_ops.npz'), intr=intr, c2w=c2w, ro_s=ro_s, rd_s=rd_s, dep_s=dep_s,
             ray_o=ray_o.numpy(), ray_d=ray_d.numpy(), fg_far=fg_far.numpy(), depth=depth.numpy(), pts=pts.numpy(),
             depth_real=depth_real.numpy(), bins=bins.numpy(), w=w.numpy(), s_det=s_det.numpy(), u=u.numpy(),
             s_u=s_u.numpy())

    # ---- one train_step batch over a 2-level cascade --------------------------------------------
    torch.manual_seed(0)
    nets = [M.NerfNetWithAutoExpo(Args, optim_autoexpo=False) for _ in range(2)]
    optims = [torch.optim.Adam(nt.parameters(), lr=Args.lrate) for nt in nets]
    wts = {}
    for m, nt in enumerate(nets):
        for k, v in nt.nerf_net.state_dict().items():
            wts[f'l{m}.{k}'] = v.detach().numpy().copy()
    np.savez(os.path.join(OUT, 'g10_pp_weights.npz'), **wts)
    N = 24
    ro = (torch.rand(N, 3, generator=g) - 0.5) * 0.8
    rd = torch.randn(N, 3, generator=g)
    tgt = torch.rand(N, 3, generator=g)
    models = {'cascade_level': 2, 'cascade_samples': [64, 128], 'net_0': nets[0], 'net_1': nets[1],
              'optim_0': optims[0], 'optim_1': optims[1]}
    Args.batch_size = N
    torch.manual_seed(7)
    rgb_pred = D.train_step(models, ro, rd, tgt, Args)
    torch.manual_seed(7)
    draws = {'fg_t': torch.rand(N, 64), 'bg_t': torch.rand(N, 64), 'fg_u': torch.rand(N, 128), 'bg_u': torch.rand(N, 128)}
    rec = {'ro': ro.numpy(), 'rd': rd.numpy(), 'target': tgt.numpy(), 'rgb_pred': rgb_pred.numpy()}
    rec.update({k: v.numpy() for k, v in draws.items()})
    for m, nt in enumerate(nets):
        for k, p in nt.nerf_net.named_parameters():
            gq = p.grad.detach().numpy().copy()
            rec[f'grad.l{m}.{k}'] = gq if gq.size <= 40000 else gq[:16]   # big matrices: first 16 rows only
    # level-0 forward outputs for the same inputs (weights were stepped once: reload the saved ones)
    nets[0].nerf_net.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in wts.items() if k.startswith('l0.')})
    fg_far = D.intersect_sphere(ro, rd)
    near = 1e-4 * torch.ones_like(rd[..., 0])
    step = (fg_far - near) / 63
    fg_depth = torch.stack([near + i * step for i in range(64)], dim=-1)
    mids = .5 * (fg_depth[..., 1:] + fg_depth[..., :-1])
    fg_depth = torch.cat([fg_depth[..., :1], mids], -1) + (torch.cat([mids, fg_depth[..., -1:]], -1) -
                                                          torch.cat([fg_depth[..., :1], mids], -1)) * draws['fg_t']
    bgd = torch.linspace(0., 1., 64).view(1, 64).expand(N, 64)
    mids = .5 * (bgd[..., 1:] + bgd[..., :-1])
    bg_depth = torch.cat([bgd[..., :1], mids], -1) + (torch.cat([mids, bgd[..., -1:]], -1) -
                                                     torch.cat([bgd[..., :1], mids], -1)) * draws['bg_t']
    with torch.no_grad():
        ret = nets[0].nerf_net(ro, rd, fg_far, fg_depth, bg_depth)
    for k in ('rgb', 'fg_weights', 'bg_weights', 'fg_rgb', 'bg_rgb', 'bg_lambda', 'fg_depth', 'bg_depth'):
        rec['l0.' + k] = ret[k].numpy()
    rec['l0.fg_z'] = fg_depth.numpy(); rec['l0.bg_z'] = bg_depth.numpy(); rec['fg_far'] = fg_far.numpy()
    np.savez(os.path.join(OUT, 'g10_pp_step.npz'), **rec)
    # ---- G12: nerf++ quadtree fork (prob=True picks, MEAN criterion) ---------------------------
    import tree as TP   # nerf++-ours/tree.py (REF is first on sys.path)

    class RS:           # the attributes QuadTreeManager reads from a RaySamplerSingleImage (tree.py:171-184)
        pass
    Ht, Wt, nimg = 48, 40, 2
    samplers = []
    for i in range(nimg):
        rs = RS()
        rs.H, rs.W = Ht, Wt
        rr, cc = np.meshgrid(np.arange(Ht), np.arange(Wt), indexing='ij')
        tex = 0.5 + 0.5 * np.sin(0.37 * rr + 0.11 * i) * np.cos(0.23 * cc)          # smooth texture for the variance map
        rs.img = np.stack([tex, (rr * Wt + cc) / 4096.0, np.full_like(tex, i / 8.0)], -1).astype(np.float32).reshape(-1, 3)
        rs.rays_o = np.tile(np.array([[0.1 * i, 0.0, 0.2]], dtype=np.float32), (Ht * Wt, 1))
        rs.rays_d = np.stack([cc.reshape(-1) / Wt - 0.5, rr.reshape(-1) / Ht - 0.5, -np.ones(Ht * Wt)], -1).astype(np.float32)
        samplers.append(rs)
    mgr = TP.QuadTreeManager(samplers, mseThres=0.0, max_depth=2)
    rec = {'H': Ht, 'W': Wt, 'images': np.stack([s_.img.reshape(Ht, Wt, 3) for s_ in samplers], 0),
           'rays_d': np.stack([s_.rays_d.reshape(Ht, Wt, 3) for s_ in samplers], 0),
           'sharp': np.stack(mgr.processor.sharp_imgs, 0)}
    for rnd in range(4):
        torch.manual_seed(200 + rnd)
        np.random.seed(300 + rnd)
        o, d, rgbt = mgr.gen_rays_v3_multiThread(down_scale=1, prob=True, rand=0.5, last_epoch=False)
        rec[f'r{rnd}_leaf_id'] = mgr.result_leaf_id.numpy().copy()
        rec[f'r{rnd}_rgb'] = rgbt.numpy().copy()
        rec[f'r{rnd}_d'] = d.numpy().copy()
        pred = torch.clamp(rgbt + (torch.rand(rgbt.shape, generator=g) - 0.5) * 0.2 *
                           (torch.rand(rgbt.shape[0], 1, generator=g) < 0.5).float(), 0, 1)
        rec[f'r{rnd}_pred'] = pred.numpy().copy()
        mgr.adjust_tree_multiThread(rgbt, pred, thres=0.012)
        for ti in range(nimg):
            rec[f'r{rnd}_after_t{ti}'] = np.array([[c_.x0, c_.y0, c_.x1, c_.y1] for c_ in mgr.childrens[ti]], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'g12_pp_tree.npz'), **rec)
    print('wrote g10 goldens', sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.startswith('g10')))


if __name__ == '__main__':
    main()
