#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE itself (build container only).

Imports /root/reference/nerf-ours with stub modules for the packages this image
lacks (cv2, threadpool, imageio, colour, configargparse) and `.cuda()` patched
to identity, runs fixed-seed cases and writes small `.npz` fixtures
to tests/golden/.  The fixtures hold DATA only (inputs, injected randoms,
expected outputs); no reference source travels.

Run:  python oracle/make_golden.py            (needs /root/reference)
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/nerf-ours'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')   # the committed fixtures
# FASTNERF_GOLDEN_OUT=<dir> writes somewhere else (oracle/check_goldens.sh regenerates everything there and compares);
# generators that build on an earlier fixture (g7_weights.npz) read it from the same directory, so run make_golden.py first
OUT = os.environ.get('FASTNERF_GOLDEN_OUT', GOLDEN)


def install_stubs():
    cv2 = types.ModuleType('cv2')

    def blur(img, k):
        # 3x3 box filter with reflect-101 border (only executed, never compared)
        pad = np.pad(img, ((1, 1), (1, 1)) + ((0, 0),) * (img.ndim - 2), mode='reflect')
        out = np.zeros_like(img)
        for dx in range(3):
            for dy in range(3):
                out = out + pad[dx:dx + img.shape[0], dy:dy + img.shape[1]]
        return out / 9.0

    cv2.blur = blur
    cv2.sqrt = np.sqrt
    cv2.COLOR_BGR2GRAY = 6
    cv2.INTER_AREA = 3
    cv2.cvtColor = lambda img, code: (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2])
    cv2.cv2 = cv2
    sys.modules['cv2'] = cv2

    tp = types.ModuleType('threadpool')

    class _Req:
        def __init__(self, fn, kw):
            self.fn, self.kw = fn, kw

    class ThreadPool:
        def __init__(self, n):
            self.q = []

        def putRequest(self, r):
            self.q.append(r)

        def wait(self):
            for r in self.q:
                r.fn(**r.kw)
            self.q = []

    tp.ThreadPool = ThreadPool
    tp.makeRequests = lambda fn, args: [_Req(fn, a[1]) for a in args]
    sys.modules['threadpool'] = tp

    for name in ('imageio', 'colour'):
        m = types.ModuleType(name)
        m.Color = object
        sys.modules[name] = m
    import argparse
    ca = types.ModuleType('configargparse')
    ca.ArgumentParser = argparse.ArgumentParser
    sys.modules['configargparse'] = ca

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def pose_spherical_np(theta, phi, radius):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from nerf_oracle import pose_spherical
    return pose_spherical(theta, phi, radius)


def main():
    assert os.path.isdir(REF), 'reference not present; goldens can only be made in the build container'
    install_stubs()
    sys.path.insert(0, REF)
    import run_nerf_helpers as H
    import render as R
    import model as M
    import run_nerf as RN
    import tree as T

    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(1234)

    # ---- G1 get_rays -------------------------------------------------------
    c2w = pose_spherical_np(30.0, -30.0, 4.0)[:3, :4]
    focal = 0.5 * 800 / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 400.0], [0, focal, 400.0], [0, 0, 1]])
    o_s, d_s = H.get_rays(6, 8, np.array([[10.0, 0, 4.0], [0, 10.0, 3.0], [0, 0, 1]]), c2w)
    o_b, d_b = H.get_rays(800, 800, K, c2w)
    idx = np.array([[0, 0], [0, 799], [799, 0], [799, 799], [400, 400], [123, 456], [700, 13]])
    onp, dnp = H.get_rays_np(6, 8, np.array([[10.0, 0, 4.0], [0, 10.0, 3.0], [0, 0, 1]]), c2w.numpy())
    np.savez(os.path.join(OUT, 'g1_get_rays.npz'), c2w=c2w.numpy(), K=K, focal=focal,
             small_o=o_s.numpy(), small_d=d_s.numpy(), idx=idx,
             big_o=o_b.numpy()[idx[:, 0], idx[:, 1]], big_d=d_b.numpy()[idx[:, 0], idx[:, 1]],
             np_o=np.ascontiguousarray(onp), np_d=dnp)

    # ---- G2 ndc_rays -------------------------------------------------------
    ro = torch.rand(64, 3, generator=g) * 0.2 - 0.1
    rd = torch.cat([torch.rand(64, 2, generator=g) - 0.5, -torch.ones(64, 1)], -1)
    no, nd = H.ndc_rays(756, 1008, 815.13, 1.0, ro, rd)
    np.savez(os.path.join(OUT, 'g2_ndc.npz'), ro=ro.numpy(), rd=rd.numpy(), no=no.numpy(), nd=nd.numpy(),
             H=756, W=1008, focal=815.13)

    # ---- G3 embedder -------------------------------------------------------
    x = (torch.rand(256, 3, generator=g) * 2 - 1) * 6.0
    e10, d10 = H.get_embedder(10, 0)
    e4, d4 = H.get_embedder(4, 0)
    np.savez(os.path.join(OUT, 'g3_embed.npz'), x=x.numpy(), e10=e10(x).numpy(), e4=e4(x).numpy())

    # ---- G5 raw2outputs ----------------------------------------------------
    for S in (64, 192):
        raw = torch.randn(64, S, 4, generator=g) * 2.0
        raw.requires_grad_(True)
        z = torch.sort(torch.rand(64, S, generator=g) * 4 + 2, -1).values
        rd = torch.randn(64, 3, generator=g)
        for wb in (False, True):
            rgb, disp, acc, w, depth = R.raw2outputs(raw, z, rd, 0, wb)
            cot = torch.randn(64, 3, generator=g)
            graw, = torch.autograd.grad((rgb * cot).sum(), raw)
            np.savez(os.path.join(OUT, f'g5_raw2out_S{S}_wb{int(wb)}.npz'), raw=raw.detach().numpy(), z=z.numpy(),
                     rd=rd.numpy(), rgb=rgb.detach().numpy(), disp=disp.detach().numpy(), acc=acc.detach().numpy(),
                     weights=w.detach().numpy(), depth=depth.detach().numpy(), cot=cot.numpy(), graw=graw.numpy())
    # noise path: reference pytest hook draws np.random.rand (seed 0) * std
    raw = torch.randn(16, 64, 4, generator=g)
    z = torch.sort(torch.rand(16, 64, generator=g), -1).values
    rd = torch.randn(16, 3, generator=g)
    rgb, disp, acc, w, depth = R.raw2outputs(raw, z, rd, 1.0, False, pytest=True)
    np.random.seed(0)
    noise = np.random.rand(16, 64).astype(np.float32) * 1.0
    np.savez(os.path.join(OUT, 'g5_raw2out_noise.npz'), raw=raw.numpy(), z=z.numpy(), rd=rd.numpy(), noise=noise,
             rgb=rgb.numpy(), disp=disp.numpy(), acc=acc.numpy(), weights=w.numpy(), depth=depth.numpy())

    # ---- G6 sample_pdf -----------------------------------------------------
    bins = torch.sort(torch.rand(32, 63, generator=g) * 4 + 2, -1).values
    w_rand = torch.rand(32, 62, generator=g)
    w_flat = torch.zeros(32, 62)
    w_spike = torch.zeros(32, 62)
    w_spike[:, 17] = 1.0
    cases = {}
    for name, w in (('rand', w_rand), ('flat', w_flat), ('spike', w_spike)):
        cases[name + '_det'] = H.sample_pdf(bins, w, 128, det=True).numpy()
        np.random.seed(0)
        u = np.random.rand(32, 128)
        cases[name + '_u'] = H.sample_pdf(bins, w, 128, det=False, pytest=True).numpy()
    np.savez(os.path.join(OUT, 'g6_sample_pdf.npz'), bins=bins.numpy(), w_rand=w_rand.numpy(), w_flat=w_flat.numpy(),
             w_spike=w_spike.numpy(), u=u.astype(np.float32), **cases)

    # ---- G7 render_rays end to end ----------------------------------------
    class A:
        pass
    args = A()
    args.multires, args.multires_views, args.i_embed = 10, 4, 0
    args.use_viewdirs, args.N_importance, args.netdepth, args.netwidth = True, 128, 8, 256
    args.netdepth_fine, args.netwidth_fine, args.netchunk = 8, 256, 65536
    args.lrate, args.basedir, args.expname, args.ft_path, args.no_reload = 5e-4, '/tmp', 'golden_tmp', None, True
    args.perturb, args.N_samples, args.white_bkgd, args.raw_noise_std = 1.0, 64, True, 0.0
    args.dataset_type, args.no_ndc, args.lindisp = 'blender', False, False
    os.makedirs('/tmp/golden_tmp', exist_ok=True)
    torch.manual_seed(0)
    kw_train, kw_test, _, _, grad_vars, optim = RN.create_nerf(args)
    o_b, d_b = H.get_rays(800, 800, K, c2w)
    sel = torch.randint(0, 800 * 800, (64,), generator=g)
    ro = o_b.reshape(-1, 3)[sel].contiguous()
    rd = d_b.reshape(-1, 3)[sel].contiguous()
    sdc = {k.replace('module.', ''): v.detach().numpy().copy() for k, v in kw_train['network_fn'].state_dict().items()}
    sdf = {k.replace('module.', ''): v.detach().numpy().copy() for k, v in kw_train['network_fine'].state_dict().items()}
    np.savez(os.path.join(OUT, 'g7_weights.npz'), **{'c.' + k: v for k, v in sdc.items()},
             **{'f.' + k: v for k, v in sdf.items()})


    # ---- G4 NeRF forward + grads (coarse net of G7's create_nerf; weights in g7_weights 'c.*') ----
    net = kw_train['network_fn']
    xin = torch.randn(512, 90, generator=g) * 0.7
    out = net(xin)
    cot = torch.randn(512, 4, generator=g)
    net.zero_grad()
    (out * cot).sum().backward()
    gr = {'grad.' + k.replace('module.', ''): p.grad.numpy().copy() for k, p in net.named_parameters()}
    net.zero_grad()
    np.savez(os.path.join(OUT, 'g4_mlp.npz'), x=xin.numpy(), out=out.detach().numpy(), cot=cot.numpy(), **gr)

    def run_render(kw, pytest, **over):
        kk = dict(kw)
        kk.update(over)
        rgb, disp, acc, ex = R.render(800, 800, K, chunk=32768, rays=torch.stack([ro, rd], 0), retraw=True,
                                      near=2.0, far=6.0, pytest=pytest, **kk)
        d = {'rgb': rgb, 'disp': disp, 'acc': acc}
        d.update(ex)
        return {k: v.detach().numpy() for k, v in d.items()}

    # (a) deterministic test-mode render (perturb 0, det sample_pdf), 64+128
    ra = run_render(kw_test, False)
    # (b) train-mode with the pytest hook: t_rand and u both = np.random.seed(0); rand(...)
    rb = run_render(kw_train, True)
    np.random.seed(0)
    t_rand = np.random.rand(64, 64).astype(np.float32)
    np.random.seed(0)
    u = np.random.rand(64, 128).astype(np.float32)
    # (c) coarse only, 32 samples (config 1 shape)
    rc = run_render(kw_test, False, N_samples=32, N_importance=0)
    np.savez(os.path.join(OUT, 'g7_render.npz'), ro=ro.numpy(), rd=rd.numpy(), t_rand=t_rand, u=u,
             **{'a.' + k: v for k, v in ra.items()}, **{'b.' + k: v for k, v in rb.items()},
             **{'c.' + k: v for k, v in rc.items()})

    # ---- G8 one full train step -------------------------------------------
    sel = torch.randint(0, 800 * 800, (256,), generator=g)
    ro8 = o_b.reshape(-1, 3)[sel].contiguous()
    rd8 = d_b.reshape(-1, 3)[sel].contiguous()
    tgt = torch.rand(256, 3, generator=g)
    # the sample depths of both passes are locals of the reference's render_rays (render.py:244-283): recorded through the
    # z_vals argument of its two raw2outputs calls, so that the fine pass can be replayed at the reference's own positions
    seen_z = []
    orig_r2o = R.raw2outputs

    def recording_r2o(raw, z_vals, *a, **k):
        seen_z.append(z_vals.detach().numpy().copy())
        return orig_r2o(raw, z_vals, *a, **k)
    R.raw2outputs = recording_r2o
    try:
        rgb, disp, acc, ex = R.render(800, 800, K, chunk=32768, rays=torch.stack([ro8, rd8], 0), retraw=True,
                                      near=2.0, far=6.0, pytest=True, **kw_train)
    finally:
        R.raw2outputs = orig_r2o
    assert len(seen_z) == 2 and seen_z[0].shape == (256, 64) and seen_z[1].shape == (256, 192)
    optim.zero_grad()
    l1 = H.img2mse(rgb, tgt)
    l0 = H.img2mse(ex['rgb0'], tgt)
    (l1 + l0).backward()
    names = [('c.' + n.replace('module.', '')) for n, _ in kw_train['network_fn'].named_parameters()] + \
            [('f.' + n.replace('module.', '')) for n, _ in kw_train['network_fine'].named_parameters()]
    grads = {'grad.' + n: p.grad.detach().numpy().copy() for n, p in zip(names, grad_vars)}
    optim.step()
    post = {'post.' + n: p.detach().numpy().copy() for n, p in zip(names, grad_vars)
            if n.endswith('bias') or 'pts_linears.0.' in n or 'rgb_linear' in n or 'alpha_linear' in n}
    np.random.seed(0)
    t8 = np.random.rand(256, 64).astype(np.float32)
    np.random.seed(0)
    u8 = np.random.rand(256, 128).astype(np.float32)
    new_lr = 5e-4 * (0.1 ** (0 / (500 * 1000)))
    np.savez(os.path.join(OUT, 'g8_train_step.npz'), ro=ro8.numpy(), rd=rd8.numpy(), target=tgt.numpy(), t_rand=t8, u=u8,
             loss=float(l1), loss0=float(l0), psnr=float(H.mse2psnr(l1.detach())[0]), rgb=rgb.detach().numpy(),
             rgb0=ex['rgb0'].detach().numpy(), new_lr=new_lr, z0=seen_z[0], z_vals=seen_z[1],
             raw=ex['raw'].detach().numpy(), **grads, **post)

    # ---- G9 quadtree -------------------------------------------------------
    tree_out = {}
    for (Hh, Ww) in ((800, 800), (378, 504), (756, 1008), (64, 64)):
        img = np.zeros((Hh, Ww, 3), dtype=np.float32)
        for depth in range(1, 8):
            qt = T.QuadTree(img, 0.0, depth)
            ch = T.get_children(qt.root)
            tree_out[f'leaves_{Hh}x{Ww}_d{depth}'] = np.array([[c.x0, c.y0, c.x1, c.y1] for c in ch], dtype=np.float64)
            tree_out[f'minarea_{Hh}x{Ww}_d{depth}'] = np.float64(qt.minArea)
    np.savez(os.path.join(OUT, 'g9_tree_leaves.npz'), **tree_out)

    # seeded gen + adjust sequence, max criterion (nerf-ours)
    for (Hh, Ww, nimg, d0) in ((64, 64, 3, 2), (100, 76, 2, 2)):
        Kt = np.array([[50.0, 0, Ww / 2], [0, 50.0, Hh / 2], [0, 0, 1]])
        imgs = torch.rand(nimg, Hh, Ww, 3, generator=g)
        poses = torch.stack([pose_spherical_np(40.0 * i, -30.0, 4.0)[:3, :4] for i in range(nimg)], 0)
        mgr = T.QuadTreeManager(Hh, Ww, Kt, imgs, poses, mseThres=0.0, max_depth=d0)
        rec = {'images': imgs.numpy(), 'poses': poses.numpy(), 'K': Kt, 'depth0': d0}
        for rnd in range(5):
            torch.manual_seed(100 + rnd)
            o, d, rgbt = mgr.gen_rays_v3_multiThread(down_scale=1, prob=False, randSamp_proc=1.0, last_epoch=False)
            rec[f'r{rnd}_leaf_id'] = mgr.result_leaf_id.numpy().copy()
            if rnd == 0:
                rec['r0_d'] = d.numpy().copy()
                rec['r0_o'] = o.numpy().copy()
            rec[f'r{rnd}_rgb'] = rgbt.numpy().copy()
            for ti in range(nimg):
                rec[f'r{rnd}_before_t{ti}'] = np.array([[c.x0, c.y0, c.x1, c.y1] for c in mgr.childrens[ti]], dtype=np.float64)
            pred = torch.clamp(rgbt + (torch.rand(rgbt.shape, generator=g) - 0.5) * 0.12 *
                               (torch.rand(rgbt.shape[0], 1, generator=g) < 0.02).float(), 0, 1)
            rec[f'r{rnd}_pred'] = pred.numpy().copy()
            mgr.adjust_tree_multiThread(rgbt, pred, thres=0.03)
            for ti in range(nimg):
                rec[f'r{rnd}_after_t{ti}'] = np.array([[c.x0, c.y0, c.x1, c.y1] for c in mgr.childrens[ti]], dtype=np.float64)
                rec[f'r{rnd}_minarea_t{ti}'] = np.float64(mgr.quadTrees[ti].minArea)
        # last epoch draw
        torch.manual_seed(999)
        o, d, rgbt = mgr.gen_rays_v3_multiThread(down_scale=1, prob=False, last_epoch=True)
        rec['last_leaf_id'] = mgr.result_leaf_id.numpy().copy()
        rec['last_rgb'] = rgbt.numpy().copy()
        rec['last_d'] = d.numpy().copy()
        np.savez_compressed(os.path.join(OUT, f'g9_tree_seq_{Hh}x{Ww}.npz'), **rec)

    # ---- G11 LLFF-style forward-facing render: ndc=True (render() default), near 0 / far 1,
    #      64+64 samples, raw_noise_std=1 through the pytest hook (np.random.rand * std) ----------
    # the G8 block above has stepped the optimiser: restore the weights saved in g7_weights.npz
    kw_train['network_fn'].load_state_dict({'module.' + k: torch.from_numpy(v) for k, v in sdc.items()})
    kw_train['network_fine'].load_state_dict({'module.' + k: torch.from_numpy(v) for k, v in sdf.items()})
    Hl, Wl, fl = 756, 1008, 815.13
    Kl = np.array([[fl, 0, 0.5 * Wl], [0, fl, 0.5 * Hl], [0, 0, 1]])
    c2w_l = torch.tensor([[1.0, 0, 0, 0.05], [0, 1.0, 0, -0.03], [0, 0, 1.0, 0.1]])
    o_l, d_l = H.get_rays(Hl, Wl, Kl, c2w_l)
    sel = torch.randint(0, Hl * Wl, (48,), generator=g)
    rol = o_l.reshape(-1, 3)[sel].contiguous()
    rdl = d_l.reshape(-1, 3)[sel].contiguous()
    kl = {k: v for k, v in kw_train.items() if k not in ('ndc', 'lindisp')}
    kl['raw_noise_std'] = 1.0
    kl['N_importance'] = 64
    rgb, disp, acc, ex = R.render(Hl, Wl, Kl, chunk=32768, rays=torch.stack([rol, rdl], 0), retraw=True,
                                  near=0., far=1., pytest=True, **kl)
    np.savez(os.path.join(OUT, 'g11_llff_render.npz'), ro=rol.numpy(), rd=rdl.numpy(), K=Kl, H=Hl, W=Wl,
             rgb=rgb.detach().numpy(), disp=disp.detach().numpy(), acc=acc.detach().numpy(),
             rgb0=ex['rgb0'].detach().numpy(), acc0=ex['acc0'].detach().numpy(), z_std=ex['z_std'].detach().numpy())

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('wrote goldens to', OUT, 'total bytes', tot)


if __name__ == '__main__':
    main()
