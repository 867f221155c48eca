"""Randomised end-to-end configurations of render_rays on the HIP path against the oracle (which oracle/fuzz_vs_reference.py
holds bit-identical to the reference over the same configuration space): odd sample counts, one importance sample, lindisp,
black / white background, sigma noise, NDC rays, no view directions, one shared net for both passes."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def _configs():
    rng = np.random.RandomState(3)
    out = []
    for ci in range(14):
        use_viewdirs = bool(rng.rand() < 0.7)
        Ns = int(rng.choice([2, 3, 8, 17, 32, 64, 65]))
        Ni = int(rng.choice([0, 1, 5, 16, 33, 64, 127])) if Ns >= 3 else 0
        ndc = bool(rng.rand() < 0.3)
        lindisp = bool(rng.rand() < 0.4) and not ndc
        out.append(dict(ci=ci, use_viewdirs=use_viewdirs, Ns=Ns, Ni=Ni, ndc=ndc, lindisp=lindisp, white=bool(rng.rand() < 0.5),
                        perturb=float(rng.choice([0.0, 1.0])), noise=float(rng.choice([0.0, 0.0, 1.0])),
                        shared=bool(Ni > 0 and rng.rand() < 0.25)))
    return out


@pytest.mark.parametrize('cfg', _configs(), ids=lambda c: 'v{use_viewdirs:d}-{Ns}+{Ni}-ndc{ndc:d}-lin{lindisp:d}-sh{shared:d}'.format(**c))
def test_random_configuration(fn, golden_dir, math_mode, cfg):
    from conftest import noview_state_dicts
    n = 40
    gen = torch.Generator().manual_seed(50 + cfg['ci'])
    args = fn.run_nerf.make_args(N_importance=cfg['Ni'], N_samples=cfg['Ns'], perturb=cfg['perturb'], white_bkgd=cfg['white'],
                                 use_viewdirs=cfg['use_viewdirs'], no_reload=True, raw_noise_std=cfg['noise'], lindisp=cfg['lindisp'])
    ktr = fn.run_nerf.create_nerf(args)[0]
    wts = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    if cfg['use_viewdirs']:
        sds = [{k[2:]: wts[k] for k in wts.files if k.startswith(pre)} for pre in ('c.', 'f.')]
    else:
        sds = noview_state_dicts(golden_dir)
        if cfg['Ni'] == 0:   # output_ch = 4 without a fine pass
            sds = [{k: (v[:4] if k.startswith('output_linear') else v) for k, v in sd.items()} for sd in sds]
    net_c, net_f = ktr['network_fn'], ktr['network_fine']
    net_c.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sds[0].items()})
    if net_f is not None:
        net_f.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sds[1].items()})
    if cfg['shared']:
        net_f = None
    if cfg['ndc']:
        Hh, Ww, focal, near, far = 378, 504, 400.0, 0.0, 1.0
        ro = torch.rand(n, 3, generator=gen) * 0.2 - 0.1
        rd = torch.cat([(torch.rand(n, 2, generator=gen) - 0.5) * 0.8, -torch.ones(n, 1)], -1)
    else:
        Hh, Ww, focal, near, far = 800, 800, 1111.1, 2.0, 6.0
        g8 = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
        ro, rd = torch.from_numpy(g8['ro'][:n]), torch.from_numpy(g8['rd'][:n])
    tgt = torch.rand(n, 3, generator=gen)
    Ns, Ni = cfg['Ns'], cfg['Ni']
    t_rand = torch.rand(n, Ns, generator=gen) if cfg['perturb'] > 0 else None
    u = torch.rand(n, Ni, generator=gen) if (cfg['perturb'] > 0 and Ni > 0) else None
    n0 = torch.randn(n, Ns, generator=gen) * cfg['noise'] if cfg['noise'] > 0 else None
    n1 = torch.randn(n, Ns + Ni, generator=gen) * cfg['noise'] if (cfg['noise'] > 0 and Ni > 0) else None
    C = lambda t: None if t is None else t.cuda()
    rays11 = fn.ops.pack_rays(ro.cuda(), rd.cuda(), near, far, ndc=cfg['ndc'], H=Hh, W=Ww, focal=focal)
    if not cfg['use_viewdirs']:
        rays11[:, 8:11] = 0.
    out, saved = fn.render._forward_core(rays11, net_c, net_f, Ns, Ni, cfg['lindisp'], cfg['perturb'], cfg['white'], C(t_rand), C(u),
                                         C(n0), C(n1), save=True)
    # ---- forward vs the oracle on the same draws ----
    sd_t = [{k: torch.from_numpy(np.ascontiguousarray(v)).clone() for k, v in sd.items()} for sd in sds]
    rb = O.make_ray_batch(ro, rd, near, far, Hh, Ww, focal, ndc=cfg['ndc'], use_viewdirs=cfg['use_viewdirs'])
    ref = O.render_rays(rb, sd_t[0], None if (cfg['shared'] or Ni == 0) else sd_t[1], Ns, Ni, cfg['lindisp'], cfg['white'], t_rand, u,
                        n0, n1)
    # the coarse pass sees the same depths on both sides (injected jitter): direct comparison
    for a in (('rgb0', 'acc0') if Ni > 0 else ('rgb_map', 'acc_map')):
        assert (out[a].cpu() - ref[a]).abs().max() < 1e-4, (a, float((out[a].cpu() - ref[a]).abs().max()))
    if Ni > 0:
        # the fine depths come out of the inverse CDF, which is discontinuous where u meets a cdf value: a 1-ulp difference in
        # the running sums moves such a sample to the neighbouring bin (with 8 coarse samples that is 2e-4 of colour; found
        # by this test).  So: nearly all depths agree, every one stays inside the coarse range, and the colours are compared
        # with the oracle evaluated AT the device's depths.
        dz = (out['z_vals'].cpu() - ref['z_vals']).abs()
        assert float((dz > 2e-5 * max(1.0, far)).float().mean()) < 0.03, float((dz > 2e-5).float().mean())
        zc = out['z0'].cpu()
        assert bool((out['z_vals'].cpu() >= zc[:, :1] - 1e-6).all()) and bool((out['z_vals'].cpu() <= zc[:, -1:] + 1e-6).all())
        zz = out['z_vals'].cpu()
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * zz[..., None]
        sdx = sd_t[0] if (cfg['shared']) else sd_t[1]
        raw = O.run_network(sdx, pts, rb[:, 8:11] if cfg['use_viewdirs'] else None)
        rgbm, dispm, accm = O.raw2outputs(raw, zz, rb[:, 3:6], n1, cfg['white'])[:3]
        assert (out['rgb_map'].cpu() - rgbm).abs().max() < 1e-4 and (out['acc_map'].cpu() - accm).abs().max() < 1e-4
        assert torch.equal(torch.isnan(out['disp_map']).cpu(), torch.isnan(dispm))
    else:
        assert torch.equal(torch.isnan(out['disp_map']).cpu(), torch.isnan(ref['disp_map']))
    # ---- backward vs the oracle's autograd at the device's own depths ----
    grads_c = torch.zeros_like(net_c.flat)
    grads_f = torch.zeros_like(net_c.flat) if (net_f is not None and Ni > 0) else None
    loss2, g1, g0 = fn.ops.mse_leafmax(out['rgb_map'], out.get('rgb0'), tgt.cuda())
    fn.render._backward_core(saved, g1, g0, out_c=grads_c, out_f=grads_f)
    passes = [('z_vals', 'c', 0, n0)] if Ni == 0 else [('z0', 'c', 0, n0), ('z_vals', 'c' if cfg['shared'] else 'f', 0 if cfg['shared'] else 1, n1)]
    want = {}
    for sd in sd_t:
        for k, v in sd.items():
            if not k.startswith('views_linears.') or cfg['use_viewdirs']:
                v.requires_grad_(True)
    total = {0: 0., 1: 0.}
    for zkey, _, which, nz in passes:
        zz = out[zkey].cpu()
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * zz[..., None]
        raw = O.run_network(sd_t[which], pts, rb[:, 8:11] if cfg['use_viewdirs'] else None)
        total[which] = total[which] + O.img2mse(O.raw2outputs(raw, zz, rb[:, 3:6], nz, cfg['white'])[0], tgt)
    bound = 2e-3 if math_mode == 'fp32' else 1e-2
    for which, (net, got_flat) in enumerate(((net_c, grads_c), (net_f, grads_f))):
        if got_flat is None or not torch.is_tensor(total[which]):
            continue
        used = [k for k, v in sd_t[which].items() if v.requires_grad]
        gr = dict(zip(used, torch.autograd.grad(total[which], [sd_t[which][k] for k in used], allow_unused=True)))
        got = dict(zip([nm for nm, _ in net.named_parameters()], net.param_grads_from(got_flat)))
        num = den = 0.0
        for k in used:
            if gr[k] is None:
                continue
            gg = gr[k][:got[k].shape[0]] if k.startswith('output_linear') else gr[k]
            num += float((got[k].cpu() - gg).pow(2).sum()); den += float(gg.pow(2).sum())
        if den == 0.0:      # every sample of every ray dead (sigma <= 0): no gradient at all, on both sides
            assert num == 0.0, (which, num)
        else:
            assert (num / den) ** 0.5 < bound, (which, (num / den) ** 0.5)


@pytest.mark.parametrize('S0,S1,N', [(3, 5, 7), (8, 5, 24), (17, 33, 12), (64, 33, 5), (32, 128, 24), (65, 1, 9)])
def test_random_cascade_nerfpp(fn, golden_dir, math_mode, S0, S1, N):
    """nerf++ train_step batch with odd cascade sample counts against the oracle (which oracle/fuzz_pp_vs_reference.py holds
    bit-identical to the reference's train_step): level-0 colour directly, every level's colour and gradients at the device's
    own depths."""
    from oracle import nerfpp_oracle as PP
    w = np.load(os.path.join(golden_dir, 'g10_pp_weights.npz'))
    T = lambda a: torch.from_numpy(np.asarray(a))
    levels = [({k[len(f'l{m}.fg_net.'):]: T(w[k]).clone() for k in w.files if k.startswith(f'l{m}.fg_net.')},
               {k[len(f'l{m}.bg_net.'):]: T(w[k]).clone() for k in w.files if k.startswith(f'l{m}.bg_net.')}) for m in range(2)]
    nets = []
    for fg, bg in levels:
        net = fn.nerfpp.NerfNetWithAutoExpo(None)
        sd = {'fg_net.' + k: v for k, v in fg.items()}
        sd.update({'bg_net.' + k: v for k, v in bg.items()})
        net.nerf_net.load_state_dict(sd)
        nets.append(net)
    gen = torch.Generator().manual_seed(S0 * 131 + S1)
    ro = (torch.rand(N, 3, generator=gen) - 0.5) * 0.9
    rd = torch.randn(N, 3, generator=gen)
    tgt = torch.rand(N, 3, generator=gen)
    rand = [{'fg_t': torch.rand(N, S0, generator=gen), 'bg_t': torch.rand(N, S0, generator=gen)},
            {'fg_u': torch.rand(N, S1, generator=gen), 'bg_u': torch.rand(N, S1, generator=gen)}]
    tr = fn.nerfpp.CascadeTrainer(nets, cascade_samples=(S0, S1), lrate=5e-4)
    losses, rgb = tr.step(ro.cuda(), rd.cuda(), tgt.cuda(), rand=[{k: v.cuda() for k, v in r.items()} for r in rand], update=False)
    ref = PP.cascade_step(levels, ro, rd, tgt, [S0, S1], rand)
    assert abs(float(losses[0]) - float(ref[0][0])) < 2e-5
    fg_far = PP.intersect_sphere(ro, rd)
    bound = 2e-3 if math_mode == 'fp32' else 1e-2
    for m in range(2):
        sd_fg, sd_bg = levels[m]
        params = list(sd_fg.values()) + list(sd_bg.values())
        for p in params:
            p.requires_grad_(True)
        fz, bz = (z.cpu() for z in tr.last_depths[m])
        assert fz.shape == (N, S0 if m == 0 else S0 + S1) and bz.shape == fz.shape
        if m == 0:
            assert (fz - ref[0][3]).abs().max() < 2e-6 and (bz - ref[0][4]).abs().max() < 2e-6
        ret = PP.nerfnet_forward(sd_fg, sd_bg, ro, rd, fg_far, fz, bz)
        if m == 1:
            assert (rgb.cpu() - ret['rgb'].detach()).abs().max() < 1e-4
        gr = torch.autograd.grad(torch.mean((ret['rgb'] - tgt) ** 2), params)
        for p in params:
            p.requires_grad_(False)
        flat_g = nets[m].nerf_net.flat_grad.cpu()
        off, it, num, den = 0, iter(gr), 0.0, 0.0
        for kind in (1, 2):
            for name, o, shape in fn.nerfpp.mlpnet_slices(kind):
                gg = next(it)
                got = flat_g[off + o: off + o + gg.numel()].view(gg.shape)
                num += float((got - gg).pow(2).sum()); den += float(gg.pow(2).sum())
            off += fn.ops.net_floats(kind, 0)
        assert (num / den) ** 0.5 < bound, (m, (num / den) ** 0.5)


def test_random_cameras_and_ndc(fn):
    """get_rays / ndc_rays / pack_rays on random cameras and odd image sizes (off-centre principal points, anisotropic
    focal lengths, 1 x 1 and 3 x 7 images, near planes other than 1) against the oracle."""
    gen = torch.Generator().manual_seed(123)
    for case in range(10):
        H = int(torch.randint(1, 70, (1,), generator=gen)) if case else 1
        W = int(torch.randint(1, 90, (1,), generator=gen)) if case else 1
        fx, fy = (float(v) for v in (torch.rand(2, generator=gen) * 900 + 20))
        cx, cy = float(torch.rand(1, generator=gen) * W), float(torch.rand(1, generator=gen) * H)
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=gen))
        c2w = torch.cat([q, torch.randn(3, 1, generator=gen)], 1).float()
        o_ref, d_ref = O.get_rays(H, W, K, c2w)
        o, d = fn.run_nerf_helpers.get_rays(H, W, K, c2w.cuda())
        assert o.shape == (H, W, 3) and torch.equal(o.cpu(), o_ref)
        assert (d.cpu() - d_ref).abs().max() <= 2e-6 * d_ref.abs().max()
        # NDC on forward-facing rays (d_z < 0), any near plane
        n = 64
        ro = torch.rand(n, 3, generator=gen) * 0.4 - 0.2
        rd = torch.cat([torch.rand(n, 2, generator=gen) - 0.5, -torch.rand(n, 1, generator=gen) - 0.2], -1)
        near = float(torch.rand(1, generator=gen) * 2 + 0.1)
        no_ref, nd_ref = O.ndc_rays(H + 10, W + 10, fx, near, ro, rd)
        no, nd = fn.run_nerf_helpers.ndc_rays(H + 10, W + 10, fx, near, ro.cuda(), rd.cuda())
        assert (no.cpu() - no_ref).abs().max() <= 2e-6 * max(1.0, float(no_ref.abs().max()))
        assert (nd.cpu() - nd_ref).abs().max() <= 2e-6 * max(1.0, float(nd_ref.abs().max()))
        for ndc in (False, True):
            rb_ref = O.make_ray_batch(ro, rd, 0.25, 3.5, H + 10, W + 10, fx, ndc=ndc)
            rb = fn.ops.pack_rays(ro.cuda(), rd.cuda(), 0.25, 3.5, ndc=ndc, H=H + 10, W=W + 10, focal=fx)
            assert (rb.cpu() - rb_ref).abs().max() <= 2e-6 * max(1.0, float(rb_ref.abs().max()))
