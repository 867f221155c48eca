"""End-to-end parity: render() vs the reference's outputs (G7), one full training step vs the
reference (G8) through both the fused Trainer and the autograd-compatible route, and
size-independent properties at the BASELINE size (4096 rays, 64+128 samples)."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

TOL_RGB = 1e-4   # north_star: rendered RGB within 1e-4 of the reference on identical inputs


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def build(fn, golden_dir, **over):
    kw = dict(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, use_viewdirs=True, no_reload=True,
              lrate=5e-4, lrate_decay=500)
    kw.update(over)
    args = fn.run_nerf.make_args(**kw)
    ktr, kte, _, _, grad_vars, optim = fn.run_nerf.create_nerf(args)
    g = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    ktr['network_fn'].load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('c.')})
    if ktr['network_fine'] is not None:
        ktr['network_fine'].load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('f.')})
    return ktr, kte, grad_vars, optim


def test_render_g7(fn, golden_dir, math_mode):
    g = np.load(os.path.join(golden_dir, 'g7_render.npz'))
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    ktr, kte, _, _ = build(fn, golden_dir)
    rays = torch.stack([torch.from_numpy(g['ro']), torch.from_numpy(g['rd'])], 0).cuda()
    with torch.no_grad():
        for tag, kw, extra in (('a', kte, {}), ('b', ktr, {'pytest': True}),
                               ('c', kte, {'N_samples': 32, 'N_importance': 0})):
            kk = dict(kw); kk.update(extra)
            rgb, disp, acc, ex = fn.render.render(800, 800, K, chunk=32768, rays=rays, retraw=True, near=2.0, far=6.0, **kk)
            res = {'rgb': rgb, 'disp': disp, 'acc': acc}; res.update(ex)
            for k in ('rgb', 'acc', 'rgb0', 'acc0', 'raw', 'z_std', 'disp', 'disp0'):
                if f'{tag}.{k}' not in g.files:
                    continue
                ref = g[f'{tag}.{k}']
                err = np.abs(res[k].cpu().numpy() - ref)
                # rgb/acc: the north_star bound.  raw at the FINE samples is input-sensitivity bound:
                # a 1e-6 shift of a sample position moves sin(2^9 x) by 5e-4, so per-sample logits
                # agree to ~1e-3 while the composited colour agrees to ~1e-5.
                tol = TOL_RGB if k in ('rgb', 'rgb0', 'acc', 'acc0') else 5e-2 * max(1e-3, np.abs(ref).max())
                assert err.max() < tol, (tag, k, err.max())
    assert set(ex.keys()) == {'raw'} and rgb.shape == (64, 3)


def test_train_step_g8_fused(fn, golden_dir, math_mode):
    g = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    ktr, _, _, _ = build(fn, golden_dir)
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    ro, rd, tgt = (torch.from_numpy(g[k]).cuda() for k in ('ro', 'rd', 'target'))
    loss2, out = tr.forward_backward(ro, rd, tgt, t_rand=torch.from_numpy(g['t_rand']).cuda(),
                                     u=torch.from_numpy(g['u']).cuda())
    assert abs(float(loss2[0]) - float(g['loss'])) < 1e-5 and abs(float(loss2[1]) - float(g['loss0'])) < 1e-5
    assert np.abs(out['rgb_map'].cpu().numpy() - g['rgb']).max() < TOL_RGB
    assert np.abs(out['rgb0'].cpu().numpy() - g['rgb0']).max() < TOL_RGB
    grad = tr.grad.cpu()
    names = ['c.' + n for n, _ in O.nerf_param_shapes()] + ['f.' + n for n, _ in O.nerf_param_shapes()]
    shapes = [s for _, s in O.nerf_param_shapes()] * 2
    # (1) vs the reference's gradients.  Bound: 3e-2 of each tensor's max -- dominated by the
    # sensitivity of the fine pass to 1-ulp differences in the inverse-CDF sample positions (the
    # same kernels agree to 5e-5 with the oracle evaluated at identical positions, see (2)).
    off = 0
    for n, shp in zip(names, shapes):
        ref = g['grad.' + n]
        k = ref.size
        got = grad[off:off + k].view(shp).numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() < 3e-2 * scale, (n, np.abs(got - ref).max(), scale)
        off += k
    # (2) kernel exactness: oracle autograd evaluated at the device's own sample positions.  The
    # remaining differences are single ReLU-mask flips (a pre-activation within 1e-7 of zero takes
    # the other branch: ~1 per 1e7 activations, tools/dbg2.py), each worth one sample's contribution
    # to a small-gradient tensor -> bound 1e-2 of the tensor max, and 1e-3 in relative L2.
    wts = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    rb = O.make_ray_batch(ro.cpu(), rd.cpu(), 2.0, 6.0)
    for pre, zkey, lo in (('c.', 'z0', 0), ('f.', 'z_vals', fn.ops.NET_PARAMS)):
        sd = {k[2:]: torch.from_numpy(wts[k]).clone().requires_grad_(True) for k in wts.files if k.startswith(pre)}
        zz = out[zkey].cpu()
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * zz[..., None]
        raw = O.run_network(sd, pts, rb[:, 8:11])
        rgbm = O.raw2outputs(raw, zz, rb[:, 3:6], None, True)[0]
        gr = torch.autograd.grad(O.img2mse(rgbm, tgt.cpu()), list(sd.values()))
        off = lo
        for (n, shp), gg in zip(O.nerf_param_shapes(), gr):
            k = gg.numel()
            got = grad[off:off + k].view(shp)
            assert (got - gg).abs().max() < 1e-2 * max(gg.abs().max().item(), 1e-7), (pre + n)
            assert (got - gg).norm() < 2e-3 * gg.norm() + 1e-12, (pre + n, float((got - gg).norm() / gg.norm()))
            off += k
    # optimiser: feed the REFERENCE's gradients to the Adam kernel -> the reference's post-step
    # weights (identical inputs => tight bound; on the device's own gradients Adam's first step
    # lr*g/(|g|+1e-8) amplifies 1e-9 absolute noise on |g|~1e-8 entries to O(lr))
    gold = torch.cat([torch.from_numpy(g['grad.' + n]).reshape(-1) for n in names]).cuda()
    before = tr.flat.clone()
    tr.grad.copy_(gold)
    fn.ops.adam_step(tr.flat, tr.grad, tr.m, tr.v, 5e-4, 1)
    flat = tr.flat.cpu()
    sl = {}
    off = 0
    for n, shp in zip(names, shapes):
        k = int(np.prod(shp)); sl[n] = flat[off:off + k].view(shp).numpy(); off += k
    n_checked = 0
    for key in g.files:
        if key.startswith('post.'):
            assert np.abs(sl[key[5:]] - g[key]).max() < 2e-7, key
            n_checked += 1
    assert n_checked > 20
    # full fused step (own gradients): bounded update, LR rule with the pre-increment iteration
    tr.flat.copy_(before); tr.m.zero_(); tr.v.zero_(); tr.repack()
    loss2, _ = tr.step(ro, rd, tgt, t_rand=torch.from_numpy(g['t_rand']).cuda(), u=torch.from_numpy(g['u']).cuda())
    upd = (tr.flat - before).abs()
    assert float(upd.max()) <= 5e-4 * 1.001 and float((upd > 4.9e-4).float().mean()) > 0.5
    assert abs(tr.lr - float(g['new_lr'])) < 1e-12 and tr.global_iter == 1 and tr.adam_t == 1


def test_train_step_g8_gradients_against_the_reference(fn, golden_dir, math_mode):
    """G8's gradients against the REFERENCE's at the tight bound (relative L2 <= 2e-3, max <= 1e-2 of each tensor's max).  The
    coarse pass's sample positions are well conditioned (the device's are the reference's to an ulp), so its gradient
    comes straight from the fused step.  The fine pass is replayed at the depths the reference itself used -- its `z_vals`
    (render.py:283), recorded in the golden -- through the per-stage entry points (fastnerf_mlp_fwd with explicit depths ->
    raw2outputs -> mse -> raw2outputs_bwd -> mlp_bwd): same positions, so logits agree to 2e-5 and what is left between the
    gradients are single ReLU-mask flips.  The 3e-2 bound of test_train_step_g8_fused is the price of the
    device's OWN inverse-CDF positions (ill-conditioned in the reference itself), not of the kernels."""
    g = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    ktr, _, _, _ = build(fn, golden_dir)
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    ro, rd, tgt = (torch.from_numpy(g[k]).cuda() for k in ('ro', 'rd', 'target'))
    loss2, out = tr.forward_backward(ro, rd, tgt, t_rand=torch.from_numpy(g['t_rand']).cuda(), u=torch.from_numpy(g['u']).cuda())
    assert np.abs(out['z0'].cpu().numpy() - g['z0']).max() <= 1e-6            # coarse depths: the reference's to an ulp
    zf = out['z_vals'].cpu().numpy()
    moved = np.abs(zf - g['z_vals']) > 2e-5                                   # the fine depths are NOT: most agree,
    # measured (round 4, recorded by this test): 3.58 % of the fine depths move (4.89 % in the narrower bf16x3 mode), none by more than
    # 1.4e-3 (2.0e-3) -- a few sit an inverse-CDF bin away and shift their neighbours' ranks in the sorted list.  Gated at twice that.
    frac, far = float(moved.mean()), float(np.abs(zf - g['z_vals']).max())
    print('G8 fine depths moved by > 2e-5: %.4f of the samples (mode %s), max %.4f' % (frac, math_mode, far))
    lim_frac, lim_far = (0.098, 4.0e-3) if math_mode == 'bf16x3' else (0.072, 2.8e-3)
    assert frac < lim_frac and far < lim_far, (frac, far)
    shapes = O.nerf_param_shapes()

    def check(flat, pre):
        off = 0
        for n, shp in shapes:
            ref = torch.from_numpy(g['grad.' + pre + n])
            k = ref.numel()
            got = flat[off:off + k].view(shp).cpu()
            assert (got - ref).norm() <= 2e-3 * ref.norm() + 1e-12, (pre + n, float((got - ref).norm() / ref.norm()))
            assert (got - ref).abs().max() <= 1e-2 * max(float(ref.abs().max()), 1e-7), (pre + n)
            off += k
        assert off == fn.ops.NET_PARAMS
    check(tr.grad[:fn.ops.NET_PARAMS], 'c.')
    # the fine pass at the reference's depths
    ops = fn.ops
    net = ktr['network_fine']
    pf, pb = net.packed()
    rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.from_numpy(g['z_vals']).cuda()
    P = z.numel()
    act = torch.empty(ops.act_floats(P), device='cuda')
    raw = ops.mlp_fwd(rays11, z, net.flat, pf, act=act)
    ref_raw = torch.from_numpy(g['raw'])
    assert ((raw.cpu() - ref_raw).abs() <= 2e-5 * ref_raw.abs().clamp(min=1.0)).all()
    rgb = ops.raw2outputs_fwd(raw, z, rays11, None, True)[0]
    assert np.abs(rgb.cpu().numpy() - g['rgb']).max() < 1e-5
    l2, grgb, _ = ops.mse_leafmax(rgb, None, tgt)
    assert abs(float(l2[0]) - float(g['loss'])) < 1e-6
    draw = ops.raw2outputs_bwd(raw, z, rays11, grgb, None, True)
    gout = torch.empty(ops.NET_PARAMS, device='cuda')
    ops.mlp_bwd(draw, act, net.flat, pb, torch.empty(ops.dact_floats(P), device='cuda'),
                torch.empty(ops.mlp_bwd_partial_floats(), device='cuda'), gout)
    check(gout, 'f.')


def test_train_step_g8_autograd_route(fn, golden_dir):
    """The reference's own loop shape: render(...); loss.backward(); optimizer.step()."""
    g = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    ktr, _, grad_vars, optim = build(fn, golden_dir)
    H = fn.run_nerf_helpers
    rays = torch.stack([torch.from_numpy(g['ro']), torch.from_numpy(g['rd'])], 0).cuda()
    tgt = torch.from_numpy(g['target']).cuda()
    rgb, disp, acc, extras = fn.render.render(800, 800, K, chunk=32768, rays=rays, retraw=True, near=2.0, far=6.0,
                                              pytest=True, **ktr)
    optim.zero_grad()
    img_loss = H.img2mse(rgb, tgt)
    loss = img_loss + H.img2mse(extras['rgb0'], tgt)
    psnr = H.mse2psnr(img_loss.cpu())
    loss.backward()
    assert abs(float(img_loss) - float(g['loss'])) < 1e-5 and abs(float(psnr[0]) - float(g['psnr'])) < 1e-3
    names = ['c.' + n for n, _ in O.nerf_param_shapes()] + ['f.' + n for n, _ in O.nerf_param_shapes()]
    for n, p in zip(names, grad_vars):
        ref = g['grad.' + n]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 3e-2 * scale, n
    optim.step()
    w = dict(zip(names, grad_vars))
    bad = tot = 0
    for key in g.files:
        if key.startswith('post.'):
            d = np.abs(w[key[5:]].detach().cpu().numpy() - g[key]); bad += int((d > 5e-6).sum()); tot += d.size
    # torch.optim.Adam on the device's own gradients: entries with |g| ~ 1e-8 are noise-amplified
    # (see the fused test); the bulk must agree
    assert bad <= tot * 0.15, (bad, tot)


def test_properties_at_baseline_size(fn, golden_dir, math_mode):
    """4096 rays, 64+128 samples (BASELINE config 2): properties that need no oracle run."""
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    ktr, kte, _, _ = build(fn, golden_dir)
    c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
    ro, rd = fn.run_nerf_helpers.get_rays(800, 800, K, c2w)
    gen = torch.Generator().manual_seed(0)
    sel = torch.randint(0, 640000, (4096,), generator=gen).cuda()
    ro, rd = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
    rays = torch.stack([ro, rd], 0)
    with torch.no_grad():
        rgb, disp, acc, ex = fn.render.render(800, 800, K, chunk=32768, rays=rays, retraw=True, near=2.0, far=6.0, **kte)
        # chunking does not change results (render.py:31-32) -- identical kernels per ray
        rgb2, _, acc2, _ = fn.render.render(800, 800, K, chunk=1000, rays=rays, near=2.0, far=6.0, **kte)
    assert torch.equal(rgb, rgb2) and torch.equal(acc, acc2)
    assert torch.isfinite(rgb).all() and (acc >= 0).all() and (acc <= 1 + 1e-5).all()
    assert ex['raw'].shape == (4096, 192, 4)
    # permutation equivariance over rays
    perm = torch.randperm(4096, generator=gen).cuda()
    with torch.no_grad():
        rgb3, _, _, _ = fn.render.render(800, 800, K, chunk=32768, rays=rays[:, perm], near=2.0, far=6.0, **kte)
    assert torch.equal(rgb3, rgb[perm])
    # ---- oracle comparison AT bench scale (one 4096-ray launch = 12 288 fine tiles on the persistent ticket scheduler) ----
    g = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    sdc = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('c.')}
    sdf = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('f.')}
    # (a) 64 of the 4096 rays, spread over the launch incl. its first and last rows: rendered colours vs the oracle
    pick = torch.cat([torch.arange(0, 4096, 67)[:60], torch.tensor([4092, 4093, 4094, 4095])])
    rb = O.make_ray_batch(ro.cpu()[pick], rd.cpu()[pick], 2.0, 6.0)
    ref = O.render_rays(rb, sdc, sdf, 64, 128, white_bkgd=True)           # perturb = 0: no randoms on either side
    assert (rgb.cpu()[pick] - ref['rgb_map']).abs().max().item() < TOL_RGB
    assert (ex['rgb0'].cpu()[pick] - ref['rgb0']).abs().max().item() < TOL_RGB
    # (opacity is not part of the north_star bound; on a nearly transparent ray -- acc 0.07 with these random weights -- it
    # inherits the inverse-CDF sensitivity of the fine sample positions, DESIGN section 5 (i): measured 1.4e-4 in the
    # split-bf16 mode, 7e-7 elsewhere)
    assert (acc.cpu()[pick] - ref['acc_map']).abs().max().item() < 3e-4
    # (b) raw logits of three 64-point tiles from the TAIL of the fine pass's tile schedule (the last tickets drawn), evaluated
    # by the oracle at the device's own sample positions (so the inverse-CDF sensitivity does not enter): kernel-level bound
    rays11 = fn.ops.pack_rays(ro, rd, 2.0, 6.0)
    out, _ = fn.render._forward_core(rays11, kte['network_fn'], kte['network_fine'], 64, 128, False, 0., True, None, None, None,
                                     None, save=False)
    assert torch.equal(out['rgb_map'], rgb)
    zf, rawf = out['z_vals'].cpu(), out['raw'].cpu()
    r11 = rays11.cpu()
    for tile in (12287, 12286, 12285 - 517):
        p = torch.arange(tile * 64, tile * 64 + 64)
        ray, smp = p // 192, p % 192
        pts = r11[ray, 0:3] + r11[ray, 3:6] * zf[ray, smp][:, None]
        want = O.run_network(sdf, pts[:, None, :], r11[ray, 8:11])[:, 0]
        got = rawf[ray, smp]
        assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), tile
    # a training step at full size decreases the loss on a fixed batch
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0)
    tgt = torch.rand(4096, 3, generator=gen).cuda()
    first = None
    for it in range(6):
        loss2, _ = tr.step(ro, rd, tgt)
        first = first if first is not None else float(loss2[0])
    assert torch.isfinite(loss2).all() and float(loss2[0]) < first


def test_render_path_g18(fn, golden_dir, tmp_path, math_mode):
    """The evaluation loop (render.py:94-146) against frames, PSNR and SSIM recorded from the reference: two 6x8 views of
    the G7 nets, test-mode kwargs, ground truth given."""
    g = np.load(os.path.join(golden_dir, 'g18_render_path.npz'))
    _, kte, _, _ = build(fn, golden_dir)
    H, W, focal = (int(g['hwf'][0]), int(g['hwf'][1]), float(g['hwf'][2]))
    d = str(tmp_path)
    rgbs, disps = fn.render.render_path(torch.from_numpy(g['poses']), [H, W, focal], g['K'], 1024, dict(kte, near=2.0, far=6.0),
                                        gt_imgs=g['gt'], savedir=d)
    assert rgbs.shape == (2, H, W, 3) and disps.shape == (2, H, W)
    assert np.abs(rgbs - g['rgbs']).max() < TOL_RGB
    assert np.abs(disps - g['disps']).max() < 1e-3 * np.abs(g['disps']).max()
    assert np.abs(np.array(fn.render.render_path.last_psnrs) - g['psnr']).max() < 2e-3       # dB
    assert np.abs(np.array(fn.render.render_path.last_ssims) - g['ssim']).max() < 1e-4
    txt = open(os.path.join(d, 'results.txt')).read().split()
    assert abs(float(txt[2]) - float(g['mean_psnr'])) < 2e-3 and abs(float(txt[5]) - float(g['mean_ssim'])) < 1e-4
    # the 8-bit frames the reference writes: at most one code value apart (a colour within 1e-4 of a rounding boundary)
    for i in range(2):
        d8 = np.abs(fn.run_nerf_helpers.to8b(rgbs[i]).astype(int) - g['png%d' % i].astype(int))
        assert d8.max() <= 1 and (d8 > 0).mean() < 0.02


def test_quadtree_device_table(fn, golden_dir):
    """Leaf tags + on-device table from a real step == oracle's segmented max; tree adjusts."""
    from oracle import tree_oracle as TO
    K = np.array([[50.0, 0, 32.0], [0, 50.0, 32.0], [0, 0, 1]])
    gen = torch.Generator().manual_seed(4)
    imgs = torch.rand(2, 64, 64, 3, generator=gen)
    poses = torch.stack([O.pose_spherical(40.0 * i, -30.0, 4.0)[:3, :4] for i in range(2)], 0)
    mgr = fn.tree.QuadTreeManager(64, 64, K, imgs, poses, 0.0, 2)
    torch.manual_seed(100)
    ro, rd, rgb = mgr.gen_rays_v3_multiThread(down_scale=1, prob=False)
    omgr = TO.Manager(64, 64, 2, 2)
    torch.manual_seed(100)
    pix = omgr.gen_pixels()
    assert torch.equal(mgr.result_leaf_id, omgr.result_leaf_id) and torch.equal(mgr.result_pix, pix)
    o_ref, d_ref = O.get_rays(64, 64, K, poses[0])
    sel = (pix[:, 0] == 0)
    assert (rd.cpu()[sel] - d_ref[pix[sel, 1], pix[sel, 2]]).abs().max() < 1e-6
    assert torch.equal(rgb.cpu(), imgs[pix[:, 0], pix[:, 1], pix[:, 2]])
    pred = (rgb + (torch.rand(rgb.shape, generator=gen).cuda() - 0.5) * 0.1).clamp(0, 1)
    mgr.adjust_tree_multiThread(rgb, pred, thres=0.04)
    omgr.adjust(rgb.cpu(), pred.cpu(), 0.04)
    for i in range(2):
        assert np.array_equal(mgr.leaves(i), omgr.leaf_array(i))


def test_llff_ndc_noise_g11(fn, golden_dir):
    """Config-4 shape: NDC rays, near 0 / far 1, 64+64 samples, raw_noise_std=1 via the reference's
    pytest hook; kwargs without 'ndc'/'lindisp' keys => render() defaults ndc=True (run_nerf.py:143-147)."""
    g = np.load(os.path.join(golden_dir, 'g11_llff_render.npz'))
    ktr, _, _, _ = build(fn, golden_dir)
    kl = {k: v for k, v in ktr.items() if k not in ('ndc', 'lindisp')}
    kl['raw_noise_std'] = 1.0
    kl['N_importance'] = 64
    rays = torch.stack([torch.from_numpy(g['ro']), torch.from_numpy(g['rd'])], 0).cuda()
    with torch.no_grad():
        rgb, disp, acc, ex = fn.render.render(int(g['H']), int(g['W']), g['K'], chunk=32768, rays=rays, retraw=True,
                                              near=0., far=1., pytest=True, **kl)
    for k, got in (('rgb', rgb), ('acc', acc), ('rgb0', ex['rgb0']), ('acc0', ex['acc0'])):
        assert np.abs(got.cpu().numpy() - g[k]).max() < TOL_RGB, k
    assert np.abs(ex['z_std'].cpu().numpy() - g['z_std']).max() < 1e-3


def test_config1_coarse_only_step(fn, golden_dir, math_mode):
    """BASELINE configs[0]: the 64 x 64 centre crop of frame 0, 1024 rays per batch, 32 coarse samples, N_importance = 0 (one
    net, loss = the single MSE term, run_nerf.py:483-490).  Identical sample depths on both sides (injected jitter), so
    every tensor's gradient is compared with the oracle's autograd tightly, then five fused steps follow the oracle's Adam."""
    ktr, kte, _, _ = build(fn, golden_dir, N_importance=0, N_samples=32)
    assert ktr['network_fine'] is None
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    c2w = fn.synthetic.pose_spherical(-180.0, -30.0, 4.0)[:3, :4]
    ro_all, rd_all = fn.run_nerf_helpers.get_rays(800, 800, K, c2w.cuda())
    gen = torch.Generator().manual_seed(11)
    sel = torch.randperm(64 * 64, generator=gen)[:1024]
    rows, cols = 368 + sel // 64, 368 + sel % 64
    ro, rd = ro_all[rows.cuda(), cols.cuda()].contiguous(), rd_all[rows.cuda(), cols.cuda()].contiguous()
    wts = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    sd = {k[2:]: torch.from_numpy(wts[k]).clone() for k in wts.files if k.startswith('c.')}
    rb = O.make_ray_batch(ro.cpu(), rd.cpu(), 2.0, 6.0)
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    opt = O.Adam(list(sd.values()), lr=5e-4) if hasattr(O, 'Adam') else None
    for it in range(5 if opt is not None else 1):
        tgt = torch.rand(1024, 3, generator=gen)
        t_rand = torch.rand(1024, 32, generator=gen)
        if it == 0:
            loss2, out = tr.forward_backward(ro, rd, tgt.cuda(), t_rand=t_rand.cuda())
            ps = list(sd.values())
            for q in ps:
                q.requires_grad_(True)
            ret = O.render_rays(rb, sd, None, 32, 0, False, True, t_rand, None)
            gr = torch.autograd.grad(O.img2mse(ret['rgb_map'], tgt), ps)
            for q in ps:
                q.requires_grad_(False)
            assert 'rgb0' not in out and float(loss2[1]) == 0.0
            assert abs(float(loss2[0]) - float(O.img2mse(ret['rgb_map'], tgt).detach())) < 1e-5
            assert (out['rgb_map'].cpu() - ret['rgb_map'].detach()).abs().max() < TOL_RGB
            grad = tr.grad.cpu()
            assert grad.numel() == fn.ops.NET_PARAMS     # one net
            off = 0
            for (n, shp), gg in zip(O.nerf_param_shapes(), gr):
                got = grad[off:off + gg.numel()].view(shp)
                assert (got - gg).norm() < 2e-3 * gg.norm() + 1e-12, (n, float((got - gg).norm() / gg.norm()))
                off += gg.numel()
        if opt is not None:
            l_dev, _ = tr.step(ro, rd, tgt.cuda(), t_rand=t_rand.cuda(), decay=False)
            l_ref, l0, _, _ = O.train_step(sd, None, opt, rb, tgt, 32, 0, True, t_rand, None)
            assert l0 is None and abs(float(l_dev[0]) - float(l_ref)) < 2e-5, (it, float(l_dev[0]), float(l_ref))
    if opt is not None:
        flat = torch.cat([v.reshape(-1) for v in sd.values()])
        d = (tr.flat.cpu() - flat).abs()
        # Adam's first steps move every weight by ~lr whatever the gradient's size: entries whose gradient is rounding noise
        # may differ by O(lr); the bulk must agree
        assert float((d > 2e-5).float().mean()) < 0.02, float((d > 2e-5).float().mean())


def test_no_view_directions_g19(fn, golden_dir, math_mode):
    """use_viewdirs=False end to end on the product surface (create_nerf -> render -> loss.backward() -> fused Trainer):
    reference state_dict names / order, [N,8] ray batches, G19's renders, loss and gradients; the fused step's gradients are
    the autograd route's, and the oracle evaluated at the device's own sample depths pins every tensor."""
    from conftest import noview_state_dicts
    g = np.load(os.path.join(golden_dir, 'g19_noview.npz'))
    args = fn.run_nerf.make_args(N_importance=32, N_samples=32, perturb=1.0, white_bkgd=True, use_viewdirs=False,
                                 no_reload=True, lrate=5e-4, lrate_decay=500)
    ktr, kte, _, _, grad_vars, optim = fn.run_nerf.create_nerf(args)
    sds = noview_state_dicts(golden_dir)
    for key, sd in zip(('network_fn', 'network_fine'), sds):
        net = ktr[key]
        assert list(net.state_dict().keys()) == list(sd.keys())
        assert [tuple(v.shape) for v in net.state_dict().values()] == [v.shape for v in sd.values()]
        net.load_state_dict({'module.' + k: torch.from_numpy(v) for k, v in sd.items()})   # a DataParallel-style checkpoint
    K = g['K']
    H = fn.run_nerf_helpers
    rays = torch.stack([torch.from_numpy(g['ro']), torch.from_numpy(g['rd'])], 0).cuda()
    tgt = torch.from_numpy(g['target']).cuda()

    def close(got, want, tol, what):
        got = got.detach().cpu().numpy()
        nan = np.isnan(want)               # disp of a ray that hits nothing: 0 / 0 on both sides (render.py:186)
        assert np.array_equal(np.isnan(got), nan), what
        assert np.abs(got[~nan] - want[~nan]).max() < tol, (what, np.abs(got[~nan] - want[~nan]).max())

    with torch.no_grad():
        rgb, disp, acc, ex = fn.render.render(800, 800, K, chunk=32768, rays=rays, near=2.0, far=6.0, **kte)
    close(rgb, g['test_rgb'], TOL_RGB, 'test rgb'); close(acc, g['test_acc'], TOL_RGB, 'test acc')
    close(disp, g['test_disp'], 5e-3 * np.nanmax(np.abs(g['test_disp'])), 'test disp')
    # train mode through the reference's pytest hook; loss.backward() as the reference's loop does
    rgb, disp, acc, ex = fn.render.render(800, 800, K, chunk=32768, rays=rays, retraw=True, near=2.0, far=6.0, pytest=True, **ktr)
    close(rgb, g['rgb'], TOL_RGB, 'rgb'); close(ex['rgb0'], g['rgb0'], TOL_RGB, 'rgb0')
    close(acc, g['acc'], TOL_RGB, 'acc'); close(ex['acc0'], g['acc0'], TOL_RGB, 'acc0')
    assert ex['raw'].shape == (64, 64, 4)      # the reference's fifth channel is never read (render.py:169-171)
    close(ex['raw'], g['raw'][..., :4], 5e-2 * np.abs(g['raw'][..., :4]).max(), 'raw')
    optim.zero_grad()
    l1, l0 = H.img2mse(rgb, tgt), H.img2mse(ex['rgb0'], tgt)
    (l1 + l0).backward()
    assert abs(float(l1.detach()) - float(g['loss'])) < 1e-5 and abs(float(l0.detach()) - float(g['loss0'])) < 1e-5
    names = [pre + k for pre, sd in zip(('c.', 'f.'), sds) for k in sd.keys()]
    assert len(names) == len(grad_vars)
    ag = {}
    for n, p in zip(names, grad_vars):
        ag[n] = p.grad.detach().clone()
        if 'views_linears' in n:
            assert float(p.grad.abs().max()) == 0.0            # unused by this model (None in the reference)
        elif 'grad.' + n in g.files:
            ref = g['grad.' + n]
            assert np.abs(ag[n].cpu().numpy() - ref).max() < 3e-2 * max(np.abs(ref).max(), 1e-6), n
    assert float(ag['c.output_linear.weight'][4].abs().max()) == 0.0 and float(ag['c.output_linear.weight'][3].abs().max()) > 0.0
    # fused Trainer on the same batch and draws: the same gradients
    np.random.seed(0); t_rand = torch.from_numpy(np.random.rand(64, 32).astype(np.float32)).cuda()
    np.random.seed(0); u = torch.from_numpy(np.random.rand(64, 32).astype(np.float32)).cuda()
    assert np.array_equal(t_rand.cpu().numpy(), g['t_rand'])
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    loss2, out = tr.forward_backward(rays[0], rays[1], tgt, t_rand=t_rand, u=u)
    assert abs(float(loss2[0]) - float(g['loss'])) < 1e-5 and abs(float(loss2[1]) - float(g['loss0'])) < 1e-5
    offs = np.cumsum([0] + [p.numel() for p in grad_vars[:-1]])      # parameters() order == the flat buffers' order
    assert int(offs[-1]) + grad_vars[-1].numel() == tr.grad.numel() == tr.flat.numel()
    fused = {n: tr.grad[o:o + p.numel()].view(p.shape).clone() for n, p, o in zip(names, grad_vars, offs)}
    for n in names:
        assert (fused[n] - ag[n]).abs().max() <= 1e-5 * max(float(ag[n].abs().max()), 1e-8), n
    # oracle autograd at the device's own depths: every used tensor within 2e-3 relative L2
    rb = O.make_ray_batch(rays[0].cpu(), rays[1].cpu(), 2.0, 6.0, use_viewdirs=False)
    for pre, sd_np, zkey in (('c.', sds[0], 'z0'), ('f.', sds[1], 'z_vals')):
        sd = {k: torch.from_numpy(v).clone() for k, v in sd_np.items()}
        used = [k for k in sd if not k.startswith('views_linears.')]
        for k in used:
            sd[k].requires_grad_(True)
        zz = out[zkey].cpu()
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * zz[..., None]
        rgbm = O.raw2outputs(O.run_network(sd, pts, None), zz, rb[:, 3:6], None, True)[0]
        gr = torch.autograd.grad(O.img2mse(rgbm, tgt.cpu()), [sd[k] for k in used])
        for k, gg in zip(used, gr):
            got = fused[pre + k].cpu()
            assert (got - gg).norm() < 2e-3 * gg.norm() + 1e-12, (pre + k, float((got - gg).norm() / (gg.norm() + 1e-12)))
    # one fused optimisation step moves the reference's parameters (and only the used ones)
    before = tr.flat.clone()
    tr.step(rays[0], rays[1], tgt, t_rand=t_rand, u=u)
    moved = {n: float((p.detach() - before[o:o + p.numel()].view(p.shape)).abs().max()) for n, p, o in zip(names, grad_vars, offs)}
    assert moved['c.views_linears.0.weight'] == 0.0 and 0.0 < moved['c.output_linear.weight'] <= 5e-4 * 1.001
    assert moved['f.pts_linears.3.weight'] > 0.0
    # the kernels' network follows the parameters: a fresh render uses the updated weights
    with torch.no_grad():
        rgb2 = fn.render.render(800, 800, K, chunk=32768, rays=rays, near=2.0, far=6.0, **kte)[0]
    assert float((rgb2.cpu() - torch.from_numpy(g['test_rgb'])).abs().max()) > 1e-4


@pytest.mark.parametrize('lindisp,white', [(False, True), (True, False)])
def test_shared_network_both_passes(fn, golden_dir, math_mode, lindisp, white):
    """N_importance > 0 with network_fine=None: ONE net serves both passes (render.py:288 `run_fn = network_fn if
    network_fine is None else network_fine`) and collects the gradients of both loss terms.  Also the only end-to-end case
    with lindisp=True / a black background.  Checked against the oracle's autograd at the device's own sample depths."""
    ktr, kte, _, _ = build(fn, golden_dir)
    net = ktr['network_fn']
    g = np.load(os.path.join(golden_dir, 'g8_train_step.npz'))
    n = 96
    ro, rd, tgt = (torch.from_numpy(g[k][:n]).cuda() for k in ('ro', 'rd', 'target'))
    gen = torch.Generator().manual_seed(5)
    t_rand, u = torch.rand(n, 24, generator=gen).cuda(), torch.rand(n, 40, generator=gen).cuda()
    rays11 = fn.ops.pack_rays(ro, rd, 2.0, 6.0)
    out, saved = fn.render._forward_core(rays11, net, None, 24, 40, lindisp, 1.0, white, t_rand, u, None, None, save=True)
    assert saved['net_f'] is net
    loss2, g1, g0 = fn.ops.mse_leafmax(out['rgb_map'], out['rgb0'], tgt)
    grads = torch.zeros_like(net.flat)
    fn.render._backward_core(saved, g1, g0, out_c=grads)
    # oracle: both passes through the same weights, at the device's depths
    wts = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    sd = {k[2:]: torch.from_numpy(wts[k]).clone().requires_grad_(True) for k in wts.files if k.startswith('c.')}
    rb = O.make_ray_batch(ro.cpu(), rd.cpu(), 2.0, 6.0)
    total = 0.
    for zkey, okey in (('z0', 'rgb0'), ('z_vals', 'rgb_map')):
        zz = out[zkey].cpu()
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * zz[..., None]
        rgbm = O.raw2outputs(O.run_network(sd, pts, rb[:, 8:11]), zz, rb[:, 3:6], None, white)[0]
        assert (out[okey].cpu() - rgbm.detach()).abs().max() < TOL_RGB, okey
        total = total + O.img2mse(rgbm, tgt.cpu())
    z_ref = O.coarse_z(rb[:, 6:7], rb[:, 7:8], 24, lindisp, t_rand.cpu())
    assert (out['z0'].cpu() - z_ref).abs().max() < 2e-6
    gr = torch.autograd.grad(total, list(sd.values()))
    off, worst = 0, 0.0
    for (name, shp), gg in zip(O.nerf_param_shapes(), gr):
        got = grads[off:off + gg.numel()].view(shp).cpu()
        worst = max(worst, float((got - gg).norm() / (gg.norm() + 1e-12)))
        off += gg.numel()
    # (96 rays at random init: the gradient is a badly conditioned difference of nearly equal colours -- measured 2.2e-3 in the
    # split-bf16 mode, whose unit roundoff is 2^-17, and < 1e-3 in the fp32 mode; cf. the nerf++ step test)
    assert worst < (2e-3 if math_mode == 'fp32' else 8e-3), worst
    # the public route: render_rays(..., network_fine=None) with autograd gives the same gradients
    for p_ in net.parameters():
        p_.grad = None
    ret = fn.render.render_rays(rays11, net, None, 24, N_importance=40, network_fine=None, perturb=1.0, lindisp=lindisp,
                                white_bkgd=white, pytest=True)
    assert {'rgb_map', 'rgb0', 'z_std'} <= set(ret.keys())
    H = fn.run_nerf_helpers
    (H.img2mse(ret['rgb_map'], tgt) + H.img2mse(ret['rgb0'], tgt)).backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in net.parameters())


def test_full_chunk_equals_small_chunks(fn, golden_dir, math_mode):
    """render()'s default chunk (32 768 rays = 6.3 M fine-pass points = 98 304 tiles in one launch, the size render_path and the
    bench's inference leg use) gives bit for bit what eight launches of 4096 rays give."""
    _, kte, _, _ = build(fn, golden_dir)
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    gen = torch.Generator().manual_seed(77)
    n = 32768
    poses = torch.stack([fn.synthetic.pose_spherical(-180.0 + 36.0 * k, -30.0, 4.0)[:3, :4] for k in range(10)], 0).cuda()
    pix = torch.stack([torch.randint(0, 10, (n,), generator=gen), torch.randint(0, 800, (n,), generator=gen),
                       torch.randint(0, 800, (n,), generator=gen)], 1).int().cuda()
    ro, rd = fn.ops.gen_rays_pixels(pix, poses, K)
    with torch.no_grad():
        big = fn.render.render(800, 800, K, chunk=n, rays=(ro, rd), near=2.0, far=6.0, **kte)
        small = fn.render.render(800, 800, K, chunk=4096, rays=(ro, rd), near=2.0, far=6.0, **kte)
    for a, b in zip(big[:3], small[:3]):
        assert a.shape[0] == n and torch.equal(a, b)
    assert torch.equal(big[3]['z_std'], small[3]['z_std']) and torch.isfinite(big[0]).all()


def test_batch_beyond_int32_offsets(fn, golden_dir, math_mode):
    """8192 rays per step: the fine pass saves 1 572 864 points x 2528 floats = 3.98e9 elements, past 2^31 (the bench's 4096
    rays stay just below).  Its gradient must be the sum of the gradients of the two halves (each scaled for the global
    batch), and its colours the halves' colours bit for bit."""
    ktr, _, _, _ = build(fn, golden_dir)
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    gen = torch.Generator().manual_seed(31)
    n = 8192
    poses = torch.stack([fn.synthetic.pose_spherical(-180.0 + 36.0 * k, -30.0, 4.0)[:3, :4] for k in range(10)], 0).cuda()
    pix = torch.stack([torch.randint(0, 10, (n,), generator=gen), torch.randint(0, 800, (n,), generator=gen),
                       torch.randint(0, 800, (n,), generator=gen)], 1).int().cuda()
    ro, rd = fn.ops.gen_rays_pixels(pix, poses, K)
    tgt = torch.rand(n, 3, generator=gen).cuda()
    t_rand, u = torch.rand(n, 64, generator=gen).cuda(), torch.rand(n, 128, generator=gen).cuda()
    old = fn.render.get_compact()
    try:
        for mode in ('0', '1'):       # plain backward (saves every point) and compacted backward
            fn.render.set_compact(mode)
            tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0)
            loss_full, out_full = tr.forward_backward(ro, rd, tgt, t_rand=t_rand, u=u)
            g_full = tr.grad.clone()
            rgb_full = out_full['rgb_map'].clone()
            g_sum = torch.zeros_like(g_full)
            for h in range(2):
                sl = slice(h * 4096, (h + 1) * 4096)
                _, out_h = tr.forward_backward(ro[sl].contiguous(), rd[sl].contiguous(), tgt[sl].contiguous(),
                                               t_rand=t_rand[sl].contiguous(), u=u[sl].contiguous(), n_global=n)
                g_sum += tr.grad
                assert torch.equal(out_h['rgb_map'], rgb_full[sl]), (mode, h)
            rel = float((g_full - g_sum).norm() / g_sum.norm())
            assert torch.isfinite(g_full).all() and rel < 2e-5, (mode, rel)
            del tr, out_full
    finally:
        fn.render.set_compact(old)
