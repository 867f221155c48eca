"""Exact zero-gradient point compaction of the training backward (csrc/train.hip cp_*, the *_live entry points,
render.cpp fastnerf_render_rays_bwd_live).

What is shown, in this order:
  1. the live list is exactly the ascending list of points with a non-zero d(loss)/d(raw);
  2. a point with a zero d(loss)/d(raw) contributes EXACTLY nothing: the plain backward is bit-invariant to what such
     points are (their inputs are replaced by garbage);
  3. the live-list kernels are bit-identical to the plain kernels run on the batch made of the live points only
     (same tiles, same order) -- including the list of all points, and an empty list;
  => the compacted gradient differs from the plain one only by fp32 summation grouping in dW (bounded below), and is
     compared with the CPU oracle's autograd like the plain one;
  4. whole training steps with and without compaction: identical losses (the forward is the same arithmetic), gradients
     to rounding, trajectories together; counters and the auto policy."""
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


@pytest.fixture(scope='module')
def weights(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    return {k[2:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith('c.')}


def flat_of(sd):
    return torch.cat([sd[n].reshape(-1) for n, _ in O.nerf_param_shapes()])


@pytest.mark.parametrize('P,frac', [(1, 1.0), (1, 0.0), (1023, 0.5), (1024, 0.5), (1025, 0.3), (4096 * 192, 0.45), (70001, 1.0), (70001, 0.0)])
def test_compact_live_list(fn, P, frac):
    gen = torch.Generator().manual_seed(P)
    draw = torch.randn(P, 4, generator=gen)
    dead = torch.rand(P, generator=gen) >= frac
    draw[dead] = 0.0
    draw[dead & (torch.rand(P, generator=gen) < 0.3)] = -0.0          # negative zeros are zeros
    one = (~dead) & (torch.rand(P, generator=gen) < 0.2)              # a single non-zero component keeps a point alive
    draw[one] = 0.0
    draw[one, 3] = 1e-30
    idx, cnt = fn.ops.compact_live(draw.cuda())
    want = torch.nonzero((draw != 0).any(1)).reshape(-1).int()
    c = cnt.cpu().tolist()
    assert c == [want.numel(), P]
    assert torch.equal(idx.cpu()[:c[0]], want)


def _setup(fn, weights, n, S, seed, dead_frac):
    gen = torch.Generator().manual_seed(seed)
    ro = torch.randn(n, 3, generator=gen) * 0.4
    rd = torch.randn(n, 3, generator=gen)
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values
    cot = torch.randn(n, S, 4, generator=gen)
    dead = torch.rand(n, S, generator=gen) < dead_frac
    cot[dead] = 0.0
    flat = flat_of(weights).cuda()
    pf, pb = fn.ops.mlp_pack(flat)
    return rb, z, cot, dead, flat, pf, pb


def _plain(fn, rb, z, cot, flat, pf, pb):
    n, S = z.shape
    act = torch.empty(fn.ops.act_floats(n * S)).cuda()
    fn.ops.mlp_fwd(rb.cuda(), z.cuda(), flat, pf, act=act)
    dact = torch.empty(fn.ops.dact_floats(n * S)).cuda()
    partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
    g = torch.full((fn.ops.NET_PARAMS,), float('nan')).cuda()
    fn.ops.mlp_bwd(cot.cuda(), act, flat, pb, dact, partial, g)
    return g


def _live(fn, rb, z, cot, flat, pf, pb, idx=None, cnt=None):
    n, S = z.shape
    if idx is None:
        idx, cnt = fn.ops.compact_live(cot.cuda())
    act = torch.full((fn.ops.act_floats(n * S),), float('nan')).cuda()      # stale scratch must not matter
    dact = torch.full((fn.ops.dact_floats(n * S),), float('nan')).cuda()
    fn.ops.mlp_fwd_live(rb.cuda(), z.cuda(), flat, pf, act, idx, cnt)
    partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
    g = torch.full((fn.ops.NET_PARAMS,), float('nan')).cuda()
    fn.ops.mlp_bwd_live(cot.cuda(), act, flat, pb, dact, partial, g, idx, cnt)
    return g, idx, cnt


# (1367, 193): 130 k live points -- every dW workgroup runs its steady-state loop over the live list (32 k-steps each, a partial last one)
@pytest.mark.parametrize('n,S,dead_frac', [(9, 50, 0.5), (64, 192, 0.55), (300, 64, 0.97), (7, 9, 0.4), (1367, 193, 0.5)])
def test_dead_points_contribute_exact_zeros_and_live_kernels_are_the_plain_kernels(fn, weights, math_mode, n, S, dead_frac):
    rb, z, cot, dead, flat, pf, pb = _setup(fn, weights, n, S, 11 * n + S, dead_frac)
    g_plain = _plain(fn, rb, z, cot, flat, pf, pb)
    assert torch.isfinite(g_plain).all()
    # (2) garbage at the dead points: bit-identical gradient
    z_bad = z.clone()
    z_bad[dead] = torch.rand(int(dead.sum())) * 40 - 20
    assert torch.equal(_plain(fn, rb, z_bad, cot, flat, pf, pb), g_plain)
    # (3) live kernels == plain kernels on the batch of the live points (one "ray" per point, S = 1)
    g_live, idx, cnt = _live(fn, rb, z, cot, flat, pf, pb)
    k = int(cnt[0])
    assert k == int((~dead).sum()) and torch.isfinite(g_live).all()
    sel = idx[:k].long().cpu()
    rb_g = rb[sel // S].contiguous()
    z_g = z.reshape(-1)[sel].reshape(-1, 1).contiguous()
    cot_g = cot.reshape(-1, 4)[sel].reshape(-1, 1, 4).contiguous()
    assert torch.equal(_plain(fn, rb_g, z_g, cot_g, flat, pf, pb), g_live)
    # => compacted vs plain: only the grouping of fp32 partial sums differs
    scale = g_plain.abs().max().item()
    assert (g_live - g_plain).abs().max().item() < 3e-6 * scale, ((g_live - g_plain).abs().max().item(), scale)
    # and against the oracle's autograd, at the size and with the bound of the plain kernels' test (tests/test_gpu_mlp.py)
    if n * S <= 1000:
        sd = {kk: v.clone().requires_grad_(True) for kk, v in weights.items()}
        pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
        out = O.run_network(sd, pts, rb[:, 8:11])
        ref = torch.cat([t.reshape(-1) for t in torch.autograd.grad((out * cot).sum(), list(sd.values()))])
        off = 0
        for (name, shape) in O.nerf_param_shapes():
            kk = int(np.prod(shape))
            r = ref[off:off + kk]
            err = (g_live.cpu()[off:off + kk] - r).abs().max().item()
            assert err < 2e-5 * max(1.0, r.abs().max().item()), (name, err)
            off += kk


def test_live_list_of_everything_and_of_nothing(fn, weights, math_mode):
    rb, z, cot, dead, flat, pf, pb = _setup(fn, weights, 33, 40, 5, 0.0)
    g_plain = _plain(fn, rb, z, cot, flat, pf, pb)
    g_live, idx, cnt = _live(fn, rb, z, cot, flat, pf, pb)
    assert cnt.cpu().tolist() == [33 * 40, 33 * 40] and torch.equal(g_live, g_plain)
    g_none, _, cnt0 = _live(fn, rb, z, torch.zeros_like(cot), flat, pf, pb)
    assert cnt0.cpu().tolist() == [0, 33 * 40] and torch.count_nonzero(g_none) == 0
    # a list that is a permutation: the kernels follow the list order (results = plain on the permuted batch)
    perm = torch.randperm(33 * 40, generator=torch.Generator().manual_seed(1)).int().cuda()
    g_perm, _, _ = _live(fn, rb, z, cot, flat, pf, pb, idx=perm, cnt=cnt)
    sel = perm.long().cpu()
    g_ref = _plain(fn, rb[sel // 40].contiguous(), z.reshape(-1)[sel].reshape(-1, 1).contiguous(),
                   cot.reshape(-1, 4)[sel].reshape(-1, 1, 4).contiguous(), flat, pf, pb)
    assert torch.equal(g_perm, g_ref)


def _trainer(fn, golden_dir, compact):
    K = np.load(os.path.join(golden_dir, 'g1_get_rays.npz'))['K']
    g = np.load(os.path.join(golden_dir, 'g7_weights.npz'))
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True)
    torch.manual_seed(0)
    ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
    ktr['network_fn'].load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('c.')})
    ktr['network_fine'].load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('f.')})
    fn.render.set_compact(compact)
    return fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0), K


def test_training_steps_with_and_without_compaction(fn, golden_dir, math_mode):
    old = fn.render.get_compact()
    try:
        imgs, poses, focal = fn.synthetic.make_dataset(n_images=2, H=48, W=48)
        K48 = np.array([[focal, 0, 24.0], [0, focal, 24.0], [0, 0, 1]])
        rays = [fn.run_nerf_helpers.get_rays(48, 48, K48, poses[i]) for i in range(2)]
        ro = torch.cat([r[0].reshape(-1, 3) for r in rays], 0)
        rd = torch.cat([r[1].reshape(-1, 3) for r in rays], 0)
        tgt = imgs.reshape(-1, 3).cuda()
        gen = torch.Generator().manual_seed(3)
        res = {}
        for mode in ('0', '1'):
            tr, _ = _trainer(fn, golden_dir, mode)
            losses, gnorm = [], []
            g2 = torch.Generator().manual_seed(9)
            for it in range(12):
                sel = torch.randint(0, ro.shape[0], (1024,), generator=g2).cuda()
                t_rand = torch.rand(1024, 64, generator=g2).cuda()
                u = torch.rand(1024, 128, generator=g2).cuda()
                if it == 0:
                    loss2, _ = tr.forward_backward(ro[sel], rd[sel], tgt[sel], t_rand=t_rand, u=u)
                    g_first = tr.grad.clone()
                    assert tr.last_step_live == (mode == '1')
                    if mode == '1':
                        c = tr.live_counts.cpu().tolist()
                        assert c[1] == 1024 * 192 and c[3] == 1024 * 64 and 0 < c[0] < c[1] and 0 < c[2] < c[3]
                        res['frac'] = (c[0] + c[2]) / (c[1] + c[3])
                loss2, _ = tr.step(ro[sel], rd[sel], tgt[sel], t_rand=t_rand, u=u)
                losses.append(loss2.cpu().tolist())
            res[mode] = (g_first, losses, tr.flat.clone())
        ga, gb = res['0'][0], res['1'][0]
        assert res['0'][1][0] == res['1'][1][0]                                   # same forward arithmetic: identical losses
        assert (ga - gb).abs().max().item() < 3e-6 * ga.abs().max().item()         # same gradient up to summation grouping
        la, lb = np.array(res['0'][1]), np.array(res['1'][1])
        assert np.abs(la - lb).max() < 2e-4 * la.max() and lb[-1, 0] < lb[0, 0]    # trajectories stay together, and train
        assert 0.2 < res['frac'] < 0.9
    finally:
        fn.render.set_compact(old)


def test_compaction_with_density_noise_and_ndc(fn, golden_dir, math_mode):
    """BASELINE configs[3] shape: NDC rays, raw_noise_std = 1 (render.py:170-180: alpha = 1 - exp(-relu(sigma + noise) * dist)):
    the dead set is where sigma + NOISE <= 0, about half of all samples at initialisation.  Same noise in both backward modes
    (seeded) -> identical losses, gradients to summation grouping."""
    old = fn.render.get_compact()
    try:
        H, W, focal = 756, 1008, 815.13
        K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
        gen = torch.Generator().manual_seed(4)
        pix = torch.stack([torch.zeros(512, dtype=torch.int64), torch.randint(0, H, (512,), generator=gen),
                           torch.randint(0, W, (512,), generator=gen)], 1).int().cuda()
        poses = torch.eye(4)[None, :3, :4].contiguous().cuda()
        ro, rd = fn.ops.gen_rays_pixels(pix, poses, K)
        tgt = torch.rand(512, 3, generator=gen).cuda()
        res = {}
        for mode in ('0', '1'):
            fn.render.set_compact(mode)
            torch.manual_seed(0)
            args = fn.run_nerf.make_args(N_importance=64, N_samples=64, perturb=1.0, raw_noise_std=1.0, no_reload=True, dataset_type='llff')
            ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
            assert 'ndc' not in ktr                                  # LLFF kwargs: render() / Trainer default to ndc=True
            tr = fn.run_nerf.Trainer(ktr, H, W, K, 0.0, 1.0)
            t_rand, u = torch.rand(512, 64, generator=torch.Generator().manual_seed(8)).cuda(), torch.rand(512, 64, generator=torch.Generator().manual_seed(9)).cuda()
            torch.manual_seed(77)                                    # the sigma noise of this step
            loss2, _ = tr.forward_backward(ro, rd, tgt, t_rand=t_rand, u=u)
            res[mode] = (loss2.cpu(), tr.grad.clone(), tr.live_counts.cpu().tolist() if mode == '1' else None)
        assert torch.equal(res['0'][0], res['1'][0])
        ga, gb = res['0'][1], res['1'][1]
        assert (ga - gb).abs().max().item() < 3e-6 * ga.abs().max().item()
        c = res['1'][2]
        assert 0.35 < c[0] / c[1] < 0.65 and 0.35 < c[2] / c[3] < 0.65         # sigma ~ 0.02 against unit noise: half are dead
    finally:
        fn.render.set_compact(old)


def test_autograd_route_and_policy(fn, golden_dir, math_mode):
    """render(...); loss.backward() picks the compacted backward too; `auto` follows the measured live fraction."""
    old = fn.render.get_compact()
    try:
        tr, K = _trainer(fn, golden_dir, '1')
        ktr = dict(network_fn=tr.net_c, network_fine=tr.net_f, N_samples=64, N_importance=128, perturb=1.0, white_bkgd=True,
                   raw_noise_std=0., use_viewdirs=True, ndc=False, lindisp=False, network_query_fn=None)
        gen = torch.Generator().manual_seed(2)
        c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
        ro, rd = fn.run_nerf_helpers.get_rays(800, 800, K, c2w)
        sel = torch.randint(0, 640000, (512,), generator=gen).cuda()
        rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
        tgt = torch.rand(512, 3, generator=gen).cuda()
        grads = {}
        for mode in ('0', '1'):
            fn.render.set_compact(mode)
            for p in list(tr.net_c.parameters()) + list(tr.net_f.parameters()):
                p.grad = None
            rgb, disp, acc, ex = fn.render.render(800, 800, K, chunk=32768, rays=rays, retraw=True, near=2.0, far=6.0,
                                                  pytest=True, **ktr)
            loss = fn.run_nerf_helpers.img2mse(rgb, tgt) + fn.run_nerf_helpers.img2mse(ex['rgb0'], tgt)
            loss.backward()
            grads[mode] = torch.cat([p.grad.reshape(-1) for p in list(tr.net_c.parameters()) + list(tr.net_f.parameters())])
        assert (grads['0'] - grads['1']).abs().max().item() < 3e-6 * grads['0'].abs().max().item()
        # the policy: a high measured fraction switches compaction off, a low one back on; probes keep measuring
        pol = fn.render.LivePolicy()
        fn.render.set_compact('auto')
        assert pol.use_live(tr.net_c, tr.net_f, 128)
        seq = ((0.9, False), (0.75, False), (0.5, True), (0.74, True)) if math_mode == 'bf16x3' else \
              ((0.9, False), (0.66, False), (0.5, True), (0.66, True))
        for frac, want in seq:
            pol._pending = (torch.tensor([int(frac * 1000), 1000, 0, 0], dtype=torch.int32), torch.cuda.Event(), 0)
            pol._pending[1].record(); torch.cuda.synchronize()
            pol.step = 1
            assert pol.use_live(tr.net_c, tr.net_f, 128) == want and abs(pol.frac - frac) < 1e-3
        pol.on, pol.step = False, pol.PROBE
        assert pol.use_live(tr.net_c, tr.net_f, 128)                       # probe step
        assert not pol.use_live(tr.net_c, tr.net_c, 128)                   # one shared net for both passes: plain backward
        assert pol.use_live(tr.net_c, None, 0)
    finally:
        fn.render.set_compact(old)


def test_skip_colour_of_dead_tiles(fn, math_mode):
    """FN_FWD_SKIP_DEAD_RGB: in the first pass of a compacted step a 64-point tile whose samples all have sigma <= 0 skips the
    feature / view / colour layers.  Those samples have weight exactly zero, so every output of the renderer and every
    gradient must be bit-identical with and without the option; only the colour logits of such tiles differ (reported as 0)."""
    from fastnerf import synthetic
    dev = torch.device('cuda')
    N = 1024
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
    focal = 0.5 * 800 / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 400.0], [0, focal, 400.0], [0, 0, 1]])
    poses = torch.stack([synthetic.pose_spherical(-180.0 + 36.0 * k, -30.0, 4.0)[:3, :4] for k in range(10)], 0).to(dev)
    gen = torch.Generator().manual_seed(4)
    batches = []
    for _ in range(16):
        pix = torch.stack([torch.randint(0, 10, (N,), generator=gen), torch.randint(0, 800, (N,), generator=gen),
                           torch.randint(0, 800, (N,), generator=gen)], 1).int().to(dev)
        ro, rd = fn.ops.gen_rays_pixels(pix, poses, K)
        batches.append((ro, rd, synthetic.render_rays(ro, rd, cutoff=1.5).contiguous()))
    torch.manual_seed(0)
    ktr = fn.run_nerf.create_nerf(args, device=dev)[0]
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    for i in range(250):                       # solid bodies in empty space: rays that miss them become all-dead tiles
        tr.step(*batches[i % 16])
    ro, rd, tgt = batches[3]
    t_rand, u = torch.rand(N, 64, generator=gen).to(dev), torch.rand(N, 128, generator=gen).to(dev)
    rays11 = fn.ops.pack_rays(ro, rd, 2.0, 6.0)
    outs = {}
    for skip in (False, True):
        outs[skip], _ = fn.render._forward_core(rays11, tr.net_c, tr.net_f, 64, 128, False, 1.0, True, t_rand, u, None, None, save=False,
                                                packed_c=tr.pc, packed_f=tr.pf, skip_dead_rgb=skip)
    a, b = outs[False], outs[True]
    for k in ('rgb_map', 'disp_map', 'acc_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'weights', 'z_vals', 'depth_map', 'weights0', 'z0'):
        assert torch.equal(a[k], b[k]) or (torch.isnan(a[k]) == torch.isnan(b[k])).all() and torch.equal(a[k].nan_to_num(), b[k].nan_to_num()), k
    ra, rb_ = a['raw'].reshape(-1, 4), b['raw'].reshape(-1, 4)          # fine pass, 1024 * 192 points = 3072 tiles
    assert torch.equal(ra[:, 3], rb_[:, 3])
    dead_tile = (ra[:, 3].reshape(-1, 64) <= 0).all(1)
    frac = float(dead_tile.float().mean())
    assert frac > 0.2, frac                                               # (measured 0.55-0.7 after 250 steps)
    live_rows = (~dead_tile)[:, None].expand(-1, 64).reshape(-1)
    assert torch.equal(ra[live_rows], rb_[live_rows])
    assert float(rb_[~live_rows][:, :3].abs().max()) == 0.0 and float(ra[~live_rows][:, :3].abs().max()) > 0.0
    # the whole compacted step: same loss, same gradient, bit for bit
    old = fn.render.get_compact()
    fn.render.set_compact('1')
    try:
        g = {}
        for skip in (False, True):
            tr.skip_dead_rgb = skip
            loss2, _ = tr.forward_backward(ro, rd, tgt, t_rand=t_rand, u=u)
            assert tr.last_step_live
            g[skip] = (loss2.clone(), tr.grad.clone(), tr.live_counts.clone())
        assert torch.equal(g[False][0], g[True][0]) and torch.equal(g[False][1], g[True][1]) and torch.equal(g[False][2], g[True][2])
    finally:
        fn.render.set_compact(old)
    # with sigma noise the option is not applied (a dead sigma can come alive): logits stay exact
    n1 = torch.randn(N, 192, generator=gen).to(dev)
    n0 = torch.randn(N, 64, generator=gen).to(dev)
    o_n, _ = fn.render._forward_core(rays11, tr.net_c, tr.net_f, 64, 128, False, 1.0, True, t_rand, u, n0, n1, save=False,
                                     packed_c=tr.pc, packed_f=tr.pf, skip_dead_rgb=True)
    o_r, _ = fn.render._forward_core(rays11, tr.net_c, tr.net_f, 64, 128, False, 1.0, True, t_rand, u, n0, n1, save=False,
                                     packed_c=tr.pc, packed_f=tr.pf, skip_dead_rgb=False)
    assert torch.equal(o_n['raw'], o_r['raw']) and torch.equal(o_n['rgb_map'], o_r['rgb_map'])


def test_render_without_retraw_uses_the_skip_and_matches(fn, golden_dir):
    """render() / render_path never hand out colour logits unless retraw=True: without it the inference launches skip the colour
    branch of tiles without a live sample, and every map is bit for bit what the retraw=True call returns."""
    from fastnerf import synthetic
    dev = torch.device('cuda')
    N = 1024
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
    focal = 0.5 * 800 / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 400.0], [0, focal, 400.0], [0, 0, 1]])
    poses = torch.stack([synthetic.pose_spherical(-180.0 + 36.0 * k, -30.0, 4.0)[:3, :4] for k in range(10)], 0).to(dev)
    gen = torch.Generator().manual_seed(4)
    torch.manual_seed(0)
    ktr, kte, _, _, _, _ = fn.run_nerf.create_nerf(args, device=dev)
    tr = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    for i in range(200):                       # a field with empty space around solid bodies
        pix = torch.stack([torch.randint(0, 10, (N,), generator=gen), torch.randint(0, 800, (N,), generator=gen),
                           torch.randint(0, 800, (N,), generator=gen)], 1).int().to(dev)
        ro, rd = fn.ops.gen_rays_pixels(pix, poses, K)
        tr.step(ro, rd, synthetic.render_rays(ro, rd, cutoff=1.5).contiguous())
    ro, rd = fn.run_nerf_helpers.get_rays(800, 800, K, poses[3])
    rays = (ro[200:328:2, 200:328:2].reshape(-1, 3).contiguous(), rd[200:328:2, 200:328:2].reshape(-1, 3).contiguous())
    with torch.no_grad():
        a = fn.render.render(800, 800, K, chunk=32768, rays=rays, near=2.0, far=6.0, retraw=True, **kte)
        b = fn.render.render(800, 800, K, chunk=32768, rays=rays, near=2.0, far=6.0, **kte)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x.nan_to_num(), y.nan_to_num()) and torch.equal(torch.isnan(x), torch.isnan(y))
    assert 'raw' in a[3] and 'raw' not in b[3] and torch.equal(a[3]['z_std'], b[3]['z_std'])
    dead = (a[3]['raw'][..., 3].reshape(-1, 64) <= 0).all(1).float().mean()
    assert 0.05 < float(dead) < 0.999, float(dead)       # the option had something to skip, and something to keep
