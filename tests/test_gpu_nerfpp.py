"""nerf++-ours path (config 5) on the HIP kernels vs the reference's
golden vectors (G10) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import nerfpp_oracle as PP

pytestmark = pytest.mark.gpu
TOL_RGB = 1e-4


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def T(a):
    return torch.from_numpy(np.asarray(a))


def load_levels(golden_dir):
    w = np.load(os.path.join(golden_dir, 'g10_pp_weights.npz'))
    return [({k[len(f'l{m}.fg_net.'):]: T(w[k]).clone() for k in w.files if k.startswith(f'l{m}.fg_net.')},
             {k[len(f'l{m}.bg_net.'):]: T(w[k]).clone() for k in w.files if k.startswith(f'l{m}.bg_net.')})
            for m in range(2)]


def make_nets(fn, golden_dir):
    nets = []
    for fg, bg in load_levels(golden_dir):
        net = fn.nerfpp.NerfNetWithAutoExpo(None)
        sd = {'fg_net.' + k: v for k, v in fg.items()}
        sd.update({'bg_net.' + k: v for k, v in bg.items()})
        assert list(net.nerf_net.state_dict().keys()) == list(sd.keys())
        net.nerf_net.load_state_dict(sd)
        nets.append(net)
    return nets


def test_pp_ops(fn, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_pp_ops.npz'))
    rays11 = fn.ops.pack_rays(G(g['ray_o']), G(g['ray_d']), 0.0, 0.0)
    far = fn.ops.pp_intersect_sphere(rays11)
    assert np.abs(far.cpu().numpy() - g['fg_far']).max() < 2e-6
    with pytest.raises(Exception, match='unit sphere'):
        fn.nerfpp.intersect_sphere(torch.tensor([[2.0, 0, 0]]).cuda(), torch.tensor([[0.0, 1.0, 0]]).cuda())
    gen = torch.Generator().manual_seed(2)
    t = torch.rand(40, 64, generator=gen)
    near = 1e-4 * torch.ones(40)
    ref = PP.perturb_samples(PP.fg_depths(T(g['fg_far']), near, 64), t)
    got = fn.ops.pp_fg_depths(G(g['fg_far']), 64, 1e-4, True, t.cuda())
    assert (got.cpu() - ref).abs().max() < 2e-6
    ref0 = PP.fg_depths(T(g['fg_far']), near, 64)
    got0 = fn.ops.pp_fg_depths(G(g['fg_far']), 64, 1e-4, False)
    assert (got0.cpu() - ref0).abs().max() < 2e-6
    # sampler variant: z such that mid(z) are sorted bins -> compare through the oracle directly
    z = torch.sort(torch.rand(40, 64, generator=gen), -1).values
    w = torch.rand(40, 64, generator=gen) ** 3
    u = torch.rand(40, 128, generator=gen)
    smp = PP.sample_pdf(0.5 * (z[:, 1:] + z[:, :-1]), w[:, 1:-1], 128, u)
    ref, _ = torch.sort(torch.cat((z, smp), -1), -1)
    zo, zs = fn.ops.pp_sample_pdf_merge(z.cuda(), w.cuda(), 128, u=u.cuda())
    width = float((z[:, 1:] - z[:, :-1]).max())
    for a, b in ((zs.cpu(), smp), (zo.cpu(), ref)):
        err = (a - b).abs()
        assert (err < 2e-5).float().mean() > 0.98 and err.max() <= width + 1e-5, float(err.max())


def test_bg_mlp_vs_oracle(fn, golden_dir, math_mode):
    """kind-2 net: inverted-sphere points + 84-channel encoding + flipped sample order."""
    fg, bg = load_levels(golden_dir)[0]
    gen = torch.Generator().manual_seed(5)
    n, S = 7, 40       # 280 points: 4 full 64-point tiles + a tail
    ro = (torch.rand(n, 3, generator=gen) - 0.5) * 0.8
    rd = torch.randn(n, 3, generator=gen)
    z = torch.sort(torch.rand(n, S, generator=gen) * 0.98 + 0.01, -1).values
    vd = rd / rd.norm(dim=-1, keepdim=True)
    pts, _ = PP.depth2pts_outside(ro[:, None].expand(n, S, 3), rd[:, None].expand(n, S, 3), z)
    inp = torch.flip(torch.cat((PP.embed(pts, 10), PP.embed(vd[:, None].expand(n, S, 3), 4)), -1), dims=[-2])
    sd = {k: v.clone().requires_grad_(True) for k, v in bg.items()}
    rgb, sigma = PP.mlpnet_forward(sd, inp, 84)
    flat = torch.cat([bg[name].reshape(-1) for name, _ in PP.mlpnet_param_shapes(84)]).cuda()
    pf, pb = fn.ops.mlp_pack(flat, kind=2)
    rays11 = fn.ops.pack_rays(ro.cuda(), rd.cuda(), 0.0, 0.0)
    act = torch.empty(fn.ops.act_floats(n * S, 2)).cuda()
    raw = fn.ops.mlp_fwd(rays11, z.cuda(), flat, pf, act=act, kind=2).cpu()
    assert (torch.sigmoid(raw[..., :3]) - rgb.detach()).abs().max() < 2e-5
    assert (raw[..., 3].abs() - sigma.detach()).abs().max() < 2e-5
    # backward w.r.t. all parameters through a random cotangent on the raw logits
    cot = torch.randn(n, S, 4, generator=gen)
    lin = torch.nn.functional.linear
    # oracle raw logits (pre sigmoid / abs): recompute with autograd on the same graph
    raw_o = torch.cat((torch.logit(rgb.clamp(1e-6, 1 - 1e-6)), torch.zeros(n, S, 1)), -1)
    loss = (torch.logit(rgb) * cot[..., :3]).sum() + (sigma * torch.sign(raw[..., 3]) * cot[..., 3]).sum()
    grads_ref = torch.autograd.grad(loss, list(sd.values()))
    dact = torch.empty(fn.ops.dact_floats(n * S, 2)).cuda()
    partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
    grads = torch.empty(fn.ops.net_floats(2, 0)).cuda()
    fn.ops.mlp_bwd(cot.cuda(), act, flat, pb, dact, partial, grads, kind=2)
    grads = grads.cpu()
    off = 0
    for (name, shape), gr in zip(PP.mlpnet_param_shapes(84), grads_ref):
        k = gr.numel()
        got = grads[off:off + k].view(shape)
        assert (got - gr).abs().max() < 1e-4 * max(1.0, gr.abs().max().item()), (name, (got - gr).abs().max().item())
        off += k


def test_nerfnet_forward_g10(fn, golden_dir, math_mode):
    g = np.load(os.path.join(golden_dir, 'g10_pp_step.npz'))
    nets = make_nets(fn, golden_dir)
    with torch.no_grad():
        ret = nets[0](G(g['ro']), G(g['rd']), G(g['fg_far']), G(g['l0.fg_z']), G(g['l0.bg_z']))
    assert list(ret.keys()) == ['rgb', 'fg_weights', 'bg_weights', 'fg_rgb', 'fg_depth', 'bg_rgb', 'bg_depth', 'bg_lambda']
    for k in ret:
        ref = g['l0.' + k]
        err = np.abs(ret[k].cpu().numpy() - ref).max()
        assert err < TOL_RGB * max(1.0, np.abs(ref).max()), (k, err)


def test_cascade_step_g10(fn, golden_dir, math_mode):
    g = np.load(os.path.join(golden_dir, 'g10_pp_step.npz'))
    nets = make_nets(fn, golden_dir)
    tr = fn.nerfpp.CascadeTrainer(nets, cascade_samples=(64, 128), lrate=5e-4)
    rand = [{'fg_t': G(g['fg_t']), 'bg_t': G(g['bg_t'])}, {'fg_u': G(g['fg_u']), 'bg_u': G(g['bg_u'])}]
    losses, rgb = tr.step(G(g['ro']), G(g['rd']), G(g['target']), rand=rand, update=False)
    assert np.abs(rgb.cpu().numpy() - g['rgb_pred']).max() < TOL_RGB
    for m in range(2):
        net = nets[m].nerf_net
        flat_g = net.flat_grad.cpu()
        off = 0
        for pre, kind in (('fg_net.', 1), ('bg_net.', 2)):
            for name, o, shape in fn.nerfpp.mlpnet_slices(kind):
                k = int(np.prod(shape))
                got = flat_g[off + o: off + o + k].view(shape).numpy()
                ref = g[f'grad.l{m}.{pre}{name}']
                if got.size > 40000:
                    got = got[:16]
                scale = max(np.abs(ref).max(), 1e-6)
                # same bound as the nerf-ours step test: input-sensitivity of the resampled positions
                assert np.abs(got - ref).max() < 3e-2 * scale, (m, pre + name, np.abs(got - ref).max(), scale)
            off += fn.ops.net_floats(kind, 0)
    # (2) kernel exactness: oracle autograd evaluated at the device's own depths of each level (level 1's come from the
    # inverse-CDF resampling, whose 1-ulp position differences dominate (1)).  This batch's gradient is badly conditioned
    # (random-init nets: d loss / d sigma is a difference of nearly equal colours): the fp32 oracle itself is 9e-5 (fg
    # trunk) .. 1.4e-3 (bg sigma head) away from its own fp64 evaluation in relative L2.  Measured here: fp32 mode
    # 4.5e-4, split-bf16 mode 3.9e-3 (= the same amplification on a 2^-17 instead of a 2^-24 unit roundoff)
    # -> bounds 2e-3 / 8e-3 in relative L2 and 1e-2 of each tensor's max.
    levels = load_levels(golden_dir)
    ro_c, rd_c, tgt_c = T(g['ro']), T(g['rd']), T(g['target'])
    fg_far_c = PP.intersect_sphere(ro_c, rd_c)
    rels = []
    for m in range(2):
        sd_fg, sd_bg = levels[m]
        params = list(sd_fg.values()) + list(sd_bg.values())
        for p in params:
            p.requires_grad_(True)
        fz, bz = (z.cpu() for z in tr.last_depths[m])
        ret = PP.nerfnet_forward(sd_fg, sd_bg, ro_c, rd_c, fg_far_c, fz, bz)
        gr = torch.autograd.grad(torch.mean((ret['rgb'] - tgt_c) ** 2), params)
        flat_g = nets[m].nerf_net.flat_grad.cpu()
        off, it = 0, iter(gr)
        for pre, kind in (('fg_net.', 1), ('bg_net.', 2)):
            for name, o, shape in fn.nerfpp.mlpnet_slices(kind):
                gg = next(it)
                got = flat_g[off + o: off + o + gg.numel()].view(gg.shape)
                rel = float((got - gg).norm() / (gg.norm() + 1e-12))
                rmax = float((got - gg).abs().max() / max(gg.abs().max().item(), 1e-7))
                rels.append((rel, rmax, m, pre + name))
            off += fn.ops.net_floats(kind, 0)
    rels.sort(reverse=True)
    print('worst relative L2 / max gradient differences vs the oracle at the device depths:', rels[:6])
    assert rels[0][0] < (2e-3 if math_mode == 'fp32' else 8e-3) and max(r[1] for r in rels) < 1e-2, rels[:6]
    # autograd route through the mirrored NerfNet API gives the same gradients as the fused trainer
    net = nets[0].nerf_net
    fused = net.flat_grad.clone()
    for p in net.parameters():
        p.grad = None
    ro, rd, tgt = G(g['ro']), G(g['rd']), G(g['target'])
    ret = nets[0](ro, rd, G(g['fg_far']), G(g['l0.fg_z']), G(g['l0.bg_z']))
    loss = torch.mean((ret['rgb'] - tgt) ** 2)
    loss.backward()
    ag = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert (ag - fused).abs().max() < 1e-4 * fused.abs().max()
    # optimiser update is bounded by lr on the first step
    before = net.flat.clone()
    tr.step(ro, rd, tgt, rand=rand, update=True)
    assert float((net.flat - before).abs().max()) <= 5e-4 * 1.001


def test_pp_gen_rays_and_sumcount(fn, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_pp_ops.npz'))
    ro, rd = fn.ops.pp_gen_rays(6, 8, g['intr'], g['c2w'])
    assert np.abs(ro.cpu().numpy() - g['ro_s']).max() < 1e-7
    assert np.abs(rd.cpu().numpy() - g['rd_s']).max() < 1e-6 * np.abs(g['rd_s']).max()
    # per-(image, leaf) fp64 sums / counts == host reduction; mean rule on the device tags
    gen = torch.Generator().manual_seed(9)
    n, ml = 5000, 13
    rgb, tgt = torch.rand(n, 3, generator=gen), torch.rand(n, 3, generator=gen)
    tag = torch.stack([torch.randint(0, 3, (n,), generator=gen), torch.randint(0, ml, (n,), generator=gen)], 1).int()
    sums = torch.zeros(3 * ml, dtype=torch.float64).cuda()
    counts = torch.zeros(3 * ml, dtype=torch.int32).cuda()
    fn.ops.leaf_sumcount(rgb.cuda(), tgt.cuda(), tag.cuda(), ml, sums, counts)
    slot = tag[:, 0].long() * ml + tag[:, 1].long()
    ref_s = torch.zeros(3 * ml, dtype=torch.float64).index_add_(0, slot, (tgt - rgb).abs().double().sum(-1))
    ref_c = torch.zeros(3 * ml, dtype=torch.int32).index_add_(0, slot, torch.ones(n, dtype=torch.int32))
    assert torch.equal(counts.cpu(), ref_c) and (sums.cpu() - ref_s).abs().max() < 1e-9


def test_pp_quadtree_on_device(fn, golden_dir):
    """nerf++ manager end to end on the GPU: picks identical to the reference's (G12), device gather,
    device sum/count reduction, MEAN split."""
    g = np.load(os.path.join(golden_dir, 'g12_pp_tree.npz'))
    H, W = int(g['H']), int(g['W'])

    class RS:
        pass
    samplers = []
    for i in range(g['images'].shape[0]):
        rs = RS(); rs.H, rs.W = H, W
        rs.img = g['images'][i].reshape(-1, 3); rs.rays_o = np.zeros((H * W, 3), dtype=np.float32)
        rs.rays_d = g['rays_d'][i].reshape(-1, 3)
        samplers.append(rs)
    mgr = fn.nerfpp.QuadTreeManager(samplers, mseThres=0.0, max_depth=2, sharp_imgs=list(g['sharp']))
    for rnd in range(4):
        torch.manual_seed(200 + rnd); np.random.seed(300 + rnd)
        o, d, rgb = mgr.gen_rays_v3_multiThread(down_scale=1, prob=True, rand=0.5)
        assert rgb.is_cuda and np.array_equal(rgb.cpu().numpy(), g[f'r{rnd}_rgb'])
        assert np.array_equal(d.cpu().numpy(), g[f'r{rnd}_d'])
        mgr.adjust_tree_multiThread(rgb, torch.from_numpy(g[f'r{rnd}_pred']).cuda(), thres=0.012)
        for ti in range(len(samplers)):
            assert np.array_equal(mgr.leaves(ti), g[f'r{rnd}_after_t{ti}']), (rnd, ti)


def test_pp_prob_picks_on_device(fn, golden_dir):
    """The vectorised prob=True sampler runs on the GPU and feeds the same gather / tag plumbing."""
    g = np.load(os.path.join(golden_dir, 'g12_pp_tree.npz'))
    H, W = int(g['H']), int(g['W'])
    imgs, rays_d = g['images'], g['rays_d']

    class RS:
        pass
    samplers = []
    for i in range(imgs.shape[0]):
        rs = RS()
        rs.H, rs.W = H, W
        rs.img = imgs[i].reshape(-1, 3)
        rs.rays_o = np.zeros((H * W, 3), dtype=np.float32)
        rs.rays_d = rays_d[i].reshape(-1, 3)
        samplers.append(rs)
    mgr = fn.nerfpp.QuadTreeManager(samplers, mseThres=0.0, max_depth=2, device='cuda', sharp_imgs=list(g['sharp']))
    torch.manual_seed(3)
    o, d, rgb = mgr.gen_rays_v3_multiThread(down_scale=1, prob=True, rand=0.5, compat_rng=False)
    n = imgs.shape[0] * H * W
    assert o.shape == (n, 3) and d.is_cuda and rgb.shape == (n, 3)
    tags = mgr.result_leaf_tag.cpu().numpy()
    pix = mgr.result_pix.cpu().numpy()
    assert np.allclose(rgb.cpu().numpy(), imgs[pix[:, 0], pix[:, 1], pix[:, 2]])
    for i in range(imgs.shape[0]):
        plan = mgr.leaf_plan(i, 1.0)
        assert np.array_equal(np.bincount(tags[tags[:, 0] == i, 1], minlength=plan.shape[0]), plan[:, 0])


def test_render_single_image_g14(fn, golden_dir, math_mode):
    """Whole-image evaluation of the cascade (ddp_test_nerf.py:126-227) vs the reference's own outputs, ragged chunks."""
    from collections import OrderedDict
    g = np.load(os.path.join(golden_dir, 'g14_pp_render.npz'))
    nets = make_nets(fn, golden_dir)

    class Sampler:
        H, W = 6, 8

        def get_all(self):
            return OrderedDict([('ray_o', torch.from_numpy(g['ray_o'])), ('ray_d', torch.from_numpy(g['ray_d'])), ('depth', None),
                                ('rgb', None), ('mask', None), ('min_depth', torch.full((48,), 1e-4))])
    models = {'cascade_level': 2, 'cascade_samples': [64, 128], 'net_0': nets[0], 'net_1': nets[1]}
    ret = fn.nerfpp.render_single_image(models, Sampler(), 20)
    assert len(ret) == 2
    for m in range(2):
        assert list(ret[m].keys()) == ['rgb', 'fg_rgb', 'fg_depth', 'bg_rgb', 'bg_depth', 'bg_lambda']
        for k, v in ret[m].items():
            ref = g['l%d.%s' % (m, k)]
            assert tuple(v.shape) == ref.shape and not v.is_cuda
            err = np.abs(v.numpy() - ref).max()
            # level 1 sits behind sample_pdf (ill-conditioned where the level-0 weights vanish): colours stay within the
            # parity bar, per-ray depths get the looser one
            tol = TOL_RGB if ('rgb' in k or k == 'bg_lambda') else 2e-3
            assert err < tol * max(1.0, np.abs(ref).max()), (m, k, err)


def test_pp_loader_rays(fn, golden_dir, tmp_path):
    """nerf++ scene-directory reader: rays / depth of every view and seeded random_sample batches vs the reference (G17);
    the samplers plug into render_single_image."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('pp_loader_golden', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'test_pp_loader_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    write_scene = mod.write_scene
    from fastnerf.data_loader_split import load_data_split
    g = np.load(os.path.join(golden_dir, 'g17_pp_loader.npz'))
    base = str(tmp_path)
    write_scene(g, base)
    for split, n in (('train', 3), ('test', 2)):
        samplers = load_data_split(base, 'scene', split)
        for i, s in enumerate(samplers):
            p = '%s.out%d.' % (split, i)
            al = s.get_all()
            assert al['ray_o'].shape == (24, 3) and al['rgb'].shape == (24, 3) and al['ray_d'].is_cuda
            assert np.abs(al['ray_o'].cpu().numpy() - g[p + 'rays_o']).max() < 1e-6
            assert np.abs(al['ray_d'].cpu().numpy() - g[p + 'rays_d']).max() < 2e-6
            assert np.abs(al['depth'].cpu().numpy() - g[p + 'depth']).max() < 1e-6
            assert (al['mask'] is None) == (split == 'test')
        np.random.seed(5)
        a = samplers[0].random_sample(7, center_crop=False)
        b = samplers[1].random_sample(4, center_crop=True)
        assert np.abs(a['ray_d'].cpu().numpy() - g['%s.rand_ray_d' % split]).max() < 2e-6
        assert np.abs(a['rgb'].numpy() - g['%s.rand_rgb' % split]).max() < 1e-7
        assert np.abs(b['ray_d'].cpu().numpy() - g['%s.crop_ray_d' % split]).max() < 2e-6
        assert a['img_name'].endswith('000.png') and list(a.keys()) == ['ray_o', 'ray_d', 'depth', 'rgb', 'mask', 'min_depth', 'img_name']
