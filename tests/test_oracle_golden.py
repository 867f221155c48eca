"""Pins the CPU oracle (oracle/nerf_oracle.py) against golden vectors recorded
from the reference itself by oracle/make_golden.py.

Tolerances: the oracle restates the same fp32 op order on the same PyTorch CPU
kernels, so most outputs are expected bit-identical; where the restatement
legitimately re-associates (e.g. gather vs expand+gather) the bound is 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

torch.set_num_threads(4)


def L(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def T(a):
    return torch.from_numpy(np.asarray(a))


def load_sd(g, prefix):
    return {k[len(prefix):]: T(g[k]).clone() for k in g.files if k.startswith(prefix)}


def test_g1_get_rays(golden_dir):
    g = L(golden_dir, 'g1_get_rays.npz')
    c2w = T(g['c2w'])
    Ks = np.array([[10.0, 0, 4.0], [0, 10.0, 3.0], [0, 0, 1]])
    o, d = O.get_rays(6, 8, Ks, c2w)
    assert np.array_equal(o.numpy(), g['small_o']) and np.array_equal(d.numpy(), g['small_d'])
    o, d = O.get_rays(800, 800, g['K'], c2w)
    idx = g['idx']
    assert np.array_equal(o.numpy()[idx[:, 0], idx[:, 1]], g['big_o'])
    assert np.array_equal(d.numpy()[idx[:, 0], idx[:, 1]], g['big_d'])
    on, dn = O.get_rays_np(6, 8, Ks, g['c2w'])
    assert np.array_equal(dn, g['np_d']) and np.array_equal(on, g['np_o'])


def test_g2_ndc(golden_dir):
    g = L(golden_dir, 'g2_ndc.npz')
    no, nd = O.ndc_rays(int(g['H']), int(g['W']), float(g['focal']), 1.0, T(g['ro']), T(g['rd']))
    assert np.array_equal(no.numpy(), g['no']) and np.array_equal(nd.numpy(), g['nd'])


def test_g3_embed(golden_dir):
    g = L(golden_dir, 'g3_embed.npz')
    x = T(g['x'])
    assert np.array_equal(O.posenc(x, 10).numpy(), g['e10'])
    assert np.array_equal(O.posenc(x, 4).numpy(), g['e4'])
    assert O.posenc_dim(10) == 63 and O.posenc_dim(4) == 27


def test_g4_mlp(golden_dir):
    g = L(golden_dir, 'g4_mlp.npz')
    sd = load_sd(L(golden_dir, 'g7_weights.npz'), 'c.')
    names = [n for n, _ in O.nerf_param_shapes()]
    assert list(sd.keys()) == names
    for n, shp in O.nerf_param_shapes():
        assert tuple(sd[n].shape) == shp
    assert sum(v.numel() for v in sd.values()) == 595844
    for v in sd.values():
        v.requires_grad_(True)
    out = O.nerf_forward(sd, T(g['x']))
    assert np.allclose(out.detach().numpy(), g['out'], rtol=0, atol=1e-6)
    grads = torch.autograd.grad((out * T(g['cot'])).sum(), list(sd.values()))
    for n, gr in zip(names, grads):
        ref = g['grad.' + n]
        assert np.allclose(gr.numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max())), n


@pytest.mark.parametrize('S', [64, 192])
@pytest.mark.parametrize('wb', [0, 1])
def test_g5_raw2outputs(golden_dir, S, wb):
    g = L(golden_dir, f'g5_raw2out_S{S}_wb{wb}.npz')
    raw = T(g['raw']).clone().requires_grad_(True)
    rgb, disp, acc, w, depth = O.raw2outputs(raw, T(g['z']), T(g['rd']), None, bool(wb))
    for a, k in ((rgb, 'rgb'), (disp, 'disp'), (acc, 'acc'), (w, 'weights'), (depth, 'depth')):
        assert np.array_equal(a.detach().numpy(), g[k]), k
    graw, = torch.autograd.grad((rgb * T(g['cot'])).sum(), raw)
    assert np.allclose(graw.numpy(), g['graw'], rtol=0, atol=1e-7)


def test_g5_noise(golden_dir):
    g = L(golden_dir, 'g5_raw2out_noise.npz')
    rgb, disp, acc, w, depth = O.raw2outputs(T(g['raw']), T(g['z']), T(g['rd']), T(g['noise']), False)
    for a, k in ((rgb, 'rgb'), (disp, 'disp'), (acc, 'acc'), (w, 'weights'), (depth, 'depth')):
        assert np.array_equal(a.numpy(), g[k]), k


def test_g6_sample_pdf(golden_dir):
    g = L(golden_dir, 'g6_sample_pdf.npz')
    bins = T(g['bins'])
    for name in ('rand', 'flat', 'spike'):
        w = T(g['w_' + name])
        assert np.array_equal(O.sample_pdf(bins, w, 128, None).numpy(), g[name + '_det']), name
        assert np.array_equal(O.sample_pdf(bins, w, 128, T(g['u'])).numpy(), g[name + '_u']), name
    # ties go right (Appendix A): cdf=[0,.25,.25,1], u=.25 -> index 3
    assert int(torch.searchsorted(torch.tensor([0., .25, .25, 1.]), torch.tensor([.25]), right=True)) == 3


def _rays(g, near=2.0, far=6.0):
    return O.make_ray_batch(T(g['ro']), T(g['rd']), near, far)


def test_g7_render(golden_dir):
    g = L(golden_dir, 'g7_render.npz')
    wts = L(golden_dir, 'g7_weights.npz')
    sdc, sdf = load_sd(wts, 'c.'), load_sd(wts, 'f.')
    rb = _rays(g)
    with torch.no_grad():
        a = O.render_rays(rb, sdc, sdf, 64, 128, white_bkgd=True, retraw=True)
        b = O.render_rays(rb, sdc, sdf, 64, 128, white_bkgd=True, t_rand=T(g['t_rand']), u=T(g['u']), retraw=True)
        c = O.render_rays(rb, sdc, None, 32, 0, white_bkgd=True, retraw=True)
    pairs = (('rgb', 'rgb_map'), ('disp', 'disp_map'), ('acc', 'acc_map'), ('raw', 'raw'), ('rgb0', 'rgb0'),
             ('disp0', 'disp0'), ('acc0', 'acc0'), ('z_std', 'z_std'))
    for tag, res in (('a', a), ('b', b), ('c', c)):
        for gk, rk in pairs:
            if f'{tag}.{gk}' not in g.files:
                continue
            ref = g[f'{tag}.{gk}']
            got = res[rk].numpy()
            tol = 2e-6 * max(1.0, np.abs(ref).max())
            assert np.allclose(got, ref, rtol=1e-5, atol=tol), (tag, gk, np.abs(got - ref).max())


def test_g8_train_step(golden_dir):
    g = L(golden_dir, 'g8_train_step.npz')
    wts = L(golden_dir, 'g7_weights.npz')
    sdc, sdf = load_sd(wts, 'c.'), load_sd(wts, 'f.')
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    rb = _rays(g)
    loss, loss0, rgb, grads = O.train_step(sdc, sdf, opt, rb, T(g['target']), 64, 128, True,
                                           t_rand=T(g['t_rand']), u=T(g['u']))
    assert abs(float(loss) - float(g['loss'])) < 1e-6 and abs(float(loss0) - float(g['loss0'])) < 1e-6
    assert abs(float(O.mse2psnr(loss)[0]) - float(g['psnr'])) < 1e-4
    assert np.allclose(rgb.numpy(), g['rgb'], atol=2e-6)
    names = ['c.' + n for n in sdc] + ['f.' + n for n in sdf]
    for n, gr in zip(names, grads):
        ref = g['grad.' + n]
        assert np.allclose(gr.numpy(), ref, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(ref).max()) + 1e-9), n
    params = dict(zip(names, list(sdc.values()) + list(sdf.values())))
    n_checked = 0
    for k in g.files:
        if k.startswith('post.'):
            # first Adam step moves every weight by ~lr*sign(g); compare the update
            assert np.allclose(params[k[5:]].numpy(), g[k], rtol=0, atol=2e-6), k
            n_checked += 1
    assert n_checked > 20
    assert abs(O.lr_schedule(5e-4, 500, 0) - float(g['new_lr'])) < 1e-12
    assert abs(O.lr_schedule(5e-4, 500, 250000) - 5e-4 * 0.1 ** 0.5) < 1e-12


def test_g11_llff_ndc_noise(golden_dir):
    """Forward-facing config-4 shape: ndc warp (render.py:69-71), near 0 / far 1, sigma noise."""
    g = L(golden_dir, 'g11_llff_render.npz')
    wts = L(golden_dir, 'g7_weights.npz')
    sdc, sdf = load_sd(wts, 'c.'), load_sd(wts, 'f.')
    rb = O.make_ray_batch(T(g['ro']), T(g['rd']), 0.0, 1.0, H=int(g['H']), W=int(g['W']), focal=float(g['K'][0][0]),
                          ndc=True)
    np.random.seed(0); t_rand = torch.Tensor(np.random.rand(48, 64))
    np.random.seed(0); n0 = torch.Tensor(np.random.rand(48, 64)) * 1.0
    np.random.seed(0); u = torch.Tensor(np.random.rand(48, 64))
    np.random.seed(0); n1 = torch.Tensor(np.random.rand(48, 128)) * 1.0
    with torch.no_grad():
        r = O.render_rays(rb, sdc, sdf, 64, 64, white_bkgd=True, t_rand=t_rand, u=u, noise0=n0, noise1=n1)
    for gk, rk in (('rgb', 'rgb_map'), ('acc', 'acc_map'), ('rgb0', 'rgb0'), ('acc0', 'acc0'), ('z_std', 'z_std')):
        assert np.allclose(r[rk].numpy(), g[gk], rtol=1e-5, atol=2e-6), gk


def test_compute_ssim_g13(golden_dir):
    """Evaluation metric of render_path vs values recorded from the reference."""
    import fastnerf
    g = np.load(os.path.join(golden_dir, 'g13_ssim.npz'))
    H = fastnerf.run_nerf_helpers
    a, b, c, d = (torch.from_numpy(g[k]) for k in 'abcd')
    assert np.allclose(H.compute_ssim(a, b).numpy(), g['ssim_ab'], rtol=0, atol=2e-6)
    assert np.allclose(H.compute_ssim(a, b, return_map=True).numpy(), g['map_ab'], rtol=0, atol=1e-5)
    assert np.allclose(H.compute_ssim(a, a).numpy(), g['ssim_aa'], rtol=0, atol=2e-6)
    assert np.allclose(H.compute_ssim(c, d).numpy(), g['ssim_cd'], rtol=0, atol=2e-6)
    assert np.allclose(H.compute_ssim(c, d, max_val=2.0, filter_size=7, filter_sigma=1.0, k1=0.02, k2=0.05).numpy(),
                       g['ssim_cd_k'], rtol=0, atol=2e-6)


def test_g19_no_view_directions(golden_dir):
    """use_viewdirs=False (model.py:35-36,60-61; render.py:218): [N,8] ray batches, `output_linear` 256 -> 5, raw with five
    channels of which compositing reads four."""
    from conftest import noview_state_dicts
    g = L(golden_dir, 'g19_noview.npz')
    sdc, sdf = [{k: torch.from_numpy(v).clone() for k, v in sd.items()} for sd in noview_state_dicts(golden_dir)]
    assert tuple(sdc['views_linears.0.weight'].shape) == (128, 256) and tuple(sdc['output_linear.weight'].shape) == (5, 256)
    rb = O.make_ray_batch(torch.from_numpy(g['ro']), torch.from_numpy(g['rd']), 2.0, 6.0, use_viewdirs=False)
    assert rb.shape[1] == 8
    used = [k for k in sdc if not k.startswith('views_linears.')]
    ps = [sdc[k] for k in used] + [sdf[k] for k in used]
    for q in ps:
        q.requires_grad_(True)
    ret = O.render_rays(rb, sdc, sdf, 32, 32, False, True, torch.from_numpy(g['t_rand']), torch.from_numpy(g['u']), retraw=True)
    tgt = torch.from_numpy(g['target'])
    l1, l0 = O.img2mse(ret['rgb_map'], tgt), O.img2mse(ret['rgb0'], tgt)
    gr = torch.autograd.grad(l1 + l0, ps)
    assert ret['raw'].shape == (64, 64, 5)
    for k, ref in (('rgb_map', 'rgb'), ('disp_map', 'disp'), ('acc_map', 'acc'), ('raw', 'raw'), ('rgb0', 'rgb0'),
                   ('disp0', 'disp0'), ('acc0', 'acc0'), ('z_std', 'z_std')):
        got, want = ret[k].detach().numpy(), g[ref]
        nan = np.isnan(want)               # disp of a ray that hits nothing is 0 / 0 in the reference too (render.py:186)
        assert np.array_equal(np.isnan(got), nan), k
        assert np.abs(got[~nan] - want[~nan]).max() < 2e-5 * max(1.0, np.abs(want[~nan]).max()), k
    assert abs(float(l1) - float(g['loss'])) < 1e-6 and abs(float(l0) - float(g['loss0'])) < 1e-6
    n_checked = 0
    for (pre, k), gg in zip([('c.', k) for k in used] + [('f.', k) for k in used], gr):
        key = 'grad.' + pre + k
        if key in g.files:
            scale = max(np.abs(g[key]).max(), 1e-8)
            assert np.abs(gg.numpy() - g[key]).max() < 2e-4 * scale, (key, np.abs(gg.numpy() - g[key]).max() / scale)
            n_checked += 1
    assert n_checked == 2 * (8 + 1 + 1 + 2)    # biases of 8 trunk layers + output bias, output weight, trunk 0 / 7 weights
    for q in ps:
        q.requires_grad_(False)
    rt = O.render_rays(rb, sdc, sdf, 32, 32, False, True, None, None)
    assert np.abs(rt['rgb_map'].numpy() - g['test_rgb']).max() < 2e-5
    assert np.abs(rt['acc_map'].numpy() - g['test_acc']).max() < 2e-5
