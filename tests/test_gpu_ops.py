"""HIP kernels vs the CPU oracle and the golden vectors recorded from the reference.
Every call goes through the C ABI (ctypes).  Tolerances are stated per test; integer/index
results are bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return fastnerf


def L(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, atol, rtol=0.0):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else b
    err = np.abs(a - b)
    ok = np.all(err <= atol + rtol * np.abs(b))
    return ok, float(err.max())


def test_gen_rays_g1(fn, golden_dir):
    g = L(golden_dir, 'g1_get_rays.npz')
    Ks = np.array([[10.0, 0, 4.0], [0, 10.0, 3.0], [0, 0, 1]])
    o, d = fn.run_nerf_helpers.get_rays(6, 8, Ks, T(g['c2w']))
    assert np.array_equal(o.cpu().numpy(), g['small_o'])
    ok, e = close(d, g['small_d'], 1e-6); assert ok, e
    o, d = fn.run_nerf_helpers.get_rays(800, 800, g['K'], T(g['c2w']))
    idx = g['idx']
    ok, e = close(d.cpu().numpy()[idx[:, 0], idx[:, 1]], g['big_d'], 1e-6); assert ok, e
    assert np.array_equal(o.cpu().numpy()[idx[:, 0], idx[:, 1]], g['big_o'])
    # selected-pixel generation == full-image generation
    pix = torch.tensor([[0, int(r), int(c)] for r, c in idx], dtype=torch.int32).cuda()
    ro, rd = fn.ops.gen_rays_pixels(pix, G(g['c2w'])[None].contiguous(), g['K'])
    assert torch.equal(rd.cpu(), d.cpu()[idx[:, 0], idx[:, 1]])
    # the numpy twin (run_nerf_helpers.py:81-88; only the dead use_batching prelude calls it): numpy in, numpy out
    onp, dnp = fn.run_nerf_helpers.get_rays_np(6, 8, Ks, g['c2w'])
    assert isinstance(onp, np.ndarray) and onp.shape == (6, 8, 3) and onp.dtype == np.float32
    assert np.array_equal(onp, g['np_o'])
    ok, e = close(dnp, g['np_d'], 1e-6); assert ok, e


def test_ndc_g2(fn, golden_dir):
    g = L(golden_dir, 'g2_ndc.npz')
    no, nd = fn.run_nerf_helpers.ndc_rays(int(g['H']), int(g['W']), float(g['focal']), 1.0, G(g['ro']), G(g['rd']))
    ok, e = close(no, g['no'], 1e-6, 1e-6); assert ok, e
    ok, e = close(nd, g['nd'], 1e-6, 1e-6); assert ok, e


def test_posenc_g3(fn, golden_dir):
    g = L(golden_dir, 'g3_embed.npz')
    e10, d10 = fn.run_nerf_helpers.get_embedder(10, 0)
    e4, d4 = fn.run_nerf_helpers.get_embedder(4, 0)
    assert (d10, d4) == (63, 27)
    # |x| up to 6 and 2^9 => arguments ~3e3 rad: accurate range reduction required; 1 ulp of the
    # argument (2.4e-4) bounds the difference between two correct sin implementations
    ok, e = close(e10(G(g['x'])), g['e10'], 2e-6); assert ok, e
    ok, e = close(e4(G(g['x'])), g['e4'], 2e-6); assert ok, e


def test_pack_and_coarse(fn):
    gen = torch.Generator().manual_seed(5)
    ro = torch.randn(300, 3, generator=gen)
    rd = torch.randn(300, 3, generator=gen)
    ref = O.make_ray_batch(ro, rd, 2.0, 6.0)
    got = fn.ops.pack_rays(ro.cuda(), rd.cuda(), 2.0, 6.0)
    ok, e = close(got, ref, 1e-6); assert ok, e
    rdn = torch.cat([torch.rand(300, 2, generator=gen) - 0.5, -torch.ones(300, 1)], -1)
    ref = O.make_ray_batch(ro * 0.1, rdn, 0.0, 1.0, H=756, W=1008, focal=815.13, ndc=True)
    got = fn.ops.pack_rays((ro * 0.1).cuda(), rdn.cuda(), 0.0, 1.0, ndc=True, H=756, W=1008, focal=815.13)
    ok, e = close(got, ref, 2e-6, 2e-6); assert ok, e
    for S in (32, 64, 65):
        tr = torch.rand(300, S, generator=gen)
        for lindisp in (False, True):
            rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
            zr = O.coarse_z(rb[:, 6:7], rb[:, 7:8], S, lindisp, tr)
            zg = fn.ops.sample_coarse(rb.cuda(), S, lindisp=lindisp, t_rand=tr.cuda())
            ok, e = close(zg, zr, 1e-6); assert ok, (S, lindisp, e)
            z0 = O.coarse_z(rb[:, 6:7], rb[:, 7:8], S, lindisp, None)
            zg0 = fn.ops.sample_coarse(rb.cuda(), S, lindisp=lindisp)
            ok, e = close(zg0, z0, 1e-6); assert ok, e
    # Philox jitter: stratified, inside the bins, deterministic per seed
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0).cuda()
    za = fn.ops.sample_coarse(rb, 64, perturb=True, seed=7)
    zb = fn.ops.sample_coarse(rb, 64, perturb=True, seed=7)
    zc = fn.ops.sample_coarse(rb, 64, perturb=True, seed=8)
    assert torch.equal(za, zb) and not torch.equal(za, zc)
    assert (za[:, 1:] >= za[:, :-1]).all() and (za >= 2.0).all() and (za <= 6.0).all()


@pytest.mark.parametrize('S', [64, 192])
@pytest.mark.parametrize('wb', [0, 1])
def test_raw2outputs_g5(fn, golden_dir, S, wb):
    g = L(golden_dir, f'g5_raw2out_S{S}_wb{wb}.npz')
    rgb, disp, acc, w, depth = fn.render.raw2outputs(G(g['raw']), G(g['z']), G(g['rd']), 0, bool(wb))
    for a, k in ((rgb, 'rgb'), (acc, 'acc'), (w, 'weights'), (depth, 'depth')):
        ok, e = close(a, g[k], 2e-6, 2e-6); assert ok, (k, e)
    ok, e = close(disp, g['disp'], 1e-6, 1e-5); assert ok, e
    rays11 = torch.zeros(64, 11).cuda(); rays11[:, 3:6] = G(g['rd'])
    draw = fn.ops.raw2outputs_bwd(G(g['raw']), G(g['z']), rays11, G(g['cot']), None, bool(wb))
    ok, e = close(draw, g['graw'], 2e-6, 1e-4); assert ok, e


def test_raw2outputs_noise_and_small(fn, golden_dir):
    g = L(golden_dir, 'g5_raw2out_noise.npz')
    rays11 = torch.zeros(16, 11).cuda(); rays11[:, 3:6] = G(g['rd'])
    rgb, disp, acc, w, depth = fn.ops.raw2outputs_fwd(G(g['raw']), G(g['z']), rays11, G(g['noise']), False)
    for a, k in ((rgb, 'rgb'), (acc, 'acc'), (w, 'weights'), (depth, 'depth')):
        ok, e = close(a, g[k], 2e-6, 2e-6); assert ok, (k, e)
    # ragged sample counts (not a multiple of the wave) against the oracle, incl. backward
    gen = torch.Generator().manual_seed(11)
    for S in (2, 7, 33, 100, 129, 257):  # S=1 is degenerate in the reference (empty dists)
        raw = (torch.randn(37, S, 4, generator=gen) * 2).requires_grad_(True)
        z = torch.sort(torch.rand(37, S, generator=gen) * 4 + 2, -1).values
        rd = torch.randn(37, 3, generator=gen)
        cot = torch.randn(37, 3, generator=gen)
        r = O.raw2outputs(raw, z, rd, None, True)
        gr, = torch.autograd.grad((r[0] * cot).sum(), raw)
        rays11 = torch.zeros(37, 11).cuda(); rays11[:, 3:6] = rd.cuda()
        out = fn.ops.raw2outputs_fwd(raw.detach().cuda(), z.cuda(), rays11, None, True)
        ok, e = close(out[0], r[0], 2e-6, 2e-6); assert ok, (S, e)
        ok, e = close(out[3], r[3], 2e-6, 2e-6); assert ok, (S, e)
        dr = fn.ops.raw2outputs_bwd(raw.detach().cuda(), z.cuda(), rays11, cot.cuda(), None, True)
        ok, e = close(dr, gr, 2e-6, 1e-4); assert ok, (S, e)
    # rays that hit nothing (every sigma <= 0): acc = depth = 0 exactly, the white background shows, and the disparity is
    # 1 / max(1e-10, 0 / 0) = NaN in the reference because torch.max propagates NaN (render.py:186) -- not 1e10
    raw = torch.randn(5, 12, 4, generator=gen)
    raw[1, :, 3] = -raw[1, :, 3].abs() - 0.1
    raw[3, :, 3] = 0.0
    z = torch.sort(torch.rand(5, 12, generator=gen) * 4 + 2, -1).values
    rd = torch.randn(5, 3, generator=gen)
    r = O.raw2outputs(raw, z, rd, None, True)
    rays11 = torch.zeros(5, 11).cuda(); rays11[:, 3:6] = rd.cuda()
    out = fn.ops.raw2outputs_fwd(raw.cuda(), z.cuda(), rays11, None, True)
    want_nan = torch.isnan(r[1])
    assert want_nan.tolist() == [False, True, False, True, False]
    assert torch.equal(torch.isnan(out[1]).cpu(), want_nan)
    assert float((out[1].cpu()[~want_nan] - r[1][~want_nan]).abs().max()) < 1e-5 * float(r[1][~want_nan].abs().max())
    assert torch.equal(out[2].cpu()[want_nan], torch.zeros(2)) and torch.equal(out[0].cpu()[want_nan], torch.ones(2, 3))


def test_sample_pdf_g6(fn, golden_dir):
    """Tolerance: the sample is bins[i] + (u-cdf[i])/(cdf[i+1]-cdf[i]) * width; a 1-ulp difference
    in the cdf (summation order) is amplified by 1/denom, so 2e-5 (<< the 0.06 bin width) is the
    bound for well-conditioned pdfs.  The 'spike' case puts 61 bins exactly AT the reference's
    `denom < 1e-5` switch (pdf = 1e-5/1.00062), where the reference itself is discontinuous in the
    last bit of the cdf; there we require the bulk to agree and every sample to stay within the
    same or the adjacent bin."""
    g = L(golden_dir, 'g6_sample_pdf.npz')
    bins = G(g['bins'])
    for name in ('rand', 'flat'):
        w = G(g['w_' + name])
        det = fn.ops.sample_pdf(bins, w, 128, det=True)
        ok, e = close(det, g[name + '_det'], 2e-5); assert ok, (name, e)
        us = fn.ops.sample_pdf(bins, w, 128, u=G(g['u']))
        ok, e = close(us, g[name + '_u'], 2e-5); assert ok, (name, e)
    w = G(g['w_spike'])
    width = float((g['bins'][:, 1:] - g['bins'][:, :-1]).max())
    for key, kw in (('spike_det', dict(det=True)), ('spike_u', dict(u=G(g['u'])))):
        got = fn.ops.sample_pdf(bins, w, 128, **kw).cpu().numpy()
        err = np.abs(got - g[key])
        out_frac = float((err >= 2e-5).mean())
        assert out_frac < 0.1 and err.max() <= width, \
            '%s: %.3f %% of the samples off by more than 2e-5 (bound 10 %% at the reference\'s own discontinuity), worst %.3e' % (
                key, 100 * out_frac, err.max())
        print('SAMPLEPDF %s outlier fraction %.4f %% max %.2e' % (key, 100 * out_frac, err.max()))
    # reference's pytest hook through the mirrored signature
    got = fn.run_nerf_helpers.sample_pdf(bins, G(g['w_rand']), 128, det=False, pytest=True)
    ok, e = close(got, g['rand_u'], 2e-5); assert ok, e


def test_sample_pdf_merge_vs_oracle(fn):
    gen = torch.Generator().manual_seed(21)
    for (S, Ni) in ((64, 128), (64, 64), (32, 17), (100, 130)):
        z = torch.sort(torch.rand(50, S, generator=gen) * 4 + 2, -1).values
        w = torch.rand(50, S, generator=gen) ** 4
        u = torch.rand(50, Ni, generator=gen)
        zm = 0.5 * (z[:, 1:] + z[:, :-1])
        smp = O.sample_pdf(zm, w[:, 1:-1], Ni, u)
        ref, _ = torch.sort(torch.cat([z, smp], -1), -1)
        zo, zs, zstd = fn.ops.sample_pdf_merge(z.cuda(), w.cuda(), Ni, u=u.cuda())
        # a 1-ulp cdf difference moves a sample by ulp/denom of a bin (denom >= 1e-5 => up to ~1e-2 bin);
        # require the bulk within 2e-5 and everything within one bin (see the det case below)
        width = float((z[:, 1:] - z[:, :-1]).max())
        for a, b in ((zs, smp), (zo, ref)):
            err = (a.cpu() - b).abs()
            out_frac = float((err >= 2e-5).float().mean())
            assert out_frac < 0.02 and err.max() <= width, \
                'S=%d Ni=%d: %.4f %% of the samples off by more than 2e-5 (bound 2 %%), worst %.3e of a %.3e bin' % (
                    S, Ni, 100 * out_frac, float(err.max()), width)
            print('SAMPLEPDF injected-u S=%d Ni=%d outlier fraction %.5f %% max %.2e' % (S, Ni, 100 * out_frac, float(err.max())))
        assert (zo[:, 1:] >= zo[:, :-1]).all()
        ok, e = close(zstd, torch.std(smp, -1, unbiased=False), 2e-5, 1e-5); assert ok, e
        # merged output is exactly the multiset union of its own inputs (bit-exact index work)
        both, _ = torch.sort(torch.cat([z.cuda(), zs], -1), -1)
        assert torch.equal(both, zo)
        zo2, _, _ = fn.ops.sample_pdf_merge(z.cuda(), w.cuda(), Ni, det=True)
        smp_d = O.sample_pdf(zm, w[:, 1:-1], Ni, None)
        ref_d, _ = torch.sort(torch.cat([z, smp_d], -1), -1)
        # deterministic u: low-probability bins (pdf < 1e-5 -> the reference's `denom -> 1` branch pins
        # the sample to the bin's left edge) make a 1-ulp cdf difference move a sample by one whole
        # bin, in the reference itself; require the bulk to agree and outliers to stay within a bin
        err = (zo2.cpu() - ref_d).abs()
        out_frac = float((err >= 2e-5).float().mean())
        assert out_frac < 0.02 and err.max() <= width, \
            'det S=%d Ni=%d: %.4f %% of the samples off by more than 2e-5 (bound 2 %%), worst %.3e of a %.3e bin' % (
                S, Ni, 100 * out_frac, float(err.max()), width)
        print('SAMPLEPDF det S=%d Ni=%d outlier fraction %.5f %% max %.2e' % (S, Ni, 100 * out_frac, float(err.max())))


def test_mse_leafmax_and_adam(fn):
    gen = torch.Generator().manual_seed(31)
    n = 1000
    rgb, rgb0, tgt = (torch.rand(n, 3, generator=gen) for _ in range(3))
    tag = torch.stack([torch.randint(0, 3, (n,), generator=gen), torch.randint(0, 17, (n,), generator=gen)], 1)
    table = torch.zeros(3 * 20, dtype=torch.int32).cuda()
    loss2, g, g0 = fn.ops.mse_leafmax(rgb.cuda(), rgb0.cuda(), tgt.cuda(), leaf_tag=tag.int().cuda(), max_leaves=20,
                                      table=table)
    assert abs(float(loss2[0]) - float(O.img2mse(rgb, tgt))) < 1e-6
    assert abs(float(loss2[1]) - float(O.img2mse(rgb0, tgt))) < 1e-6
    ok, e = close(g, 2 * (rgb - tgt) / (3 * n), 1e-9, 1e-6); assert ok, e
    ok, e = close(g0, 2 * (rgb0 - tgt) / (3 * n), 1e-9, 1e-6); assert ok, e
    ref = O.leaf_loss_max(tgt, rgb, tag, 3, 20)
    assert torch.equal(table.view(torch.float32).view(3, 20).cpu(), ref)  # max is exact
    # Adam vs the oracle's restatement of torch.optim.Adam, 3 steps, odd length (tail path)
    p = torch.randn(1003, generator=gen)
    opt = O.Adam([p.clone()], lr=5e-4)
    pg = p.clone().cuda(); m = torch.zeros_like(pg); v = torch.zeros_like(pg)
    for step in range(1, 4):
        gr = torch.randn(1003, generator=gen) * 1e-3
        opt.step([gr])
        fn.ops.adam_step(pg, gr.cuda(), m, v, 5e-4, step)
    ok, e = close(pg, opt.params[0], 1e-7, 1e-6); assert ok, e


def test_sigma_noise_is_one_philox_launch(fn):
    """render.py:162 `noise = torch.randn(raw[..., 3].shape) * raw_noise_std`: ops.sigma_noise fills the noise of both passes in one launch.
    N(0, std^2) to sampling accuracy (mean, variance, fourth moment, tails), a function of the seed alone, no repeated values between the
    passes, odd sizes complete."""
    n, S0, S1 = 4096, 64, 192
    a0, a1 = fn.ops.sigma_noise(n, S0, S1, 2.0, 77, torch.device('cuda'))
    b0, b1 = fn.ops.sigma_noise(n, S0, S1, 2.0, 77, torch.device('cuda'))
    c0, _ = fn.ops.sigma_noise(n, S0, S1, 2.0, 78, torch.device('cuda'))
    assert a0.shape == (n, S0) and a1.shape == (n, S1) and a0.is_contiguous() and a1.is_contiguous()
    assert torch.equal(a0, b0) and torch.equal(a1, b1) and not torch.equal(a0, c0)
    x = torch.cat([a0.reshape(-1), a1.reshape(-1)]).double() / 2.0
    N = x.numel()
    assert abs(float(x.mean())) < 5.0 / N ** 0.5 and abs(float(x.var()) - 1.0) < 5.0 * (2.0 / N) ** 0.5
    assert abs(float((x ** 4).mean()) - 3.0) < 5.0 * (96.0 / N) ** 0.5 and torch.isfinite(x).all()
    assert 3.5 < float(x.abs().max()) < 6.5 and abs(float((x.abs() > 1.96).double().mean()) - 0.05) < 2e-3
    assert len(torch.unique(x[:100000])) > 99000
    o0, o1 = fn.ops.sigma_noise(3, 5, 7, 1.0, 5, torch.device('cuda'))      # 15 + 21 values, the second view starts at a multiple of 4
    assert o0.shape == (3, 5) and o1.shape == (3, 7) and torch.isfinite(o0).all() and torch.isfinite(o1).all() and float(o1.abs().min()) > 0
    z0, z1 = fn.ops.sigma_noise(4, 8, 0, 1.0, 5, torch.device('cuda'))
    assert z1 is None and z0.shape == (4, 8)
