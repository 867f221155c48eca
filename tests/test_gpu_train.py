"""Training-level parity: PSNR at equal iteration count vs the CPU oracle (north_star: within
0.1 dB), and the epoch driver with the quadtree in the loop."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fn():
    import fastnerf
    return fastnerf


def test_psnr_at_equal_iterations(fn, math_mode):
    imgs, poses, focal = fn.synthetic.make_dataset(n_images=6, H=24, W=24)
    H = W = 24
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    torch.manual_seed(0)
    args = fn.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, no_reload=True,
                                 lrate=5e-4, lrate_decay=500)
    ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
    sdc = {k: v.detach().cpu().clone() for k, v in ktr['network_fn'].state_dict().items()}
    sdf = {k: v.detach().cpu().clone() for k, v in ktr['network_fine'].state_dict().items()}
    tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    rays = [O.get_rays(H, W, K, poses[i]) for i in range(6)]
    ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3)
    rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3)
    tgt_all = imgs.reshape(-1, 3)
    gen = torch.Generator().manual_seed(1)
    n_iters, N = 40, 192
    lg, lc = [], []
    for it in range(n_iters):
        sel = torch.randint(0, ro_all.shape[0], (N,), generator=gen)
        t_rand, u = torch.rand(N, 16, generator=gen), torch.rand(N, 16, generator=gen)
        ro, rd, tgt = ro_all[sel], rd_all[sel], tgt_all[sel]
        loss2, _ = tr.step(ro.cuda(), rd.cuda(), tgt.cuda(), t_rand=t_rand.cuda(), u=u.cuda())
        lg.append(float(loss2[0]))
        opt.lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4   # pre-increment rule (run_nerf.py:498-508)
        l1, l0, _, _ = O.train_step(sdc, sdf, opt, O.make_ray_batch(ro, rd, 2.0, 6.0), tgt, 16, 16, True,
                                    t_rand=t_rand, u=u)
        lc.append(float(l1))
    lg, lc = np.array(lg), np.array(lc)
    assert lg[-1] < lg[0] * 0.8                         # it trains
    assert abs(lg[0] - lc[0]) < 1e-5
    # Free-running trajectories of the same batches are chaotic (sample_pdf is ill-conditioned): on this problem every math
    # mode follows the CPU oracle to 1e-4 for ~25 iterations and is 5 - 15 % away per iteration by iteration 30 - 39
    # (tools/psnr_prefix_divergence.py; fp32 6e-2, bf16x6 1.4e-1, bf16x3 2.6e-2 at iteration 39).  What a single trajectory can assert:
    # (a) BEFORE the decorrelation GPU and CPU are the same run -- per-iteration losses within 2e-3 over the first 20 iterations, the PSNR
    #     of iterations 10 .. 19 within 0.01 dB (north_star's 0.1 dB with an order of magnitude to spare);
    # (b) AFTER it they are two draws of one distribution: within 0.5 dB here.  The 0.1 dB statement about that distribution is the paired
    #     test over 60+ initialisation seeds (test_psnr_paired_with_the_cpu_ensemble_g22) and the per-step lockstep replay.
    assert np.max(np.abs(lg[:20] - lc[:20]) / lc[:20]) < 2e-3, np.abs(lg[:20] - lc[:20]) / lc[:20]
    assert abs(-10 * np.log10(np.mean(lg[10:20])) + 10 * np.log10(np.mean(lc[10:20]))) < 0.01
    psnr_g = -10 * np.log10(np.mean(lg[-5:]))
    psnr_c = -10 * np.log10(np.mean(lc[-5:]))
    assert abs(psnr_g - psnr_c) < 0.5, (psnr_g, psnr_c)
    assert abs(tr.lr - opt.lr * 0 - O.lr_schedule(5e-4, 500, n_iters - 1)) < 1e-12


def _psnr_300_iterations(fn, n_seeds, study):
    """Long-horizon equivalence of the math modes: 300 optimisation steps on the synthetic scene, identical batches and
    injected randoms.  north_star: PSNR within 0.1 dB at equal iteration count.

    Single trajectories are chaotic at this horizon (sample_pdf is ill-conditioned; at the BASELINE shape
    two runs decorrelate within ~30 iterations, tools/psnr_lockstep.py): the SAME arithmetic with another grouping of the dW
    partial sums -- the compacted instead of the plain backward -- already moves one seed's PSNR by up to 0.4 dB.  A mode is
    therefore a DISTRIBUTION of trajectories, and the statement tested is about its mean: enough seeds that the standard error
    of the mean per-seed difference is below 0.04 dB, then a hard 0.1 dB bound on that mean (2.5 standard errors)."""
    imgs, poses, focal = fn.synthetic.make_dataset(n_images=6, H=24, W=24)
    H = W = 24
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    rays = [O.get_rays(H, W, K, poses[i]) for i in range(6)]
    ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3).cuda()
    rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3).cuda()
    tgt_all = imgs.reshape(-1, 3).cuda()
    n_iters, n_prefix, N = 300, 40, 192       # (at 60 iterations single seeds are already 0.3 dB apart)
    n_full = 4                                # seeds 0..3 additionally run the compacted backward and the CPU oracle's prefix
    old, old_c = fn.ops.get_math(), fn.render.get_compact()
    keys = ('fp32', 'bf16x3', 'bf16x6', 'fp32_compacted', 'bf16x3_compacted')
    psnr = {k: [] for k in keys + ('oracle_prefix',) + tuple(k + '_prefix' for k in keys)}
    try:
        for seed in range(n_seeds):
            gen = torch.Generator().manual_seed(100 + seed)
            sched = []
            for it in range(n_iters):
                sched.append((torch.randint(0, ro_all.shape[0], (N,), generator=gen), torch.rand(N, 16, generator=gen),
                              torch.rand(N, 16, generator=gen)))
            dsched = [(s.cuda(), t.cuda(), u.cuda()) for s, t, u in sched]
            init = None
            for key in (keys if seed < n_full else keys[:3]):
                fn.ops.set_math(key.split('_')[0])
                fn.render.set_compact('1' if key.endswith('compacted') else '0')
                torch.manual_seed(seed)
                args = fn.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, no_reload=True,
                                             lrate=5e-4, lrate_decay=500)
                ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
                if init is None:
                    init = ({k: v.detach().cpu().clone() for k, v in ktr['network_fn'].state_dict().items()},
                            {k: v.detach().cpu().clone() for k, v in ktr['network_fine'].state_dict().items()})
                tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
                losses = []
                for sel, t_rand, u in dsched:
                    loss2, _ = tr.step(ro_all[sel], rd_all[sel], tgt_all[sel], t_rand=t_rand, u=u)
                    losses.append(loss2[0])
                assert tr.last_step_live == key.endswith('compacted')
                losses = torch.stack(losses).cpu().numpy()
                psnr[key].append(-10 * np.log10(np.mean(losses[-50:])))
                psnr[key + '_prefix'].append(-10 * np.log10(np.mean(losses[n_prefix - 10:n_prefix])))
            if seed >= n_full:
                continue
            sdc, sdf = init
            opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
            lc = []
            for it, (sel, t_rand, u) in enumerate(sched[:n_prefix]):
                opt.lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4
                l1, _, _, _ = O.train_step(sdc, sdf, opt, O.make_ray_batch(ro_all[sel].cpu(), rd_all[sel].cpu(), 2.0, 6.0), tgt_all[sel].cpu(),
                                           16, 16, True, t_rand=t_rand, u=u)
                lc.append(float(l1))
            psnr['oracle_prefix'].append(-10 * np.log10(np.mean(lc[-10:])))
    finally:
        fn.ops.set_math(old)
        fn.render.set_compact(old_c)
    m = {k: float(np.mean(v)) for k, v in psnr.items()}
    a32, a16, a6 = np.array(psnr['fp32']), np.array(psnr['bf16x3']), np.array(psnr['bf16x6'])
    # About one seed in six never leaves the empty-scene solution within 300 iterations (the SAME seeds in both modes: a property
    # of the initial weights; training PSNR 6.5 dB), and a few sit on the edge of it, where a rounding decides between 6 and 29 dB.
    # The modes are compared on the seeds whose runs train in BOTH modes; how many collapse must agree too.
    ok = (a32 > 20) & (a16 > 20)
    assert ok.sum() >= 0.7 * n_seeds and abs(int((a32 > 20).sum()) - int((a16 > 20).sum())) <= 3, (int((a32 > 20).sum()), int((a16 > 20).sum()))
    assert float(a32[ok].mean()) > float(np.mean(np.array(psnr['fp32_prefix'])[ok])) + 1.0   # 240 more iterations did train
    d = a16[ok] - a32[ok]                                                    # per-seed difference of the modes, plain backward
    se = float(np.std(d, ddof=1) / np.sqrt(len(d)))
    print('PSNR300 modes: %d of %d seeds train in both modes; mean fp32 %.3f, mean bf16x3 %.3f, mean per-seed difference %+.3f dB, '
          'per-seed std %.3f, standard error %.3f' % (len(d), n_seeds, a32[ok].mean(), a16[ok].mean(), float(d.mean()), float(np.std(d, ddof=1)), se))
    if study:
        assert se < 0.05, se                                                 # the comparison has the power to see 0.1 dB (2 standard errors; measured 0.042 - 0.045) ...
        assert abs(float(d.mean())) < 0.1, (float(d.mean()), se)             # ... and the modes agree within it (north_star)
    else:
        assert abs(float(d.mean())) < 0.1 + 2.0 * se, (float(d.mean()), se)  # regression form: a bias of the north_star's size would show
    # the headline arithmetic (bf16x6) against the fp32 FMA chain, same seeds, same statistic
    ok6 = (a32 > 20) & (a6 > 20)
    assert ok6.sum() >= 0.7 * n_seeds and abs(int((a32 > 20).sum()) - int((a6 > 20).sum())) <= 3, (int((a32 > 20).sum()), int((a6 > 20).sum()))
    d6 = a6[ok6] - a32[ok6]
    se6 = float(np.std(d6, ddof=1) / np.sqrt(len(d6)))
    print('PSNR300 bf16x6 - fp32: %d seeds, mean per-seed difference %+.3f dB, per-seed std %.3f, standard error %.3f' % (
        len(d6), float(d6.mean()), float(np.std(d6, ddof=1)), se6))
    if study:
        assert se6 < 0.05, se6
        assert abs(float(d6.mean())) < 0.1, (float(d6.mean()), se6)
    else:
        assert abs(float(d6.mean())) < 0.1 + 2.0 * se6, (float(d6.mean()), se6)
    for k in keys:                                                           # vs the oracle, on its prefix (before the divergence)
        assert abs(float(np.mean(psnr[k + '_prefix'][:n_full])) - m['oracle_prefix']) < 0.1, (k, m)
    # the noise floor: same arithmetic, other summation grouping (one seed moves by up to 0.4 dB, the 4-seed mean by 0.12-0.15 dB)
    for base in ('fp32', 'bf16x3'):
        assert abs(float(np.mean(psnr[base + '_compacted'])) - float(np.mean(psnr[base][:n_full]))) < 0.3, m


def test_psnr_300_iterations_both_modes(fn):
    """The regression form of _psnr_300_iterations: 32 seeds per mode (~40 s): collapse counts agree, the mean per-seed difference between the
    modes is within 0.1 dB + 2 standard errors, every mode follows the CPU oracle on its 40-iteration prefix, the compacted backward stays
    inside the noise floor.  The 128-seed study with the power to see 0.1 dB is test_psnr_300_iterations_study (`-m "gpu and slow"`)."""
    _psnr_300_iterations(fn, 32, False)


@pytest.mark.slow
def test_psnr_300_iterations_study(fn):
    """128 seeds per mode (~2.5 minutes): standard error of the mean per-seed difference < 0.05 dB (measured 0.042 - 0.045), |mean| < 0.1 dB (VERDICT r5 item 3:
    a statistical study is not a regression test)."""
    _psnr_300_iterations(fn, 128, True)


def test_psnr_vs_cpu_at_the_baseline_shape(fn, request):
    """The metric's second half at BASELINE configs[1]'s shape: 100 cameras of 800 x 800, 64 + 128 samples, 256 uniformly drawn
    rays per iteration (bench.PSNR_RAYS), 200 iterations, identical batches / injected t_rand, u / initial weights on the GPU and on the CPU oracle
    (bench.py's `psnr_vs_cpu` leg: the same functions).  Three statements:
      1. LOCKSTEP -- the GPU step taken from the CPU run's state before every iteration gives the CPU's loss (median 1e-6) and hence its
         PSNR window to < 0.01 dB (north_star's 0.1 dB with an order of magnitude to spare), in every math mode: there is no bias;
      2. the first iterations of the FREE runs agree to 1e-4 (they decorrelate later: chaotic trajectories);
      3. (reported, not asserted here) the free runs' PSNR at 200 iterations.  Round 3 bounded the CPU's value by 6 sigma of a 9-member
         ulp-jitter ensemble; round 4 showed that such an ensemble is NOT a yardstick for another arithmetic: its members stay closer to
         each other than to the un-jittered run (tools/psnr_jitter_study.py, profiles/r04_psnr_paired.md).  The falsifiable statement
         about free runs is the PAIRED test over initialisation seeds against a committed CPU ensemble:
         test_psnr_paired_with_the_cpu_ensemble_g22.
    The CPU run takes ~5 minutes of host time (PSNR_TEST_ITERS shortens it for local runs); tests/conftest.py starts it when the
    collection is known, so it runs beside the rest of the suite."""
    import json
    import conftest
    import bench as B
    run = getattr(request.config, '_psnr_cpu_run', None)
    if run is None:
        run = conftest.start_psnr_cpu_run()
    assert isinstance(run, dict), run
    _, _, K, _, new_trainer = conftest.psnr_protocol(fn)
    dev = torch.device('cuda')
    data, iters = run['data'], run['data']['iters']
    old, old_c = fn.ops.get_math(), fn.render.get_compact()
    try:
        dd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.items()}
        modes = (B.MAIN_MODE, 'fp32', 'bf16x3')
        free = {m: B.psnr_gpu_free(fn, dd, new_trainer, K, m) for m in modes}
        _, err = run['proc'].communicate()
        assert run['proc'].returncode == 0, err.decode()[-2000:]
        cpu = json.load(open(run['out']))
        states = np.load(run['out'] + '.states', mmap_mode='r')
        cpu_train = B.psnr_of(cpu['losses'], 20)
        cpu_held = -10.0 * np.log10(cpu['held_out_mse'])
        # 1. lockstep
        for mode in modes:
            r = B.psnr_gpu_lockstep(fn, dd, new_trainer, states, cpu['losses'], mode)
            print('PSNR-vs-CPU lockstep', mode, r, 'cpu', cpu_train)
            # (same weights, same batch: the typical iteration agrees to fp32 rounding; a late iteration can hold a ray whose
            # inverse-CDF sample sits on a bin edge -- DESIGN 5 (i) -- worth 1e-4..1e-3 of that batch's loss)
            assert r['median_rel_loss_diff'] < 1e-5 and r['max_rel_loss_diff'] < 5e-3, (mode, r)
            assert abs(r['train_psnr_db'] - cpu_train) < 0.01, (mode, r, cpu_train)
            assert r['max_rel_update_diff_l2'] < 1e-2, (mode, r)
        # 2. the free runs before they decorrelate
        n0 = min(10, iters)
        for mode in modes:
            g = np.asarray(free[mode][1][:n0])
            c = np.asarray(cpu['losses'][:n0])
            assert np.max(np.abs(g - c) / c) < 1e-4, (mode, g, c)
        # 3. (the free runs' PSNR at 200 iterations is a statement about distributions: test_psnr_paired_with_the_cpu_ensemble_g22)
        for key, cpu_v in (('train_psnr_db', cpu_train), ('held_out_psnr_db', cpu_held)):
            print('PSNR-vs-CPU free', key, 'cpu %.3f' % cpu_v, ' '.join('%s %.3f' % (m, free[m][0][key]) for m in modes))
    finally:
        fn.ops.set_math(old)
        fn.render.set_compact(old_c)


def _paired_runner(fn, P, data, iters, dev):
    """-> gpu_run(seed, mode, member): one free GPU run of the paired protocol from init_weights(seed) (member > 0: weights x (1 + 1e-6 N(0, 1)))."""
    dd = {k: v.to(dev) for k, v in data.items()}
    K = np.array([[P.FOCAL, 0, 0.5 * P.W], [0, P.FOCAL, 0.5 * P.H], [0, 0, 1]])
    args = fn.run_nerf.make_args(N_importance=P.N_IMPORTANCE, N_samples=P.N_SAMPLES, perturb=1.0, white_bkgd=True, no_reload=True,
                                 lrate=5e-4, lrate_decay=500)

    def gpu_run(seed, mode, member):
        fn.ops.set_math(mode)
        fn.render.set_compact('0')
        ktr, kte, _, _, _, _ = fn.run_nerf.create_nerf(args, device=dev)
        sdc, sdf = P.init_weights(seed)
        ktr['network_fn'].load_state_dict(sdc)
        ktr['network_fine'].load_state_dict(sdf)
        tr = fn.run_nerf.Trainer(ktr, P.H, P.W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
        if member:
            g = torch.Generator(device=dev).manual_seed(7000 + 10 * seed + member)
            with torch.no_grad():
                tr.flat.mul_(1.0 + 1e-6 * torch.randn(tr.flat.shape, generator=g, device=dev))
        tr.repack()
        ls = []
        for it in range(iters):
            ls.append(tr.step(dd['ro'][it], dd['rd'][it], dd['tgt'][it], t_rand=dd['t_rand'][it], u=dd['u'][it])[0][0])
        ls = torch.stack(ls).cpu().numpy()
        with torch.no_grad():
            rgb = fn.render.render(P.H, P.W, K, chunk=P.HELD_OUT, rays=(dd['ho_ro'], dd['ho_rd']), near=2.0, far=6.0, **kte)[0]
            mse = float(torch.mean((rgb - dd['ho_tgt']) ** 2))
        return P.psnr(np.mean(ls[-P.WINDOW:])), P.psnr(mse), float(ls[0])
    return gpu_run


def _check_inputs(z, data):
    digest = [float(data['ro'].double().sum()), float(data['tgt'].double().sum()), float(data['u'].double().sum())]
    # rays and jitter streams regenerate bit for bit from their seeds; the targets go through exp / cumprod of the host's vector math
    # library, which may differ in the last bit between CPU models (the per-seed first-loss check bounds what that is worth)
    assert np.allclose([digest[0], digest[2]], [z['input_digest'][0], z['input_digest'][2]], rtol=1e-12, atol=0) and \
        abs(digest[1] - z['input_digest'][1]) < 1e-7 * abs(z['input_digest'][1]), 'the inputs regenerated here are not the recorded run\'s'


G22_MEMBERS_MAIN, G22_MEMBERS_OTHER, G22_SEEDS_OTHER = 2, 2, 88
G22_FAST_SEEDS, G23_FAST_SEEDS = 40, 4      # the default `-m gpu` run: a fixed subset, one GPU member per seed


def _paired_stats(gv, cv):
    """gv [seeds, members], cv [seeds] -> (pairs that train on both sides, d = mean_members GPU - CPU of those, collapsed masks)."""
    alive_g, alive_c = gv[:, 0] > 15.0, cv > 15.0
    ok = alive_c & (gv > 15.0).all(1)
    return ok, gv[ok].mean(1) - cv[ok], alive_g, alive_c


def test_psnr_paired_with_the_cpu_ensemble_g22(fn, golden_dir):
    """north_star's "PSNR within 0.1 dB at equal iteration count" as a PAIRED regression test at the BASELINE shape (100 cameras of
    800 x 800, 64 + 128 samples, 200 iterations of 256 uniformly drawn rays): tests/golden/g22_psnr_cpu_ensemble.npz holds the CPU
    oracle's PSNR per initialisation seed (oracle/make_golden_psnr_ensemble.py); the GPU starts from the SAME weights
    (oracle/psnr_protocol.py init_weights), sees the SAME batches / t_rand / u (generated on the CPU from seeds, digest checked), and the
    statistic is the mean over seeds of d(s) = PSNR_gpu(s) - PSNR_cpu(s).

    This is the REGRESSION form (the first G22_FAST_SEEDS recorded seeds, one GPU run each, headline arithmetic only; ~30 s): asserted are
      (a) the first loss of every seed equals the CPU's to 2e-5 relative (same weights, same batch, same arithmetic class);
      (b) the seeds that collapse to the empty-scene solution (PSNR < 15 dB) are the same on both sides, at most one apart;
      (c) |mean(d)| < 0.1 + 2 SE for the training and the held-out PSNR -- a bias of the size the north_star excludes would show,
          a re-drawn chaotic trajectory (SE ~0.1 dB at 33 pairs) does not fail it.
    The statistical STUDY -- all 227 seeds x 2 members, fp32-MFMA sibling, confidence intervals -- is test_psnr_paired_study_g22 (`-m "gpu
    and slow"`); its results and the NULL distribution of the same design (CPU' - CPU, CPU' = the CPU oracle from weights x (1 + 1e-6 N(0, 1)))
    are profiles/r05_psnr_paired.md and profiles/r06_psnr_null.md."""
    import os
    from oracle import psnr_protocol as P
    z = np.load(os.path.join(golden_dir, 'g22_psnr_cpu_ensemble.npz'))
    seeds = [int(s) for s in z['seeds']]
    assert len(seeds) >= G22_FAST_SEEDS and [int(x) for x in z['protocol']] == [P.ITERS, P.RAYS, P.HELD_OUT, P.WINDOW, P.N_SAMPLES, P.N_IMPORTANCE]
    data = P.inputs(lambda o, d: fn.synthetic.render_rays(o, d, cutoff=0.0))
    _check_inputs(z, data)
    gpu_run = _paired_runner(fn, P, data, P.ITERS, torch.device('cuda'))
    old, old_c = fn.ops.get_math(), fn.render.get_compact()
    use = list(range(G22_FAST_SEEDS))
    try:
        runs = [gpu_run(seeds[i], 'bf16x6', 0) for i in use]
    finally:
        fn.ops.set_math(old)
        fn.render.set_compact(old_c)
    for i, r in zip(use, runs):
        assert abs(r[2] - float(z['first_loss'][i])) < 2e-5 * float(z['first_loss'][i]) + 1e-7, (seeds[i], r[2], float(z['first_loss'][i]))
    for k, (name, cv) in enumerate((('train', z['train_psnr_db'][use]), ('held-out', z['held_out_psnr_db'][use]))):
        gv = np.array([[r[k]] for r in runs])
        ok, d, alive_g, alive_c = _paired_stats(gv, cv)
        se = float(np.std(d, ddof=1) / np.sqrt(len(d)))
        print('G22 paired (regression subset) bf16x6 %s PSNR: %d of %d seeds train on both sides (collapsed: CPU %d, GPU %d, mismatching %d); '
              'mean difference %+.4f dB, per-seed std %.3f, SE %.4f' % (name, len(d), len(use), int((~alive_c).sum()), int((~alive_g).sum()),
                                                                        int((alive_g != alive_c).sum()), d.mean(), np.std(d, ddof=1), se))
        assert int((alive_g != alive_c).sum()) <= 1, (name, np.array(seeds)[use][alive_g != alive_c].tolist())
        assert len(d) >= 25 and abs(float(d.mean())) < 0.1 + 2.0 * se, (name, float(d.mean()), se)


@pytest.mark.slow
def test_psnr_paired_study_g22(fn, golden_dir):
    """The statistical study behind "PSNR within 0.1 dB at equal iteration count" (selected with `-m "gpu and slow"`; ~6 minutes of GPU):
    all recorded seeds of G22 (227), 2 GPU members per seed (member > 0: weights x (1 + 1e-6 N(0, 1))), headline arithmetic bf16x6 and the
    fp32-MFMA sibling on the first G22_SEEDS_OTHER seeds.  Pairing removes the 3 dB the initialisation moves a run's PSNR by; what is left per
    seed is the chaos of two free trajectories (std ~0.45 dB on either side), so the statement is about the MEAN and its standard error.
      bf16x6: (a) power: SE(mean d) <= 0.05 dB; (b) |mean(d)| < 0.1 dB; (c) mean +- 2 SE inside +-0.2 dB (margins of >= 2 SE -- the
              measured +0.040 +- 0.040 / +0.054 +- 0.039 dB of round 5 became +0.006 +- 0.037 / -0.023 +- 0.039 dB when round 6 regrouped the dW partial sums: every change of
              rounding re-draws the trajectories); (d) at most three mismatching collapses among 227 seeds (a seed on the edge of the empty-scene solution flips with the rounding:
              0 in round 5, 1 / 2 in round 6).
      fp32:   SE < 0.09, |mean| < 0.05 + 2.6 SE, at most two mismatching collapses (round 4's statement).
    What it cannot establish with 227 seeds is a 95 % interval inside +-0.1 dB (0.55 dB of per-seed scatter: ~590 seeds); what replaces that
    claim is the NULL experiment of profiles/r06_psnr_null.md (tests/test_psnr_null_golden.py): GPU - CPU is distributed like CPU' - CPU."""
    import os
    from oracle import psnr_protocol as P
    z = np.load(os.path.join(golden_dir, 'g22_psnr_cpu_ensemble.npz'))
    seeds = [int(s) for s in z['seeds']]
    assert len(seeds) >= 120 and [int(x) for x in z['protocol']] == [P.ITERS, P.RAYS, P.HELD_OUT, P.WINDOW, P.N_SAMPLES, P.N_IMPORTANCE]
    data = P.inputs(lambda o, d: fn.synthetic.render_rays(o, d, cutoff=0.0))
    _check_inputs(z, data)
    dev = torch.device('cuda')
    gpu_run = _paired_runner(fn, P, data, P.ITERS, dev)
    old, old_c = fn.ops.get_math(), fn.render.get_compact()
    checks = []
    try:
        for mode in ('bf16x6', 'fp32'):
            main = mode == 'bf16x6'
            use = list(range(len(seeds))) if main else list(range(min(G22_SEEDS_OTHER, len(seeds))))
            members = G22_MEMBERS_MAIN if main else G22_MEMBERS_OTHER
            g_train, g_held = [], []
            for i in use:
                runs = [gpu_run(seeds[i], mode, j) for j in range(members)]
                assert abs(runs[0][2] - float(z['first_loss'][i])) < 2e-5 * float(z['first_loss'][i]) + 1e-7      # same weights, same batch
                g_train.append([r[0] for r in runs]); g_held.append([r[1] for r in runs])
            out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
            if os.path.isdir(out_dir):      # (a record of the GPU side of the pairs for profiles/; not part of the test)
                np.savez(os.path.join(out_dir, 'g22_gpu_%s.npz' % mode), seeds=np.array(seeds)[use], train=np.array(g_train), held=np.array(g_held))
            for name, gv, cv in (('train', np.array(g_train), z['train_psnr_db'][use]), ('held-out', np.array(g_held), z['held_out_psnr_db'][use])):
                ok, d, alive_g, alive_c = _paired_stats(gv, cv)
                se = float(np.std(d, ddof=1) / np.sqrt(len(d)))
                lo, hi = float(d.mean()) - 2.0 * se, float(d.mean()) + 2.0 * se
                print('G22 paired %s %s PSNR: %d of %d seeds train on both sides (collapsed: CPU %d, GPU %d, mismatching %d); CPU mean %.3f, GPU mean '
                      '%.3f, mean difference %+.4f dB, per-seed std %.3f, SE %.4f, mean +- 2 SE [%+.4f, %+.4f]; within-seed GPU std %.3f' % (
                          mode, name, len(d), len(use), int((~alive_c).sum()), int((~alive_g).sum()), int((alive_g != alive_c).sum()), cv[ok].mean(),
                          gv[ok].mean(), d.mean(), np.std(d, ddof=1), se, lo, hi, float(np.mean(np.std(gv[ok], axis=1, ddof=1)))))
                n_mis = int((alive_g != alive_c).sum())
                if main:
                    checks.append((n_mis <= 3, (mode, name, 'collapsed seeds differ', np.array(seeds)[use][alive_g != alive_c].tolist())))
                    checks.append((len(d) >= 150 and se <= 0.05, (mode, name, 'too few pairs / too large a standard error for the 0.1 dB statement', len(d), se)))
                    checks.append((abs(float(d.mean())) < 0.1, (mode, name, 'the mean difference leaves +-0.1 dB', float(d.mean()), se)))
                    checks.append((lo > -0.2 and hi < 0.2, (mode, name, 'mean +- 2 SE leaves +-0.2 dB', float(d.mean()), se, lo, hi)))
                else:
                    checks.append((n_mis <= 2, (mode, name, alive_g.tolist(), alive_c.tolist())))
                    checks.append((len(d) >= 20 and se < 0.09, (mode, name, len(d), se)))
                    checks.append((abs(float(d.mean())) < 0.05 + 2.6 * se, (mode, name, float(d.mean()), se)))
    finally:
        fn.ops.set_math(old)
        fn.render.set_compact(old_c)
    assert all(ok for ok, _ in checks), [info for ok, info in checks if not ok]


def _g23(fn, golden_dir, n_seeds, members):
    import os
    from oracle import psnr_protocol as P
    path = os.path.join(golden_dir, 'g23_psnr_cpu_long.npz')
    if not os.path.exists(path):
        pytest.skip('G23 not recorded')
    z = np.load(path)
    seeds = [int(s) for s in z['seeds']]
    assert len(seeds) >= 4 and [int(x) for x in z['protocol']] == [P.LONG_ITERS, P.RAYS, P.HELD_OUT, P.WINDOW, P.N_SAMPLES, P.N_IMPORTANCE]
    data = P.inputs(lambda o, d: fn.synthetic.render_rays(o, d, cutoff=0.0), iters=P.LONG_ITERS, batch_seed=3)
    _check_inputs(z, data)
    use = list(range(len(seeds) if n_seeds is None else min(n_seeds, len(seeds))))
    gpu_run = _paired_runner(fn, P, data, P.LONG_ITERS, torch.device('cuda'))
    old, old_c = fn.ops.get_math(), fn.render.get_compact()
    try:
        g = np.array([[gpu_run(seeds[i], 'bf16x6', j)[:2] for j in range(members)] for i in use])      # [seed, member, (train, held-out)]
    finally:
        fn.ops.set_math(old)
        fn.render.set_compact(old_c)
    res = []
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir) and members > 1:      # (the study's GPU side for profiles/r06_psnr_null.md; not part of the test)
        np.savez(os.path.join(out_dir, 'g23_gpu_bf16x6.npz'), seeds=np.array(seeds)[use], train=g[:, :, 0], held=g[:, :, 1])
    for k, (name, cv) in enumerate((('train', z['train_psnr_db'][use]), ('held-out', z['held_out_psnr_db'][use]))):
        alive_c = cv > 15.0
        assert ((g[:, 0, k] > 15.0) == alive_c).all(), (name, g[:, 0, k].tolist(), cv.tolist())
        ok = alive_c & (g[:, :, k] > 15.0).all(1)
        d = g[ok, :, k].mean(1) - cv[ok]
        se = float(np.std(d, ddof=1) / np.sqrt(len(d)))
        print('G23 paired bf16x6 %s PSNR @ %d iterations: %d of %d seeds; CPU mean %.3f, GPU mean %.3f, mean difference %+.3f dB, per-seed std %.3f, SE %.3f' % (
            name, P.LONG_ITERS, len(d), len(use), cv[ok].mean(), g[ok, :, k].mean(), d.mean(), np.std(d, ddof=1), se))
        assert cv[ok].mean() > 28.0, cv[ok].mean()        # (G22's 200-iteration level is 26.5 / 27.8 dB)
        res.append((name, d, se))
    return res


def test_psnr_paired_long_horizon_g23(fn, golden_dir):
    """"PSNR@N-iters" at a second N, regression form: the paired protocol with 1000 iterations (batch seed 3) on the first G23_FAST_SEEDS
    recorded seeds (tests/golden/g23_psnr_cpu_long.npz, ~1 hour of one host core each), one GPU run per seed (~10 s).  At 41 dB two free
    trajectories differ by ~1 dB on the held-out rays, so with 3 - 4 pairs this only catches gross errors: every seed that trains on the CPU
    trains on the GPU and vice versa, 800 more iterations did raise the PSNR above G22's level, and |mean(GPU - CPU)| < 3 SE + 0.5 dB.  The
    20-seed study is test_psnr_paired_long_horizon_study_g23 (`-m "gpu and slow"`); its NULL counterpart profiles/r06_psnr_null.md."""
    for name, d, se in _g23(fn, golden_dir, G23_FAST_SEEDS, 1):
        assert len(d) >= 2 and abs(float(d.mean())) < 3.0 * se + 0.5, (name, float(d.mean()), se)


@pytest.mark.slow
def test_psnr_paired_long_horizon_study_g23(fn, golden_dir):
    """The 1000-iteration study: all recorded G23 seeds, 3 GPU members each (~3 minutes): the mean of GPU - CPU is within 3 standard errors
    + 0.15 dB of zero -- a consistency check, not an equivalence test."""
    for name, d, se in _g23(fn, golden_dir, None, 3):
        assert len(d) >= 3 and abs(float(d.mean())) < 3.0 * se + 0.15, (name, float(d.mean()), se)


def test_train_driver_with_quadtree(fn):
    imgs, poses, focal = fn.synthetic.make_dataset(n_images=4, H=32, W=32)
    torch.manual_seed(0)
    np.random.seed(0)
    args = fn.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, no_reload=True,
                                 N_rand=256, n_epoch=4, init_level=2, subdivide_every=1, subdivide_thres=0.05,
                                 lrate=5e-4, lrate_decay=500)
    logs = []
    ktr, kte, trainer, mgr, hist = fn.run_nerf.train(imgs, poses, 32, 32, focal, args, log=logs.append,
                                                     compat_rng=False)
    assert len(hist) == 4 and all(np.isfinite(h[2]) for h in hist)
    assert hist[-1][2] < hist[0][2]                       # loss went down over the epochs
    # trees were refined (epochs 1 and 2 subdivide; the last two do not: run_nerf.py:520)
    assert mgr.cur_level == 4 and max(mgr.num_leaves(i) for i in range(4)) > 4
    # last epoch uses every pixel once per image in expectation (tree.py:390-400)
    assert hist[-1][1] == (4 * 32 * 32 + 255) // 256
    # evaluation path renders a held-in view close to its target
    K = np.array([[focal, 0, 16.0], [0, focal, 16.0], [0, 0, 1]])
    rgbs, disps = fn.render.render_path(poses[:1], (32, 32, focal), K, 4096, dict(kte, near=2.0, far=6.0), gt_imgs=imgs[:1])
    assert rgbs.shape == (1, 32, 32, 3) and np.isfinite(rgbs).all()
    assert fn.render.render_path.last_psnrs[0] > 8.0


def test_optimizer_state_interchange(fn, tmp_path):
    """the fused trainer's Adam state round-trips through torch.optim.Adam's state_dict (the
    `optimizer_state_dict` entry of the reference's .tar), and a checkpoint written by train() resumes."""
    torch.manual_seed(0)
    args = fn.run_nerf.make_args(N_importance=8, N_samples=8, perturb=1.0, white_bkgd=True, no_reload=True)
    ktr, _, _, _, grad_vars, opt = fn.run_nerf.create_nerf(args, device='cuda')
    K = np.array([[40.0, 0, 8.0], [0, 40.0, 8.0], [0, 0, 1]])
    tr = fn.run_nerf.Trainer(ktr, 16, 16, K, 2.0, 6.0)
    ro = torch.randn(64, 3).cuda() * 0.1 + torch.tensor([0., 0., 4.]).cuda()
    rd = torch.randn(64, 3).cuda()
    tgt = torch.rand(64, 3).cuda()
    for _ in range(2):
        tr.step(ro, rd, tgt)
    sd = tr.torch_optimizer_state_dict()
    opt.load_state_dict(sd)                      # torch accepts it as an Adam state over grad_vars
    st = opt.state_dict()['state']
    assert len(st) == len(grad_vars) == 48 and int(float(st[0]['step'])) == 2
    off = 0
    for i, p in enumerate(grad_vars):
        k = p.numel()
        assert torch.equal(st[i]['exp_avg'].reshape(-1), tr.m[off:off + k])
        assert torch.equal(st[i]['exp_avg_sq'].reshape(-1), tr.v[off:off + k])
        off += k
    # a third step with torch's Adam on the fused gradients == the fused Adam step
    w_before = tr.flat.clone()
    tr.forward_backward(ro, rd, tgt)
    g = tr.grad.clone()
    tr2 = fn.run_nerf.Trainer(ktr, 16, 16, K, 2.0, 6.0)
    tr2.load_torch_optimizer(opt)
    assert tr2.adam_t == 2 and torch.equal(tr2.m, tr.m) and torch.equal(tr2.v, tr.v)
    off = 0
    for p in grad_vars:
        k = p.numel()
        p.grad = g[off:off + k].view(p.shape).clone()
        off += k
    opt.step()
    w_torch = tr.flat.clone()
    with torch.no_grad():
        tr.flat.copy_(w_before)
    tr.grad.copy_(g)
    tr.adam_t += 1
    fn.ops.adam_step(tr.flat, tr.grad, tr.m, tr.v, tr.lr, tr.adam_t, tr.beta1, tr.beta2, tr.eps)
    assert (tr.flat - w_torch).abs().max() < 2e-7


def test_resume_from_adam_state_with_missing_entries(fn, tmp_path):
    """torch.optim.Adam keeps no state for a parameter that never received a gradient (the reference's unused
    views_linears.0 without view directions, model.py:60-61): loading such an optimizer gives zero moments for it, and a
    single-pass model (N_importance = 0, network_fine None) checkpoints."""
    torch.manual_seed(0)
    args = fn.run_nerf.make_args(N_importance=0, N_samples=8, perturb=1.0, white_bkgd=True, no_reload=True, use_viewdirs=False,
                                 basedir=str(tmp_path), expname='x')
    ktr, _, _, _, grad_vars, opt = fn.run_nerf.create_nerf(args, device='cuda')
    K = np.array([[40.0, 0, 8.0], [0, 40.0, 8.0], [0, 0, 1]])
    tr = fn.run_nerf.Trainer(ktr, 16, 16, K, 2.0, 6.0)
    ro = torch.randn(32, 3).cuda() * 0.1 + torch.tensor([0., 0., 4.]).cuda()
    rd, tgt = torch.randn(32, 3).cuda(), torch.rand(32, 3).cuda()
    # the reference's optimizer after two steps: gradients for everything but views_linears.0
    names = [n for n, _ in ktr['network_fn'].named_parameters()]
    for _ in range(2):
        tr.forward_backward(ro, rd, tgt)
        for n, p in zip(names, grad_vars):
            p.grad = None if n.startswith('views_linears') else p.grad.clone()
        opt.step()
    sd = opt.state_dict()
    assert len(sd['state']) == len(grad_vars) - 2
    tr2 = fn.run_nerf.Trainer(ktr, 16, 16, K, 2.0, 6.0)
    tr2.load_torch_optimizer(sd)
    assert tr2.adam_t == 2
    off = 0
    for i, (n, p) in enumerate(zip(names, grad_vars)):
        k = p.numel()
        if n.startswith('views_linears'):
            assert float(tr2.m[off:off + k].abs().max()) == 0.0 and float(tr2.v[off:off + k].abs().max()) == 0.0
        else:
            assert torch.equal(tr2.m[off:off + k], sd['state'][i]['exp_avg'].reshape(-1))
        off += k
    mgr = fn.tree.QuadTreeManager(16, 16, K, torch.rand(2, 16, 16, 3), torch.eye(4)[None, :3].repeat(2, 1, 1), mseThres=0.0, max_depth=1,
                                  device='cuda')
    path = fn.run_nerf.save_checkpoint(args, 1, tr2, ktr, mgr)
    ck = torch.load(path, weights_only=False)
    assert ck['network_fine_state_dict'] is None and 'module.output_linear.weight' in ck['network_fn_state_dict']
