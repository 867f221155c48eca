"""BASELINE configs[2] at the reference's scale WHERE THE TREE ADAPTS (nerf-ours/tree.py:377-428, 578-581, 629-652,
run_nerf.py:437-452): 100 views of 800 x 800, init_level 2, subdivide_every 3, n_epoch 18, but a subdivide_thres at which only PART of the finest
leaves split at every adjust -- profiles/r05_quadtree_full.json ran lego.txt's 1e-3, under which every leaf of the analytic scene splits every
time (400 -> 409 588 leaves) and an epoch stays at 64 M rays.  The paper's point is the other regime: a leaf that stops splitting stays
coarse and receives 10 rays per epoch instead of area x rays-per-pixel (tree.py:578-581), so the epoch shrinks to the pixels that still need work.

Scene: the analytic bodies with their density cut to EXACTLY zero beyond 1.5 sigma (--cutoff), i.e. solid objects in empty space on a white
background -- the situation of the reference's Blender scenes, where a background pixel's target is exactly 1.0 and a net with sigma <= 0 along
the ray reproduces it exactly (alpha = 0, rgb = 1 - acc = 1): max |gt - pred| of a background leaf becomes EXACTLY 0 and lego.txt's subdivide_thres
1e-3 separates it from a leaf that sees a body.  (The Gaussian-tail scene of the r05 run has no exactly-empty pixel: everything splits.)
--thres (default lego.txt's 1e-3) is held fixed like the reference's argument; --thres 0 CALIBRATES it once, at the first adjust, as the --quantile
of the finest leaves' table.  Reported per epoch: rays generated (the whole epoch), leaves per image min / median / max, gen seconds;
per adjust: seconds, fraction of the finest leaves that split; at the end: the leaf-area histogram, and two checks against
oracle/tree_oracle.py (the restatement pinned by G9) on --check-images images: (a) every adjust's split decisions from the device table,
(b) the final per-leaf plan (ray count incl. the 10-ray rule, integer pixel ranges).  Sub-sampling of the TRAINING as in bench_quadtree_full.py
(stated there); generation, adjustment and the checks are full scale.

  python tools/bench_quadtree_adaptive.py [--views 100] [--res 800] [--out gpurun_out/r06_quadtree_adaptive.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=100)
    ap.add_argument('--res', type=int, default=800)
    ap.add_argument('--steps-subdivide', type=int, default=1000)
    ap.add_argument('--steps-other', type=int, default=100)
    ap.add_argument('--warmup-steps', type=int, default=300)
    ap.add_argument('--n-epoch', type=int, default=18)
    ap.add_argument('--quantile', type=float, default=0.5)
    ap.add_argument('--thres', type=float, default=1e-3, help='subdivide_thres (lego.txt: 1e-3); 0: calibrate at the first adjust (--quantile)')
    ap.add_argument('--check-images', type=int, default=2)
    ap.add_argument('--full', action='store_true', help='NO sub-sampling: every batch of every epoch is trained (the reference\'s run; ~1 GPU-hour)')
    ap.add_argument('--always-split', action='store_true', help='the NON-adaptive member of the pair: thres -1, every finest leaf splits at every adjust, every epoch is every pixel')
    ap.add_argument('--eval-views', type=int, default=0, help='after training: PSNR of this many HELD-OUT full-resolution views (cameras half-way between training views, '
                    'another elevation) through render() with the test kwargs')
    ap.add_argument('--save', default='', help='write the final parameters (both nets, flat fp32) here (.pt)')
    ap.add_argument('--compact', default='0', help="FASTNERF_COMPACT policy of the steps: '0' plain backward (the headline protocol), 'auto' the product default")
    ap.add_argument('--cutoff', type=float, default=1.5, help='scene: density exactly zero beyond this many sigma of a blob (solid bodies in EMPTY space, '
                    'like the Lego bulldozer on its white background: 27 %% of the pixels covered); 0 = Gaussian tails that never vanish')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r06_quadtree_adaptive.json'))
    a = ap.parse_args()
    import fastnerf as fn
    from fastnerf import ops
    from oracle import tree_oracle as TO          # (a tool, not the product: the checker)
    dev = torch.device('cuda')
    H = W = a.res
    N_RAND, INIT_LEVEL, EVERY = 4096, 2, 3
    ops.set_math('bf16x6')
    fn.render.set_compact(a.compact)
    if a.full:
        a.steps_subdivide = a.steps_other = 10 ** 9
    if a.always_split:
        a.thres = -1.0
    fn.synthetic.CUTOFF = a.cutoff
    imgs, poses, focal = fn.synthetic.make_dataset(n_images=a.views, H=H, W=W, device='cuda')
    covered = float((imgs < 1.0).any(-1).float().mean())      # pixels that see a body (the others are EXACTLY white)
    torch.manual_seed(0); np.random.seed(0)
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, N_rand=N_RAND, n_epoch=a.n_epoch,
                                 init_level=INIT_LEVEL, subdivide_every=EVERY, subdivide_thres=1e-3, lrate=5e-4, lrate_decay=500)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    kw_train, kw_test = fn.run_nerf.create_nerf(args, device=dev)[:2]
    kw_test.update(near=2.0, far=6.0)
    kw_train.update(near=2.0, far=6.0)
    trainer = fn.run_nerf.Trainer(kw_train, H, W, K, 2.0, 6.0, lrate=args.lrate, lrate_decay=args.lrate_decay)
    mgr = fn.tree.QuadTreeManager(H, W, K, imgs, poses[:, :3, :4], mseThres=0.0, max_depth=INIT_LEVEL, device=dev)
    chk = list(range(min(a.check_images, a.views)))
    omgr = TO.Manager(H, W, len(chk), INIT_LEVEL)      # the oracle's trees of the checked images, adjusted from the same device tables

    def run(ro, rd, tgt, tags, table, ml, n_steps, decay=True):
        n_total, it, loss2 = ro.shape[0], 0, None
        acc.zero_()
        for b0 in range(0, n_total, N_RAND):
            sl = slice(b0, min(b0 + N_RAND, n_total))
            loss2, _ = trainer.step(ro[sl], rd[sl], tgt[sl], leaf_tag=None if tags is None else tags[sl], table=table, max_leaves=ml, decay=decay)
            acc.add_(loss2[:2].detach().to(acc.dtype))
            it += 1
            if it >= n_steps:
                break
        return loss2, it

    acc = torch.zeros(2, device=dev, dtype=torch.float64)      # the epoch's summed [fine, coarse] batch MSEs (one tiny launch per step, no host sync)
    g = torch.Generator().manual_seed(1)
    pix = torch.stack([torch.randint(0, a.views, (a.warmup_steps * N_RAND,), generator=g), torch.randint(H // 4, 3 * H // 4, (a.warmup_steps * N_RAND,), generator=g),
                       torch.randint(W // 4, 3 * W // 4, (a.warmup_steps * N_RAND,), generator=g)], 1)
    ro, rd, tgt = mgr.gather(pix)
    run(ro, rd, tgt, None, None, 0, a.warmup_steps, decay=False)
    torch.cuda.synchronize()
    thres, epochs, adjust_checks = (a.thres if a.thres != 0 else None), [], []
    for ep in range(1, a.n_epoch + 1):
        last = ep == a.n_epoch
        subdiv = ep % EVERY == 0 and ep < a.n_epoch - 1
        if last:
            mgr.epoch_size = mgr.n_images * mgr.h * mgr.w
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ro, rd, tgt = mgr.gen_rays_v3_multiThread(down_scale=1, prob=False, randSamp_proc=1.0, last_epoch=last, compat_rng=False)
        tags = mgr.result_leaf_tag
        torch.cuda.synchronize(); t_gen = time.perf_counter() - t0
        ml = mgr.max_leaves()
        nl = np.array([mgr.num_leaves(i) for i in range(mgr.n_images)])
        table = torch.zeros(mgr.n_images * ml, device=dev, dtype=torch.int32)
        n_steps = a.steps_subdivide if subdiv else a.steps_other
        t0 = time.perf_counter()
        loss2, it = run(ro, rd, tgt, tags, table, ml, n_steps)
        torch.cuda.synchronize(); t_steps = time.perf_counter() - t0
        rec = {'epoch': ep, 'last_epoch_full_images': bool(last), 'rays_generated': int(ro.shape[0]), 'fraction_of_all_pixels': ro.shape[0] / float(a.views * H * W),
               'backward': 'compacted' if trainer.last_step_live else 'plain', 'live_fraction': trainer.live.frac,
               'gen_seconds': t_gen, 'steps': it, 'steps_seconds': t_steps, 'rays_per_s_steps': it * N_RAND / t_steps,
               'leaves_per_image': {'min': int(nl.min()), 'median': float(np.median(nl)), 'max': int(nl.max())}, 'leaves_total': int(nl.sum()),
               'psnr_db': float(-10 * np.log10(float(loss2[0]))), 'psnr_db_epoch_mean_mse': float(-10 * np.log10(float(acc[0]) / it)), 'sampled_fraction_of_epoch': min(1.0, it * N_RAND / float(ro.shape[0]))}
        if subdiv:
            tab = table.view(mgr.n_images, ml).view(torch.float32)
            finest = torch.zeros(mgr.n_images, ml, dtype=torch.bool)
            for i in range(mgr.n_images):
                lv = np.asarray(mgr.leaves(i))
                area = (lv[:, 2] - lv[:, 0]) * (lv[:, 3] - lv[:, 1])
                finest[i, :len(lv)] = torch.from_numpy(area == mgr.min_area(i))
            vals = tab.cpu()[finest]
            if thres is None:
                thres = float(torch.quantile(vals[vals > 0].double(), a.quantile))
                rec['threshold_calibrated_here'] = thres
            will = int((vals > thres).sum())
            t0 = time.perf_counter()
            tot = mgr.adjust_tree_from_table(table.view(mgr.n_images, ml), thres=thres)
            torch.cuda.synchronize()
            t_adj = time.perf_counter() - t0
            # oracle on the checked images: the same decisions from the same table
            omgr.adjust_from_table([tab[i].cpu().numpy() for i in chk], thres)
            same = all(np.array_equal(np.asarray(mgr.leaves(i)), omgr.leaf_array(k)) for k, i in enumerate(chk))
            adjust_checks.append(bool(same))
            rec.update(adjust_seconds=t_adj, threshold=thres, finest_leaves=int(finest.sum()), finest_leaves_with_a_ray=int((vals > 0).sum()),
                       finest_leaves_split=will, fraction_of_finest_split=will / max(1, int(finest.sum())), leaves_total_after=int(tot),
                       table_quantiles_finest={q: float(torch.quantile(vals.double(), q)) for q in (0.1, 0.25, 0.5, 0.75, 0.9)},
                       oracle_agrees_on_checked_images=bool(same))
        epochs.append(rec)
        print(json.dumps(rec), flush=True)
        del ro, rd, tgt, tags, table
    # ---- held-out views (never trained on): cameras half-way between training views at another elevation ----------------------------
    held = None
    if a.eval_views > 0:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        def eval_view(theta, phi):
            c2w = fn.synthetic.pose_spherical(theta, phi, 4.0)[:3, :4].to(dev)
            rgb = fn.render.render(H, W, K, chunk=1 << 16, c2w=c2w, **kw_test)[0]
            ro_e, rd_e = ops.gen_rays(H, W, K, c2w)
            fo, fd = ro_e.reshape(-1, 3), rd_e.reshape(-1, 3)
            gt = torch.cat([fn.synthetic.render_rays(fo[s:s + 65536], fd[s:s + 65536], n_quad=256) for s in range(0, H * W, 65536)], 0).reshape(H, W, 3)
            return float(-10 * torch.log10(((rgb - gt) ** 2).mean()))
        # off the ring: another elevation (the training cameras all sit at -30 deg: view directions the nets never saw); on the ring: the
        # training elevation, azimuths half-way between two training cameras (interpolation)
        vals = [eval_view(-180.0 + 360.0 * (k + 0.5) / a.eval_views, -20.0) for k in range(a.eval_views)]
        ring = [eval_view(-180.0 + 360.0 * (k * (a.views // a.eval_views) + 0.5) / a.views, -30.0) for k in range(a.eval_views)]
        torch.cuda.synchronize()
        held = {'views': a.eval_views, 'psnr_db_mean': float(np.mean(vals)), 'psnr_db_min': float(np.min(vals)), 'psnr_db_max': float(np.max(vals)), 'psnr_db': vals,
                'on_ring_psnr_db_mean': float(np.mean(ring)), 'on_ring_psnr_db': ring,
                'seconds': time.perf_counter() - t0, 'what': 'full %dx%d views; psnr_db*: elevation -20 deg (training: -30), azimuths between the training cameras; on_ring_*: the training '
                                                               'elevation, azimuths half-way between two neighbouring training cameras; render() with the test kwargs (perturb 0), '
                                                               'targets by the 256-step quadrature of the scene' % (H, W)}
        print(json.dumps({'held_out': held}), flush=True)
    if a.save:
        torch.save({'flat': trainer.flat.detach().cpu(), 'what': 'coarse | fine parameters, flat fp32 in model.parameters() order (run_nerf.create_nerf)'}, a.save)
    # ---- final state: leaf-area histogram and the per-leaf plan against the oracle ------------------------------------------------
    plan, n_rays = mgr.epoch_plan(1, False)
    hist = {}
    for i in range(mgr.n_images):
        lv = np.asarray(mgr.leaves(i))
        for ar in ((lv[:, 2] - lv[:, 0]) * (lv[:, 3] - lv[:, 1])).tolist():
            hist[ar] = hist.get(ar, 0) + 1
    depth_of = {ar: 1 + int(round(np.log(H * W / ar) / np.log(4))) for ar in hist}
    plan_ok, ten = True, 0
    for k, i in enumerate(chk):
        rows = plan[plan[:, 0] == i]
        tr = omgr.trees[k]
        assert len(rows) == len(tr.leaves)
        for li, b in enumerate(tr.leaves):
            n = TO.leaf_ray_num(tr, b, 1.0)
            r0, r1, c0, c1 = TO.leaf_pixel_range(b)
            ten += n == 10
            plan_ok = plan_ok and rows[li].tolist() == [i, li, n, r0, r1, c0, c1]
    sub = [e for e in epochs if 'adjust_seconds' in e]
    out = {
        'workload': 'BASELINE configs[2] at the reference\'s scale, ADAPTIVE regime: %d analytic views of %dx%d, init_level 2, subdivide_every 3, n_epoch %d, 4096 rays x '
                    '(64+128) samples per step, bf16x6, %s, device ray generation; scene cutoff %.1f sigma (%.1f %% of the pixels see a body, the rest are '
                    'exactly white); subdivide_thres %s' % (a.views, H, W, a.n_epoch, 'plain backward' if a.compact == '0' else 'backward policy %r (exact zero-gradient compaction when it pays)' % a.compact,
                                                            a.cutoff, 100 * covered,
                                                            ('-1: EVERY finest leaf splits' if a.thres < 0 else '%g (fixed)' % a.thres) if a.thres != 0 else 'calibrated once at the first adjust (quantile %.2f)' % a.quantile),
        'pixels_covered_by_a_body': covered,
        'threshold': thres, 'epochs': epochs, 'sub_sampling': 'NONE: every batch of every epoch trained' if a.full else 'training sub-sampled as in bench_quadtree_full.py',
        'member': 'NON-adaptive (thres -1: every finest leaf splits, every epoch is every pixel)' if a.always_split else 'adaptive', 'held_out': held,
        'total_gen_seconds': float(sum(e['gen_seconds'] for e in epochs)), 'total_adjust_seconds': float(sum(e.get('adjust_seconds', 0.0) for e in epochs)),
        'compact_policy': a.compact, 'total_training_seconds': float(sum(e['steps_seconds'] for e in epochs)), 'total_rays_trained': int(sum(e['steps'] for e in epochs)) * N_RAND,
        'final': {'depth_reached': max(depth_of.values()), 'leaves_per_image': epochs[-2]['leaves_per_image'], 'leaves_total': int(sum(hist.values())),
                  'leaves_by_depth': {str(depth_of[ar]): n for ar, n in sorted(hist.items(), reverse=True)},
                  'rays_per_epoch': int(n_rays), 'fraction_of_all_pixels': n_rays / float(a.views * H * W),
                  'rays_per_epoch_if_every_leaf_had_split': a.views * H * W},
        'gpu': {'gen_seconds_last_adaptive_epoch': epochs[-2]['gen_seconds'], 'adjust_seconds': [e['adjust_seconds'] for e in sub],
                'fraction_of_finest_split': [e['fraction_of_finest_split'] for e in sub],
                'steps_rays_per_s_mean': float(np.mean([e['rays_per_s_steps'] for e in epochs if e['steps'] >= 100]))},
        'oracle_checks': {'images': chk, 'adjust_decisions_identical_at_every_adjust': adjust_checks, 'final_plan_identical': bool(plan_ok),
                          'leaves_on_the_10_ray_rule_in_checked_images': int(ten),
                          'what': 'oracle/tree_oracle.py (pinned by G9): adjust_from_table on the device table; leaf_ray_num (tree.py:578-581) and leaf_pixel_range '
                                  '(tree.py:598-599) against fastnerf_tree_epoch_plan rows'},
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('threshold', 'final', 'gpu', 'oracle_checks')}))
    assert plan_ok and all(adjust_checks), 'the native tree and the oracle disagree'


if __name__ == '__main__':
    main()
