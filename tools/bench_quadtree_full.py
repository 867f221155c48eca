"""BASELINE configs[2] AT THE REFERENCE'S SCALE (nerf-ours/configs/lego.txt:1-32): 100 views of 800 x 800, init_level 2, subdivide_every 3,
subdivide_thres 1e-3, n_epoch 18 -- the quadtree goes from 4 leaves per image to 4^6 = 4096 (five subdivisions, epochs 3 .. 15) -- 4096
rays x (64 + 128) samples per step at the headline arithmetic, with the reference's quadtree cost on the host cores beside it.

What is timed per epoch (seconds, GPU synchronised around each phase):
  gen     QuadTreeManager.gen_rays_v3_multiThread(compat_rng=False): the host's leaf plans + ONE device launch that draws the WHOLE epoch
          (100 x 640 000 rays: origins, directions, colours, (image, leaf) tags)                            -- tree.py:377-428
  steps   the fused training steps feeding the on-device per-(image, leaf) max-error table                  -- run_nerf.py:479-508
  adjust  table -> host -> native tree adjustment (fastnerf_tree_adjust) -> new leaf lists                  -- tree.py:533-557, 629-652
SUB-SAMPLING (stated, not hidden): an epoch of the reference trains on all 64 M rays (15 625 steps of 4096; 4.7 minutes at 225 k rays/s, 85
minutes for the 18 epochs).  This leg GENERATES every epoch in full and TRAINS on its first `--steps-subdivide` batches in the epochs whose
table drives a subdivision (default 1000 steps = 4.1 M rays = 6.4 % of the epoch: >= 10 rays per finest leaf at every level, so every leaf
is seen) and on `--steps-other` batches in the others (their table is discarded by the reference too).  The rays of an epoch are shuffled,
so a prefix is a uniform sample.  gen and adjust are NOT sub-sampled: they are the full-scale costs.

CPU baseline (`cpu_quadtree`): oracle/tree_oracle.py -- the restatement of the reference's host-side gen / adjust, pinned by G9 -- timed on
`--cpu-images` images at the finest level reached (gen at 4096 leaves per image; adjust 1024 -> 4096 leaves over an image's 640 000 rays) and
scaled to 100 images.  The reference itself reports 0.41 s gen and 1.7 s / image adjust.

  python tools/bench_quadtree_full.py [--views 100] [--res 800] [--out gpurun_out/r05_quadtree_full.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)


def cpu_quadtree(H, W, n_img, level, thres):
    """The oracle's gen + adjust at the finest transition (level - 1 -> level) on n_img images; seconds per image."""
    from oracle import tree_oracle as TO
    torch.manual_seed(0)
    mgr = TO.Manager(H, W, n_img, level - 1)
    t0 = time.perf_counter()
    pix = mgr.gen_pixels(1, False)
    t_gen_coarse = time.perf_counter() - t0
    n = pix.shape[0]
    gt, pred = torch.rand(n, 3), torch.rand(n, 3)       # (max |gt - pred| > thres for every leaf: every finest leaf splits, as in the GPU run's early epochs)
    t0 = time.perf_counter()
    mgr.adjust(gt, pred, thres)
    t_adjust = time.perf_counter() - t0
    leaves = [len(t.leaves) for t in mgr.trees]
    t0 = time.perf_counter()
    pix = mgr.gen_pixels(1, False)
    t_gen = time.perf_counter() - t0
    return {'images_timed': n_img, 'level': level, 'leaves_per_image_after': int(max(leaves)), 'rays_per_image': int(pix.shape[0] // n_img),
            'gen_seconds_per_image': t_gen / n_img, 'gen_seconds_per_image_one_level_up': t_gen_coarse / n_img,
            'adjust_seconds_per_image': t_adjust / n_img,
            'what': 'oracle/tree_oracle.py Manager.gen_pixels / Manager.adjust (restatement of nerf-ours/tree.py:377-428, 533-557, 629-652; one thread, '
                    'torch CPU); per image -- multiply by the number of views for an epoch'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=100)
    ap.add_argument('--res', type=int, default=800)
    ap.add_argument('--steps-subdivide', type=int, default=1000)
    ap.add_argument('--steps-other', type=int, default=100)
    ap.add_argument('--warmup-steps', type=int, default=300)
    ap.add_argument('--cpu-images', type=int, default=2)
    ap.add_argument('--n-epoch', type=int, default=18)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r05_quadtree_full.json'))
    a = ap.parse_args()
    import fastnerf as fn
    from fastnerf import ops
    dev = torch.device('cuda')
    H = W = a.res
    N_RAND, THRES, INIT_LEVEL, EVERY = 4096, 1e-3, 2, 3
    ops.set_math('bf16x6')
    fn.render.set_compact('0')
    t0 = time.perf_counter()
    imgs, poses, focal = fn.synthetic.make_dataset(n_images=a.views, H=H, W=W, device='cuda')
    torch.cuda.synchronize()
    t_data = time.perf_counter() - t0
    torch.manual_seed(0); np.random.seed(0)
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, N_rand=N_RAND, n_epoch=a.n_epoch,
                                 init_level=INIT_LEVEL, subdivide_every=EVERY, subdivide_thres=THRES, lrate=5e-4, lrate_decay=500)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    kw_train = fn.run_nerf.create_nerf(args, device=dev)[0]
    kw_train.update(near=2.0, far=6.0)
    trainer = fn.run_nerf.Trainer(kw_train, H, W, K, 2.0, 6.0, lrate=args.lrate, lrate_decay=args.lrate_decay)
    mgr = fn.tree.QuadTreeManager(H, W, K, imgs, poses[:, :3, :4], mseThres=0.0, max_depth=INIT_LEVEL, device=dev)

    def run(ro, rd, tgt, tags, table, ml, n_steps, decay=True):
        n_total, it, loss2 = ro.shape[0], 0, None
        for b0 in range(0, n_total, N_RAND):
            sl = slice(b0, min(b0 + N_RAND, n_total))
            loss2, _ = trainer.step(ro[sl], rd[sl], tgt[sl], leaf_tag=None if tags is None else tags[sl], table=table, max_leaves=ml, decay=decay)
            it += 1
            if it >= n_steps:
                break
        return loss2, it

    # warm-up on uniformly drawn rays (the reference's centre-crop warm-up is 500 batches: run_nerf.py:367-423), untimed
    g = torch.Generator().manual_seed(1)
    pix = torch.stack([torch.randint(0, a.views, (a.warmup_steps * N_RAND,), generator=g), torch.randint(H // 4, 3 * H // 4, (a.warmup_steps * N_RAND,), generator=g),
                       torch.randint(W // 4, 3 * W // 4, (a.warmup_steps * N_RAND,), generator=g)], 1)
    ro, rd, tgt = mgr.gather(pix)
    run(ro, rd, tgt, None, None, 0, a.warmup_steps, decay=False)
    torch.cuda.synchronize()
    epochs = []
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
        table = torch.zeros(mgr.n_images * ml, device=dev, dtype=torch.int32)
        n_steps = a.steps_subdivide if subdiv else a.steps_other
        t0 = time.perf_counter()
        loss2, it = run(ro, rd, tgt, tags, table, ml, n_steps)
        torch.cuda.synchronize(); t_steps = time.perf_counter() - t0
        rec = {'epoch': ep, 'rays_generated': int(ro.shape[0]), 'gen_seconds': t_gen, 'steps': it, 'steps_seconds': t_steps,
               'rays_per_s_steps': it * N_RAND / t_steps, 'leaves_per_image_max': int(ml),
               'leaves_total': int(sum(mgr.num_leaves(i) for i in range(mgr.n_images))), 'psnr_db': float(-10 * np.log10(float(loss2[0]))),
               'sampled_fraction_of_epoch': it * N_RAND / float(ro.shape[0])}
        if subdiv:
            seen = int((table != 0).sum())
            t0 = time.perf_counter()
            tot = mgr.adjust_tree_from_table(table.view(mgr.n_images, ml), thres=THRES)
            torch.cuda.synchronize()
            rec.update(adjust_seconds=time.perf_counter() - t0, leaves_total_after=int(tot), leaves_with_a_ray=seen,
                       leaves_per_image_max_after=int(mgr.max_leaves()))
        epochs.append(rec)
        print(json.dumps(rec), flush=True)
        del ro, rd, tgt, tags, table
    level = INIT_LEVEL + sum(1 for e in epochs if 'adjust_seconds' in e)
    cpu = cpu_quadtree(H, W, a.cpu_images, level, THRES)
    sub = [e for e in epochs if 'adjust_seconds' in e]
    finest = sub[-1]
    gen_finest = [e for e in epochs if e['epoch'] > finest['epoch'] and e['epoch'] < a.n_epoch]
    out = {
        'workload': 'BASELINE configs[2] at the reference\'s scale (nerf-ours/configs/lego.txt): %d analytic views of %dx%d, init_level 2, subdivide_every 3, '
                    'subdivide_thres 1e-3, n_epoch %d, 4096 rays x (64+128) samples per step, bf16x6, plain backward; device ray generation' % (a.views, H, W, a.n_epoch),
        'sub_sampling': 'every epoch is GENERATED in full (%d rays); trained on its first %d batches in subdividing epochs, %d in the others (a uniform sample: '
                        'the epoch is shuffled); gen / adjust are full-scale costs' % (a.views * H * W, a.steps_subdivide, a.steps_other),
        'dataset_seconds_untimed': t_data, 'epochs': epochs, 'level_reached': level, 'leaves_max': int(mgr.max_leaves()),
        'gpu': {'gen_seconds_full_epoch_at_finest_level': float(np.mean([e['gen_seconds'] for e in gen_finest])) if gen_finest else None,
                'adjust_seconds_100_images_at_finest_transition': finest['adjust_seconds'],
                'adjust_seconds_per_image': finest['adjust_seconds'] / a.views,
                'steps_rays_per_s_mean': float(np.mean([e['rays_per_s_steps'] for e in epochs if e['steps'] >= 100]))},
        'cpu_quadtree': cpu,
        'cpu_vs_gpu': {'gen_full_epoch': None if not gen_finest else cpu['gen_seconds_per_image'] * a.views / float(np.mean([e['gen_seconds'] for e in gen_finest])),
                       'adjust_finest_transition': cpu['adjust_seconds_per_image'] * a.views / finest['adjust_seconds']},
        'reference_reported': {'gen_seconds': 0.41, 'adjust_seconds_per_image': 1.7, 'source': 'BASELINE.md section 2 (the reference\'s own log lines)'},
    }
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('level_reached', 'leaves_max', 'gpu', 'cpu_quadtree', 'cpu_vs_gpu')}))


if __name__ == '__main__':
    main()
