#!/usr/bin/env python3
"""The fused optimisation step (bench.py's headline protocol: bf16x6, plain backward, leaf table on) at the reference's own batch sizes and at
an 8-way strong-scaling shard: N = 512 (4096 / 8), 1024 (9 of the 16 shipped configs), 1920 (lego.txt:16), 4096 (BASELINE configs[1]).
  python tools/time_step_sizes.py                 ms / step, rays/s and the fraction of the 4096-ray rate per size (HIP events, 3 rounds)
  python tools/time_step_sizes.py steps N K       exactly K steps of N rays (for rocprofv3 --kernel-trace --stats)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf  # noqa: E402
from fastnerf import ops, synthetic  # noqa: E402

NS, NI = 64, 128
H = W = 800
FOCAL = 0.5 * W / np.tan(0.5 * 0.6911112070083618)


def setup(n_rays, mode='bf16x6'):
    dev = torch.device('cuda')
    ops.set_math(mode)
    fastnerf.render.set_compact('0')
    args = fastnerf.run_nerf.make_args(N_importance=NI, N_samples=NS, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4,
                                       lrate_decay=500, N_rand=n_rays)
    K = np.array([[FOCAL, 0, 0.5 * W], [0, FOCAL, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
    gen = torch.Generator().manual_seed(1000)
    batches = []
    for _ in range(16):
        pix = torch.stack([torch.randint(0, 100, (n_rays,), generator=gen), torch.randint(0, H, (n_rays,), generator=gen),
                           torch.randint(0, W, (n_rays,), generator=gen)], 1).int()
        ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
        tag = torch.stack([pix[:, 0], (pix[:, 1] // 50) * 16 + pix[:, 2] // 50], 1).int().to(dev).contiguous()
        batches.append((ro, rd, torch.rand(n_rays, 3, generator=gen).to(dev), tag))
    torch.manual_seed(0)
    ktr = fastnerf.run_nerf.create_nerf(args, device=dev)[0]
    tr = fastnerf.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    table = torch.zeros(100 * 256, device=dev, dtype=torch.int32)

    def step(i):
        ro, rd, tgt, tag = batches[i % 16]
        return tr.step(ro, rd, tgt, leaf_tag=tag, table=table, max_leaves=256)
    return step


def time_steps(step, warm, k):
    for i in range(warm):
        step(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(k):
        step(warm + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'steps':
        n, k = int(sys.argv[2]), int(sys.argv[3])
        step = setup(n, sys.argv[4] if len(sys.argv) > 4 else 'bf16x6')
        for i in range(k):
            step(i)
        torch.cuda.synchronize()
        return
    sizes = (512, 1024, 1920, 4096)
    steps = {n: setup(n) for n in sizes}
    best = {n: 1e9 for n in sizes}
    for rnd in range(3):
        for n in sizes:
            ms = time_steps(steps[n], 5, 40 if n < 4096 else 20)
            best[n] = min(best[n], ms)
            print('round %d  N=%5d  %.3f ms/step  %.1f k rays/s' % (rnd, n, ms, n / ms), flush=True)
    r4096 = 4096 / best[4096]
    for n in sizes:
        print('N=%5d  %.3f ms/step  %.1f k rays/s  = %.3f of the 4096-ray rate' % (n, best[n], n / best[n], (n / best[n]) / r4096))
    print('predicted_strong_8 (no collective) = %.2f x' % (8 * 512 / best[512] / r4096))


if __name__ == '__main__':
    main()
