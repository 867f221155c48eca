#!/usr/bin/env python3
"""G20: the variance-weighted pick distribution of the nerf++-ours quadtree fork, recorded from the REFERENCE itself (build
container only; same stubbing as make_golden.py): ImageProcessor.to_prob_v2 on blocks of a variance map (image_process.py:58-72)
and -- as the end-to-end witness of `expected_pixel_counts` in oracle/tree_oracle.py -- the empirical pixel histogram of 400
seeded epochs of gen_rays_v3_multiThread(prob=True, randSamp_proc=0.25) on a depth-2 manager whose variance maps are an input
fixture (tree.py:548-607).  Data only -> tests/golden/g20_pp_prob.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import install_stubs, OUT  # noqa: E402

REF = '/root/reference/nerf++-ours'


def main():
    assert os.path.isdir(REF)
    install_stubs()
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, REF)
    import image_process as IP
    import tree as T
    rng = np.random.RandomState(3)
    out = {}
    proc = IP.ImageProcessor.__new__(IP.ImageProcessor)      # to_prob_v2 uses no state
    blocks = [np.abs(rng.randn(16, 12)) ** 2 * 0.05, np.zeros((8, 8)), rng.rand(5, 7).astype(np.float32)]
    blocks[0][:6, :5] = 0.0
    for i, b in enumerate(blocks):
        out['block%d' % i] = b
        out['prob%d' % i] = proc.to_prob_v2(b)
    # seeded epochs on the reference's manager; the variance maps replace the (unpinned, cv2-based) get_sharp_img output.  The
    # picked pixels are not returned: every ray's direction is made to CARRY its pixel (rays_d = (col, row, image)).
    H, W, n_img = 32, 32, 2

    class RS:           # the attributes QuadTreeManager reads from a RaySamplerSingleImage (tree.py:171-184)
        pass
    samplers = []
    rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    for i in range(n_img):
        rs = RS()
        rs.H, rs.W = H, W
        rs.img = rng.rand(H * W, 3).astype(np.float32)
        rs.rays_o = np.zeros((H * W, 3), dtype=np.float32)
        rs.rays_d = np.stack([cc.reshape(-1), rr.reshape(-1), np.full(H * W, i)], -1).astype(np.float32)
        samplers.append(rs)
    sharp = [np.abs(rng.randn(H, W)) ** 2 * 0.05 for _ in range(n_img)]
    sharp[0][:8, :8] = 0.0
    mgr = T.QuadTreeManager(samplers, mseThres=0.0, max_depth=2)
    mgr.processor.sharp_imgs = sharp
    hist = np.zeros((n_img, H, W), dtype=np.int64)
    rounds = 400
    torch.manual_seed(0)
    np.random.seed(0)
    n_rays = None
    for r in range(rounds):
        _, rays_d, _ = mgr.gen_rays_v3_multiThread(down_scale=1, prob=True, rand=0.25, last_epoch=False)
        d = rays_d.long().numpy()
        np.add.at(hist, (d[:, 2], d[:, 1], d[:, 0]), 1)
        n_rays = rays_d.shape[0]
    out.update(sharp0=sharp[0], sharp1=sharp[1], hist=hist, rounds=np.int64(rounds), rand=np.float64(0.25), n_rays=np.int64(n_rays),
               H=np.int64(H), W=np.int64(W))
    np.savez(os.path.join(OUT, 'g20_pp_prob.npz'), **out)
    print('wrote g20_pp_prob.npz', n_rays, 'rays per epoch')


if __name__ == '__main__':
    main()
