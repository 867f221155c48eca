#!/usr/bin/env python3
"""Differential check of the nerf++ quadtree fork on the product's native manager against the REFERENCE's
nerf++-ours/tree.py over random image sizes, depths, weighted-pick fractions (`prob=True`, `rand`) and MEAN-rule adjust
thresholds (build container only: needs /root/reference; nothing here travels or is imported by tests).  The variance map is
taken from the reference's ImageProcessor (computed with the blur stand-in: OpenCV is absent) and handed to the
product as an input.  Seeded picks (torch + numpy streams), leaf tags, colours, directions and the leaf lists after every adjust
must be bit-identical.  Exit code 1 on mismatch.

Run:  python oracle/fuzz_pp_tree_vs_reference.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from make_golden import install_stubs  # noqa: E402
from make_golden_pp import REF  # noqa: E402


class RS:
    pass


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    install_stubs()
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, REF)
    import tree as TP
    import fastnerf
    rng = np.random.RandomState(23)
    bad = 0
    for ci in range(n_cases):
        Ht = int(rng.choice([16, 24, 37, 48, 64, 100]))
        Wt = int(rng.choice([16, 20, 40, 53, 64, 76]))
        nimg = int(rng.randint(1, 4))
        d0 = int(rng.randint(1, 4))
        rand = float(rng.choice([0.3, 0.5, 0.7, 1.0]))
        thres = float(rng.choice([0.004, 0.012, 0.03]))
        prob = bool(rng.rand() < 0.8)
        samplers = []
        for i in range(nimg):
            rs = RS()
            rs.H, rs.W = Ht, Wt
            rr, cc = np.meshgrid(np.arange(Ht), np.arange(Wt), indexing='ij')
            tex = 0.5 + 0.5 * np.sin((0.2 + 0.3 * rng.rand()) * rr + 0.11 * i) * np.cos((0.1 + 0.3 * rng.rand()) * cc)
            rs.img = np.stack([tex, (rr * Wt + cc) / float(Ht * Wt), rng.rand(Ht, Wt)], -1).astype(np.float32).reshape(-1, 3)
            rs.rays_o = np.tile(np.array([[0.1 * i, 0.0, 0.2]], dtype=np.float32), (Ht * Wt, 1))
            rs.rays_d = np.stack([cc.reshape(-1) / Wt - 0.5, rr.reshape(-1) / Ht - 0.5, -np.ones(Ht * Wt)], -1).astype(np.float32)
            samplers.append(rs)
        so = sys.stdout
        sys.stdout = open(os.devnull, 'w')
        try:
            ref = TP.QuadTreeManager(samplers, mseThres=0.0, max_depth=d0)
        finally:
            sys.stdout = so
        own = fastnerf.nerfpp.QuadTreeManager(samplers, mseThres=0.0, max_depth=d0, device='cpu',
                                              sharp_imgs=[np.asarray(s) for s in ref.processor.sharp_imgs])
        g = torch.Generator().manual_seed(ci)
        errs = []
        for rnd in range(3):
            try:
                torch.manual_seed(200 + rnd); np.random.seed(300 + rnd)
                sys.stdout = open(os.devnull, 'w')
                o, d, rgbt = ref.gen_rays_v3_multiThread(down_scale=1, prob=prob, rand=rand, last_epoch=False)
            except Exception as e:
                sys.stdout = so
                print(f'case {ci:2d} round {rnd}: reference gen raises {type(e).__name__}: stop')
                break
            finally:
                sys.stdout = so
            torch.manual_seed(200 + rnd); np.random.seed(300 + rnd)
            o2, d2, rgb2 = own.gen_rays_v3_multiThread(down_scale=1, prob=prob, rand=rand, last_epoch=False)
            if not np.array_equal(own.result_leaf_id.numpy(), ref.result_leaf_id.numpy()):
                errs.append(f'r{rnd}:leaf_id')
            elif not (np.array_equal(rgb2.numpy(), rgbt.numpy()) and np.array_equal(d2.numpy(), d.numpy())):
                errs.append(f'r{rnd}:rgb/d')
            pred = torch.clamp(rgbt + (torch.rand(rgbt.shape, generator=g) - 0.5) * 0.2 *
                               (torch.rand(rgbt.shape[0], 1, generator=g) < 0.5).float(), 0, 1)
            sys.stdout = open(os.devnull, 'w')
            try:
                ref.adjust_tree_multiThread(rgbt, pred, thres=thres)
            finally:
                sys.stdout = so
            ml = own.max_leaves()
            tags = own.result_leaf_id.long()
            slot = tags[:, 0] * ml + tags[:, 1]
            sums = torch.zeros(nimg * ml, dtype=torch.float64).index_add_(0, slot, (rgb2 - pred).abs().double().sum(-1))
            counts = torch.zeros(nimg * ml, dtype=torch.int32).index_add_(0, slot, torch.ones_like(slot, dtype=torch.int32))
            own.adjust_tree_from_sumcount(sums, counts, thres)
            for ti in range(nimg):
                want = np.array([[c_.x0, c_.y0, c_.x1, c_.y1] for c_ in ref.childrens[ti]], dtype=np.float64)
                if not np.array_equal(own.leaves(ti), want):
                    errs.append(f'r{rnd}:leaves[{ti}]')
        bad += bool(errs)
        print(f'case {ci:2d} {Ht}x{Wt} n={nimg} depth={d0} prob={int(prob)} rand={rand} thres={thres}: leaves '
              f'{[own.num_leaves(i) for i in range(nimg)]} ' + ('OK' if not errs else 'MISMATCH ' + ' '.join(errs)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
