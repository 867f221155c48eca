#!/usr/bin/env python3
"""Differential check of the product's quadtree manager (native leaf arrays, host side: runs without a GPU) against the
REFERENCE's tree.py over random image sizes (odd and tiny ones included), depths, variance thresholds and gen -> adjust
sequences (build container only: needs /root/reference; nothing here travels or is imported by tests).  Leaf lists, minArea,
seeded pixel picks (compat RNG: the reference's own torch.randint / randperm stream), leaf tags and gathered colours must be
bit-identical.  Exit code 1 on mismatch.

Run:  python oracle/fuzz_tree_vs_reference.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from make_golden import REF, install_stubs, pose_spherical_np  # noqa: E402


def boxes(children):
    return np.array([[c.x0, c.y0, c.x1, c.y1] for c in children], dtype=np.float64)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    install_stubs()
    sys.path.insert(0, REF)
    import tree as T
    from fastnerf.tree import QuadTreeManager
    rng = np.random.RandomState(11)
    bad = 0
    for ci in range(n_cases):
        Hh = int(rng.choice([5, 8, 16, 37, 64, 100, 101, 128, 200]))
        Ww = int(rng.choice([6, 9, 16, 53, 64, 76, 77, 128, 150]))
        nimg = int(rng.randint(1, 4))
        d0 = int(rng.randint(1, 5))
        mse = float(rng.choice([0.0, 0.0, 0.02, 0.08]))
        thres = float(rng.choice([0.01, 0.03, 0.1]))
        g = torch.Generator().manual_seed(ci)
        # piecewise-smooth images so that variance-gated trees are not trivially full or trivially empty
        imgs = torch.rand(nimg, Hh, Ww, 3, generator=g) * 0.15
        imgs[:, : Hh // 2, : Ww // 3] += 0.6
        Kt = np.array([[50.0, 0, Ww / 2], [0, 50.0, Hh / 2], [0, 0, 1]])
        poses = torch.stack([pose_spherical_np(40.0 * i, -30.0, 4.0)[:3, :4] for i in range(nimg)], 0)
        errs = []
        try:
            ref = T.QuadTreeManager(Hh, Ww, Kt, imgs, poses, mseThres=mse, max_depth=d0)
        except Exception as e:     # sizes the reference itself cannot handle are not part of the contract
            print(f'case {ci:2d} {Hh}x{Ww} n={nimg} depth={d0} mse={mse}: reference raises {type(e).__name__}: skipped')
            continue
        own = QuadTreeManager(Hh, Ww, Kt, imgs, poses, mse, d0, device='cpu')

        def same_trees(tag):
            for ti in range(nimg):
                if not np.array_equal(own.leaves(ti), boxes(ref.childrens[ti])):
                    errs.append(f'{tag}:leaves[{ti}]')
                if own.min_area(ti) != float(ref.quadTrees[ti].minArea):
                    errs.append(f'{tag}:minArea[{ti}]')
        same_trees('init')
        for rnd in range(3):
            last = rnd == 2 and rng.rand() < 0.5
            try:
                torch.manual_seed(1000 + rnd)
                o, d, rgbt = ref.gen_rays_v3_multiThread(down_scale=1, prob=False, randSamp_proc=1.0, last_epoch=last)
            except Exception as e:
                print(f'case {ci:2d} round {rnd}: reference gen raises {type(e).__name__}: stop')
                break
            torch.manual_seed(1000 + rnd)
            pix = own.gen_pixels(down_scale=1, last_epoch=last, compat_rng=True)
            if not np.array_equal(own.result_leaf_id.numpy(), ref.result_leaf_id.numpy()):
                errs.append(f'r{rnd}:leaf_id')
            elif not np.array_equal(imgs[pix[:, 0], pix[:, 1], pix[:, 2]].numpy(), rgbt.numpy()):
                errs.append(f'r{rnd}:rgb')
            if last:
                same_trees(f'r{rnd}-last')
                break
            pred = torch.clamp(rgbt + (torch.rand(rgbt.shape, generator=g) - 0.5) * 0.3 *
                               (torch.rand(rgbt.shape[0], 1, generator=g) < 0.05).float(), 0, 1)
            if ci % 4 == 3:
                # the older single-thread pair gen_rays_v3_1 / adjust_tree (mean rule): v3_1 must draw what multiThread drew,
                # and the mean-rule split must match the product's sum / count path
                st = torch.get_rng_state()
                torch.manual_seed(1000 + rnd)
                _, _, rgb31 = ref.gen_rays_v3_1(down_scale=1, last_epoch=False)
                torch.set_rng_state(st)
                if not torch.equal(rgb31, rgbt):
                    errs.append(f'r{rnd}:v3_1')
                thres_m = thres * 0.1
                ref.adjust_tree(rgbt, pred, thres=thres_m)
                ml = own.max_leaves()
                tags = own.result_leaf_id.long()
                slot = tags[:, 0] * ml + tags[:, 1]
                sums = torch.zeros(nimg * ml, dtype=torch.float64).index_add_(0, slot, (rgbt - pred).abs().double().sum(-1))
                counts = torch.zeros(nimg * ml, dtype=torch.int32).index_add_(0, slot, torch.ones(slot.shape[0], dtype=torch.int32))
                own.adjust_tree_from_sumcount(sums, counts, thres_m)
                same_trees(f'r{rnd}-mean')
                continue
            ref.adjust_tree_multiThread(rgbt, pred, thres=thres)
            # (the product's adjust_tree_multiThread reduces on the GPU; here the same per-(image, leaf) max is formed on the
            # host and handed to the native split logic -- the part under test)
            ml = own.max_leaves()
            table = torch.zeros(nimg * ml)
            tags = own.result_leaf_id.long()
            table.scatter_reduce_(0, tags[:, 0] * ml + tags[:, 1], torch.abs(rgbt - pred).max(dim=-1).values, reduce='amax',
                                  include_self=True)
            own.adjust_tree_from_table(table.view(nimg, ml), thres=thres)
            same_trees(f'r{rnd}')
        bad += bool(errs)
        print(f'case {ci:2d} {Hh}x{Ww} n={nimg} depth={d0} mse={mse} thres={thres}: leaves {[own.num_leaves(i) for i in range(nimg)]} '
              + ('OK' if not errs else 'MISMATCH ' + ' '.join(errs)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
