#!/usr/bin/env python3
"""G16: the quadtree checkpoint files of run_nerf.py:338-345 / 542-544, recorded from the REFERENCE
(build container only; same stub set as make_golden.py).

Writes
  tests/golden/g16_treeDivide_ref.pkl   pickle.dump(treeManager.quadTrees, f) exactly as the reference does it, after two
                                        gen -> adjust rounds on two 32x24 images (non-uniform trees)
  tests/golden/g16_treepkl.npz          the inputs of that run, the leaf lists / minArea the file must decode to, and the
                                        leaf lists of the reference's variance-gated constructor QuadTree(image, thres>0, d)
and checks the opposite direction here: a file written by fastnerf.tree.save_quadtrees is loaded by the reference's
own `pickle.load` + `get_children` in a fresh interpreter that has never imported this package.

Run:  python oracle/make_golden_treepkl.py            (needs /root/reference)
"""
import os
import pickle
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import OUT, REF, install_stubs, pose_spherical_np  # noqa: E402


def leaf_array(T, tree):
    return np.array([[c.x0, c.y0, c.x1, c.y1] for c in T.get_children(tree.root)], dtype=np.float64)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import tree as T
    g = torch.Generator().manual_seed(16)
    H, W, n = 32, 24, 2
    K = np.array([[30.0, 0, W / 2], [0, 30.0, H / 2], [0, 0, 1]])
    imgs = torch.rand(n, H, W, 3, generator=g)
    poses = torch.stack([pose_spherical_np(50.0 * i, -30.0, 4.0)[:3, :4] for i in range(n)], 0)
    mgr = T.QuadTreeManager(H, W, K, imgs, poses, mseThres=0.0, max_depth=2)
    rec = {'images': imgs.numpy(), 'poses': poses.numpy(), 'K': K}
    for rnd in range(2):
        torch.manual_seed(500 + rnd)
        o, d, rgbt = mgr.gen_rays_v3_multiThread(down_scale=1, prob=False, randSamp_proc=1.0, last_epoch=False)
        pred = torch.clamp(rgbt + (torch.rand(rgbt.shape, generator=g) - 0.5) * 0.2 *
                           (torch.rand(rgbt.shape[0], 1, generator=g) < 0.01).float(), 0, 1)
        mgr.adjust_tree_multiThread(rgbt, pred, thres=0.05)
    with open(os.path.join(OUT, 'g16_treeDivide_ref.pkl'), 'wb') as f:
        pickle.dump(mgr.quadTrees, f)            # run_nerf.py:543-544
    for i in range(n):
        rec[f'leaves_t{i}'] = leaf_array(T, mgr.quadTrees[i])
        rec[f'minarea_t{i}'] = np.float64(mgr.quadTrees[i].minArea)
    rec['cur_level'] = mgr.cur_level

    # variance-gated constructor (tree.py:86-99, 655-676 with get_error): a picture with flat and busy quadrants
    yy, xx = np.meshgrid(np.arange(48), np.arange(64))
    pic = np.zeros((64, 48, 3), dtype=np.float32)
    pic[:32, :24] = 0.25
    pic[32:, 24:] = (np.sin(xx[32:, 24:, None] * 0.9) * np.cos(yy[32:, 24:, None] * 0.7) * 0.5 + 0.5)
    pic[:32, 24:, 1] = (xx[:32, 24:] % 8 < 4) * 0.8
    rec['gate_image'] = pic
    for k, (thres, depth) in enumerate(((0.01, 4), (0.05, 5), (0.2, 3))):
        qt = T.QuadTree(pic, thres, depth)
        rec[f'gate{k}_leaves'] = leaf_array(T, qt)
        rec[f'gate{k}_minarea'] = np.float64(qt.minArea)
        rec[f'gate{k}_cfg'] = np.array([thres, depth])
    np.savez_compressed(os.path.join(OUT, 'g16_treepkl.npz'), **rec)

    # ---- opposite direction: our file -> the reference's loader, in a clean interpreter ----
    sys.path.insert(0, os.path.join(HERE, '..'))
    import fastnerf
    ours = [fastnerf.tree.QuadTree.from_leaves(H, W, rec[f'leaves_t{i}'], float(rec[f'minarea_t{i}'])) for i in range(n)]
    tmp = os.path.join(tempfile.mkdtemp(), 'treeDivide_0002.pkl')
    fastnerf.tree.save_quadtrees(ours, tmp)
    code = (
        "import sys, pickle, numpy as np\n"
        "sys.path.insert(0, %r); from make_golden import install_stubs; install_stubs()\n"
        "sys.path.insert(0, %r); import tree as T\n"
        "qt = pickle.load(open(%r, 'rb'))\n"
        "assert type(qt[0]) is T.QuadTree and type(qt[0].root) is T.QuadTreeNode\n"
        "g = np.load(%r)\n"
        "for i, t in enumerate(qt):\n"
        "    a = np.array([[c.x0, c.y0, c.x1, c.y1] for c in T.get_children(t.root)])\n"
        "    assert np.array_equal(a, g['leaves_t%%d' %% i]) and t.minArea == float(g['minarea_t%%d' %% i])\n"
        "print('reference loaded our treeDivide pkl: ok')\n" % (HERE, REF, tmp, os.path.join(OUT, 'g16_treepkl.npz')))
    subprocess.run([sys.executable, '-c', code], check=True)
    print('wrote g16 fixtures to', OUT)


if __name__ == '__main__':
    main()
