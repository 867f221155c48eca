"""Pins oracle/tree_oracle.py against golden leaf lists / seeded gen+adjust
sequences recorded from the reference's tree.py.  Integer /
index work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import tree_oracle as TO


def test_leaf_lists(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g9_tree_leaves.npz'))
    for (H, W) in ((800, 800), (378, 504), (756, 1008), (64, 64)):
        for depth in range(1, 8):
            tr = TO.Tree(H, W, depth)
            ref = g[f'leaves_{H}x{W}_d{depth}']
            assert np.array_equal(np.array(tr.leaves, dtype=np.float64).reshape(-1, 4), ref)
            assert tr.minArea == float(g[f'minarea_{H}x{W}_d{depth}'])


@pytest.mark.parametrize('shape', ['64x64', '100x76'])
def test_seeded_gen_adjust_sequence(golden_dir, shape):
    g = np.load(os.path.join(golden_dir, f'g9_tree_seq_{shape}.npz'))
    H, W = (int(v) for v in shape.split('x'))
    images = torch.from_numpy(g['images'])
    n = images.shape[0]
    mgr = TO.Manager(H, W, n, int(g['depth0']))
    for rnd in range(5):
        torch.manual_seed(100 + rnd)
        pix = mgr.gen_pixels(down_scale=1, last_epoch=False)
        assert np.array_equal(mgr.result_leaf_id.numpy(), g[f'r{rnd}_leaf_id'])
        rgb = images[pix[:, 0], pix[:, 1], pix[:, 2]]
        assert np.array_equal(rgb.numpy(), g[f'r{rnd}_rgb'])
        for ti in range(n):
            assert np.array_equal(mgr.leaf_array(ti), g[f'r{rnd}_before_t{ti}'])
        pred = torch.from_numpy(g[f'r{rnd}_pred'])
        # table-driven variant must take the same decisions as the row-scan variant
        import copy
        m2 = copy.deepcopy(mgr)
        tab = [[-np.inf] * len(t.leaves) for t in mgr.trees]
        err = torch.abs(rgb - pred).max(dim=-1).values.numpy()
        tags = mgr.result_leaf_id.numpy().astype(np.int64)
        for e, (ti, li) in zip(err, tags):
            tab[ti][li] = max(tab[ti][li], e)
        m2.adjust_from_table(tab, 0.03)
        mgr.adjust(rgb, pred, 0.03)
        for ti in range(n):
            assert np.array_equal(mgr.leaf_array(ti), g[f'r{rnd}_after_t{ti}'])
            assert np.array_equal(m2.leaf_array(ti), g[f'r{rnd}_after_t{ti}'])
            assert mgr.trees[ti].minArea == float(g[f'r{rnd}_minarea_t{ti}'])
    torch.manual_seed(999)
    pix = mgr.gen_pixels(down_scale=1, last_epoch=True)
    assert np.array_equal(mgr.result_leaf_id.numpy(), g['last_leaf_id'])
    assert np.array_equal(images[pix[:, 0], pix[:, 1], pix[:, 2]].numpy(), g['last_rgb'])
    assert pix.shape[0] == n * H * W


def test_weighted_pick_distribution_g20(golden_dir):
    """G20 (recorded from nerf++-ours by oracle/make_golden_prob.py): the oracle's to_prob_v2 is the reference's bit for bit, and
    its per-pixel expectation of one prob=True epoch (tree.py:548-607) explains the reference's own pixel histogram over 400 seeded
    epochs: 5 sigma per pixel, and chi-square over all pixels.  This is what tests/test_gpu_epoch_rays.py takes the device
    sampler's expectation from.  (The variance maps are an input fixture: get_sharp_img rests on OpenCV, parity unpinned.)"""
    g = np.load(os.path.join(golden_dir, 'g20_pp_prob.npz'))
    for i in range(3):
        p = TO.to_prob_v2(g['block%d' % i])
        assert p.dtype == np.float64 and np.array_equal(p, g['prob%d' % i])
    H, W, rounds, rand = int(g['H']), int(g['W']), int(g['rounds']), float(g['rand'])
    mgr = TO.Manager(H, W, 2, 2)
    exp = rounds * TO.expected_pixel_counts(mgr.trees, H, W, [g['sharp0'], g['sharp1']], 1.0, True, rand)
    hist = g['hist'].astype(np.float64)
    assert abs(exp.sum() - hist.sum()) < 1e-6 * hist.sum() and hist.sum() == rounds * int(g['n_rays'])
    sig = np.sqrt(np.maximum(exp, 1.0))
    assert (np.abs(hist - exp) <= 5 * sig + 2).all()
    chi2 = float((((hist - exp) ** 2) / np.maximum(exp, 1e-9)).sum())
    dof = hist.size - 1
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), (chi2, dof)
    # the flat block of image 0 gets the clipped floor, not zero
    assert exp[0, :8, :8].min() > 0.25 * exp[0].mean() and exp[0, :8, :8].max() < 0.3 * exp[0].mean()   # uniform share + floor
    assert TO.leaf_pick_split(10, 0.25) == (7, 3) and TO.leaf_pick_split(256, 0.7) == (76, 180)
