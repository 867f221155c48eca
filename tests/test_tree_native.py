"""Native (C++) quadtree vs golden vectors recorded from the reference's tree.py: leaf lists,
per-leaf ray plan, seeded pixel picks (compat RNG) and five gen->adjust rounds.  Bit-exact."""
import os

import numpy as np
import pytest
import torch

import fastnerf
from fastnerf.tree import QuadTreeManager


def _mgr(H, W, n, depth, images=None):
    imgs = images if images is not None else torch.zeros(n, H, W, 3)
    poses = torch.eye(4)[None, :3, :4].repeat(n, 1, 1)
    return QuadTreeManager(H, W, np.eye(3), imgs, poses, 0.0, depth, device='cpu')


def test_leaf_lists(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g9_tree_leaves.npz'))
    for (H, W) in ((800, 800), (378, 504), (756, 1008), (64, 64)):
        for depth in range(1, 8):
            m = _mgr(H, W, 1, depth)
            assert np.array_equal(m.leaves(0), g[f'leaves_{H}x{W}_d{depth}'])
            assert m.min_area(0) == float(g[f'minarea_{H}x{W}_d{depth}'])
            assert m.max_leaves() == 4 ** (depth - 1)


@pytest.mark.parametrize('shape', ['64x64', '100x76'])
def test_seeded_gen_adjust_sequence(golden_dir, shape):
    g = np.load(os.path.join(golden_dir, f'g9_tree_seq_{shape}.npz'))
    H, W = (int(v) for v in shape.split('x'))
    images = torch.from_numpy(g['images'])
    n = images.shape[0]
    m = _mgr(H, W, n, int(g['depth0']), images)
    for rnd in range(5):
        torch.manual_seed(100 + rnd)
        pix = m.gen_pixels(down_scale=1, last_epoch=False, compat_rng=True)
        assert np.array_equal(m.result_leaf_id.numpy(), g[f'r{rnd}_leaf_id'])
        rgb = images[pix[:, 0], pix[:, 1], pix[:, 2]]
        assert np.array_equal(rgb.numpy(), g[f'r{rnd}_rgb'])
        for ti in range(n):
            assert np.array_equal(m.leaves(ti), g[f'r{rnd}_before_t{ti}'])
        pred = torch.from_numpy(g[f'r{rnd}_pred'])
        ml = m.max_leaves()
        table = torch.zeros(n * ml)
        err = torch.abs(rgb - pred).max(dim=-1).values
        tags = m.result_leaf_id.long()
        table.scatter_reduce_(0, tags[:, 0] * ml + tags[:, 1], err, reduce='amax', include_self=True)
        m.adjust_tree_from_table(table.view(n, ml), thres=0.03)
        for ti in range(n):
            assert np.array_equal(m.leaves(ti), g[f'r{rnd}_after_t{ti}'])
            assert m.min_area(ti) == float(g[f'r{rnd}_minarea_t{ti}'])
    torch.manual_seed(999)
    pix = m.gen_pixels(down_scale=1, last_epoch=True, compat_rng=True)
    assert np.array_equal(m.result_leaf_id.numpy(), g['last_leaf_id'])
    assert np.array_equal(images[pix[:, 0], pix[:, 1], pix[:, 2]].numpy(), g['last_rgb'])


def test_export_import_roundtrip():
    m = _mgr(64, 64, 2, 3)
    t = torch.zeros(2, m.max_leaves())
    t[0, 3] = 1.0
    t[1, 0] = 1.0
    m.adjust_tree_from_table(t, thres=0.5)
    st = m.export_leaves()
    m2 = _mgr(64, 64, 2, 1)
    m2.import_leaves(st)
    for i in range(2):
        assert np.array_equal(m.leaves(i), m2.leaves(i)) and m.min_area(i) == m2.min_area(i)
    assert m.num_leaves(0) == 16 + 3


def test_vectorised_gen_respects_plan():
    m = _mgr(64, 64, 2, 3)
    t = torch.zeros(2, m.max_leaves()); t[0, 5] = 1.0
    m.adjust_tree_from_table(t, thres=0.5)
    torch.manual_seed(3)
    pix = m.gen_pixels(down_scale=1, compat_rng=False)
    tags = m.result_leaf_id.long()
    for i in range(2):
        plan = m.leaf_plan(i, 1.0)
        for li in range(plan.shape[0]):
            sel = (tags[:, 0] == i) & (tags[:, 1] == li)
            assert int(sel.sum()) == int(plan[li, 0])
            p = pix[sel]
            assert (p[:, 1] >= plan[li, 1]).all() and (p[:, 1] < plan[li, 2]).all()
            assert (p[:, 2] >= plan[li, 3]).all() and (p[:, 2] < plan[li, 4]).all()
