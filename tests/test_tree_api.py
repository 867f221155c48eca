"""The object surface of nerf-ours/tree.py that run_nerf.py touches besides the manager's methods: QuadTreeNode /
QuadTree / get_children / recursive_subdivide, the settable `quadTrees` / `childrens` views, and the
treeDivide_*.pkl files (run_nerf.py:338-345, 542-544) in both directions.  Goldens: G16 (oracle/make_golden_treepkl.py,
recorded from the reference) and the G9 adjust sequences."""
import os
import pickle
import pickletools
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

import fastnerf
from fastnerf.tree import (QuadTree, QuadTreeManager, QuadTreeNode, get_children, load_quadtrees, recursive_subdivide,
                           save_quadtrees)


def _mgr(H, W, n, depth, images=None, thres=0.0):
    imgs = images if images is not None else torch.zeros(n, H, W, 3)
    poses = torch.eye(4)[None, :3, :4].repeat(n, 1, 1)
    return QuadTreeManager(H, W, np.eye(3), imgs, poses, thres, depth, device='cpu')


def _boxes(nodes):
    return np.array([[c.x0, c.y0, c.x1, c.y1] for c in nodes], dtype=np.float64).reshape(-1, 4)


def test_reference_pickle_loads(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g16_treepkl.npz'))
    trees = load_quadtrees(os.path.join(golden_dir, 'g16_treeDivide_ref.pkl'))     # written by the reference itself
    assert len(trees) == 2 and all(type(t) is QuadTree and type(t.root) is QuadTreeNode for t in trees)
    for i, t in enumerate(trees):
        assert np.array_equal(_boxes(get_children(t.root)), g[f'leaves_t{i}'])
        assert t.minArea == float(g[f'minarea_t{i}']) and (t.H, t.W) == (32, 24)
    # the statements of run_nerf.py:339-345 with the product manager
    treeManager = _mgr(32, 24, 2, 2, torch.from_numpy(g['images']))
    global_epoch = 2
    with open(os.path.join(golden_dir, 'g16_treeDivide_ref.pkl'), 'rb') as f:
        treeManager.load_trees(f.name)
    treeManager.quadTrees = trees
    treeManager.childrens = [get_children(treeManager.quadTrees[i].root) for i in range(treeManager.n_images)]
    treeManager.cur_level = global_epoch
    for i in range(2):
        assert np.array_equal(treeManager.leaves(i), g[f'leaves_t{i}'])
        assert treeManager.min_area(i) == float(g[f'minarea_t{i}'])
        assert np.array_equal(_boxes(treeManager.childrens[i]), g[f'leaves_t{i}'])
        assert treeManager.quadTrees[i].minArea == float(g[f'minarea_t{i}'])


def test_written_pickle_names_reference_classes(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, 'g16_treepkl.npz'))
    m = _mgr(32, 24, 2, 2)
    m.quadTrees = [QuadTree.from_leaves(32, 24, g[f'leaves_t{i}'], float(g[f'minarea_t{i}'])) for i in range(2)]
    p = str(tmp_path / 'treeDivide_0002.pkl')
    had = 'tree' in sys.modules
    m.save_trees(p)
    assert ('tree' in sys.modules) == had                       # the temporary alias is gone again
    globs = {arg for op, arg, _ in pickletools.genops(open(p, 'rb').read()) if op.name == 'GLOBAL'}
    assert globs == {'tree QuadTree', 'tree QuadTreeNode'}      # nothing of this package is named in the file
    # what the reference's `pickle.load` does with it: classes looked up in a module called `tree`, __dict__ restored
    stub = types.ModuleType('tree')
    stub.QuadTree = type('QuadTree', (), {'__module__': 'tree'})
    stub.QuadTreeNode = type('QuadTreeNode', (), {'__module__': 'tree'})
    prev = sys.modules.get('tree')
    sys.modules['tree'] = stub
    try:
        with open(p, 'rb') as f:
            qt = pickle.load(f)
    finally:
        if prev is None:
            del sys.modules['tree']
        else:
            sys.modules['tree'] = prev
    for i, t in enumerate(qt):
        assert type(t) is stub.QuadTree and t.minArea == float(g[f'minarea_t{i}']) and (t.H, t.W) == (32, 24)
        assert np.array_equal(_boxes(get_children(t.root)), g[f'leaves_t{i}'])
    # and our own reader
    back = load_quadtrees(p)
    assert all(np.array_equal(back[i].leaf_array(), g[f'leaves_t{i}']) for i in range(2))


@pytest.mark.skipif(not os.path.isdir('/root/reference/nerf-ours'), reason='needs the reference checkout (build container)')
def test_reference_loads_written_pickle(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, 'g16_treepkl.npz'))
    p = str(tmp_path / 'treeDivide_0002.pkl')
    save_quadtrees([QuadTree.from_leaves(32, 24, g[f'leaves_t{i}'], float(g[f'minarea_t{i}'])) for i in range(2)], p)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, pickle, numpy as np\n"
            f"sys.path.insert(0, {os.path.join(root, 'oracle')!r}); from make_golden import install_stubs; install_stubs()\n"
            "sys.path.insert(0, '/root/reference/nerf-ours'); import tree as T\n"
            f"qt = pickle.load(open({p!r}, 'rb')); g = np.load({os.path.join(golden_dir, 'g16_treepkl.npz')!r})\n"
            "assert type(qt[0]) is T.QuadTree\n"
            "for i, t in enumerate(qt):\n"
            "    a = np.array([[c.x0, c.y0, c.x1, c.y1] for c in T.get_children(t.root)])\n"
            "    assert np.array_equal(a, g['leaves_t%d' % i]) and t.minArea == float(g['minarea_t%d' % i])\n"
            "    assert t.root.children[0].area > 0 and str(t.root)\n")
    subprocess.run([sys.executable, '-c', code], check=True)


def test_variance_gated_constructor(golden_dir):
    """QuadTree(image, thres > 0, depth) and QuadTreeManager(mseThres > 0) == the reference's get_error-gated trees."""
    g = np.load(os.path.join(golden_dir, 'g16_treepkl.npz'))
    pic = g['gate_image']
    for k in range(3):
        thres, depth = float(g[f'gate{k}_cfg'][0]), int(g[f'gate{k}_cfg'][1])
        qt = QuadTree(pic, thres, depth)
        assert np.array_equal(qt.leaf_array(), g[f'gate{k}_leaves']) and qt.minArea == float(g[f'gate{k}_minarea'])
        m = _mgr(64, 48, 2, depth, torch.from_numpy(np.stack([pic, pic], 0)), thres=thres)
        assert np.array_equal(m.leaves(1), g[f'gate{k}_leaves']) and m.min_area(1) == float(g[f'gate{k}_minarea'])
        # torch images go through the same statistics
        qt2 = QuadTree(torch.from_numpy(pic), thres, depth)
        assert np.array_equal(qt2.leaf_array(), g[f'gate{k}_leaves'])
    node = QuadTreeNode(0, 0, 64, 48)
    recursive_subdivide(node, 0.0, pic, 1, 3)
    assert len(get_children(node)) == 16 and node.children[3].children[0].box() == (32.0, 24.0, 48.0, 36.0)
    assert str(node) == '(0.0, 0.0), (64.0, 48.0)' and node.area == 64 * 48


@pytest.mark.parametrize('shape', ['64x64', '100x76'])
def test_views_follow_the_native_trees(golden_dir, shape):
    """Through five seeded gen -> adjust rounds (G9) the object views enumerate exactly the reference's leaves, and
    re-assigning them is the identity."""
    g = np.load(os.path.join(golden_dir, f'g9_tree_seq_{shape}.npz'))
    H, W = (int(v) for v in shape.split('x'))
    images = torch.from_numpy(g['images'])
    n = images.shape[0]
    m = _mgr(H, W, n, int(g['depth0']), images)
    for rnd in range(5):
        torch.manual_seed(100 + rnd)
        m.gen_pixels(down_scale=1, last_epoch=False, compat_rng=True)
        m.result_leaf_tag = m._tags_i32
        # CPU stand-in of the device table: per-(image, leaf) max |gt - pred|
        rgb = images[m.result_pix[:, 0], m.result_pix[:, 1], m.result_pix[:, 2]]
        pred = torch.from_numpy(g[f'r{rnd}_pred'])
        err = (rgb - pred).abs().max(1).values
        ml = m.max_leaves()
        table = torch.zeros(n, ml)
        flat = m._tags_i32[:, 0].long() * ml + m._tags_i32[:, 1].long()
        table.view(-1).scatter_reduce_(0, flat, err, reduce='amax')
        m.adjust_tree_from_table(table, thres=0.03)
        for ti in range(n):
            want = g[f'r{rnd}_after_t{ti}']
            assert np.array_equal(_boxes(m.childrens[ti]), want)
            assert np.array_equal(_boxes(get_children(m.quadTrees[ti].root)), want)
            assert m.quadTrees[ti].minArea == float(g[f'r{rnd}_minarea_t{ti}'])
            assert all(c.children == [] for c in m.childrens[ti])
        m.quadTrees = m.quadTrees
        m.childrens = m.childrens
        for ti in range(n):
            assert np.array_equal(m.leaves(ti), g[f'r{rnd}_after_t{ti}'])


def test_from_leaves_rejects_garbage():
    with pytest.raises(ValueError):
        QuadTree.from_leaves(8, 8, np.array([[0, 0, 4, 4], [4, 0, 8, 4]], dtype=np.float64), 16.0)
    with pytest.raises(ValueError):
        QuadTree.from_leaves(8, 8, np.array([[0, 0, 3, 3]], dtype=np.float64), 16.0)
