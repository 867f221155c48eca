"""INTEGRATION.md option A end to end: tests/ref_loop_driver.py uses the package the way the reference's driver uses its own
modules (create_nerf's 6-tuple, render(retraw=True), img2mse, loss.backward(), torch.optim.Adam, the decayed learning rate,
`from ... import QuadTreeManager, get_children`, .tar / treeDivide pkl files) for 4 epochs on the synthetic set with a save ->
FRESH-PROCESS resume after epoch 2; the files written by that loop load in the fused train() path and in a reference-shaped
reader, and vice versa."""
import json
import os
import pickle
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(basedir, stop_after=0, n_epoch=4):
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'ref_loop_driver.py'), '--basedir', basedir, '--n_epoch', str(n_epoch)]
    if stop_after:
        cmd += ['--stop_after', str(stop_after)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('LOOPLOG ')][-1]
    return json.loads(line[len('LOOPLOG '):])


def _reference_side_unpickle(path):
    """What the reference's `pickle.load(f)` does: classes resolved in a module named `tree`."""
    stub = types.ModuleType('tree')
    stub.QuadTree = type('QuadTree', (), {'__module__': 'tree'})
    stub.QuadTreeNode = type('QuadTreeNode', (), {'__module__': 'tree'})
    prev = sys.modules.get('tree')
    sys.modules['tree'] = stub
    try:
        with open(path, 'rb') as f:
            return pickle.load(f)
    finally:
        if prev is None:
            del sys.modules['tree']
        else:
            sys.modules['tree'] = prev


def test_reference_shaped_loop_with_fresh_process_resume(tmp_path):
    import fastnerf
    base = str(tmp_path)
    a = _run(base, stop_after=2)
    d = os.path.join(base, 'loop')
    assert a['resumed_from'] == 0 and not a['loaded_tree'] and [e['epoch'] for e in a['epochs']] == [1, 2]
    assert sorted(os.listdir(d)) == ['001.tar', '002.tar', 'treeDivide_0001.pkl', 'treeDivide_0002.pkl']
    assert a['epochs'][0]['leaves_after'] > a['epochs'][0]['leaves_before'] == 16     # the trees were refined
    assert a['epochs'][1]['cur_level'] == 4
    # ---- fresh process: resumes from 002.tar + treeDivide_0002.pkl -------------------------------------------
    b = _run(base)
    assert b['resumed_from'] == 2 and b['loaded_tree'] and [e['epoch'] for e in b['epochs']] == [3, 4]
    assert b['leaves_at_start'] == a['leaves_at_end']                                  # subdivision survived the restart
    assert b['iter_at_start'] == a['global_iter'] and b['global_iter'] > a['global_iter']
    assert b['adam_step'] == a['adam_step'] + sum(e['iters'] for e in b['epochs'])      # optimizer state continued
    assert 'warmup_loss' not in b                                                       # no second warm-up (run_nerf.py:367)
    assert b['epochs'][0]['leaves_before'] == a['epochs'][1]['leaves_after']
    assert b['epochs'][0]['leaves_after'] == b['epochs'][0]['leaves_before']            # epoch 3 = n_epoch-1: no subdivide
    assert b['epochs'][1]['rays'] == 4 * 32 * 32                                        # last epoch: every pixel (tree.py:390-400)
    assert all(np.isfinite(e['loss_last']) for e in a['epochs'] + b['epochs'])
    assert b['epochs'][-1]['loss_last'] < a['epochs'][0]['loss_first']                  # it trains across the restart
    lr_want = 5e-4 * 0.1 ** ((b['global_iter'] - 1) / 500000.0)
    assert abs(b['epochs'][-1]['lr'] - lr_want) < 1e-12                                 # pre-increment LR rule
    # ---- the files, read the way the reference reads them ----------------------------------------------------
    ck = torch.load(os.path.join(d, '004.tar'), map_location='cpu', weights_only=False)
    assert set(ck) == {'global_epoch', 'global_iter', 'network_fn_state_dict', 'network_fine_state_dict', 'optimizer_state_dict'}
    assert all(k.startswith('module.') for k in ck['network_fn_state_dict']) and len(ck['network_fn_state_dict']) == 24
    assert len(ck['optimizer_state_dict']['state']) == 48
    qt = _reference_side_unpickle(os.path.join(d, 'treeDivide_0004.pkl'))
    leaves = [np.array([[c.x0, c.y0, c.x1, c.y1] for c in fastnerf.tree.get_children(t.root)]) for t in qt]
    assert [l.tolist() for l in leaves] == b['leaves_at_end']
    # ---- ... and by the fused path: train() resumes from what the reference-shaped loop wrote ------------------
    imgs, poses, focal = fastnerf.synthetic.make_dataset(n_images=4, H=32, W=32)
    args = fastnerf.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, N_rand=256, n_epoch=6,
                                       init_level=2, subdivide_every=1, subdivide_thres=0.05, lrate=5e-4, lrate_decay=500,
                                       basedir=base, expname='loop', no_reload=False, save_ckpt=True)
    logs = []
    ktr, kte, trainer, mgr, hist = fastnerf.run_nerf.train(imgs, poses, 32, 32, focal, args, log=logs.append)
    assert [h[0] for h in hist] == [5, 6] and any('treeDivide_0004.pkl' in l for l in logs)
    assert trainer.adam_t == b['adam_step'] + sum(h[1] for h in hist)
    assert os.path.exists(os.path.join(d, '006.tar')) and os.path.exists(os.path.join(d, 'treeDivide_0006.pkl'))
    # ---- and back: a fresh reference-shaped process continues from the fused path's epoch-6 files ---------------
    c = _run(base, n_epoch=7)
    assert c['resumed_from'] == 6 and c['loaded_tree'] and c['leaves_at_start'] == [mgr.leaves(i).tolist() for i in range(4)]
    assert c['adam_step'] == trainer.adam_t + c['epochs'][0]['iters']
