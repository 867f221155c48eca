"""Pins oracle/nerfpp_oracle.py against golden vectors recorded from nerf++-ours (G10)."""
import os

import numpy as np
import torch

from oracle import nerfpp_oracle as PP

torch.set_num_threads(4)


def T(a):
    return torch.from_numpy(np.asarray(a))


def load_levels(golden_dir):
    w = np.load(os.path.join(golden_dir, 'g10_pp_weights.npz'))
    levels = []
    for m in range(2):
        fg = {k[len(f'l{m}.fg_net.'):]: T(w[k]).clone() for k in w.files if k.startswith(f'l{m}.fg_net.')}
        bg = {k[len(f'l{m}.bg_net.'):]: T(w[k]).clone() for k in w.files if k.startswith(f'l{m}.bg_net.')}
        levels.append((fg, bg))
    return levels


def test_ops(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_pp_ops.npz'))
    ro, rd, dep = PP.get_rays_single_image(6, 8, g['intr'], g['c2w'])
    assert np.array_equal(ro, g['ro_s']) and np.array_equal(rd, g['rd_s']) and np.array_equal(dep, g['dep_s'])
    o, d = T(g['ray_o']), T(g['ray_d'])
    assert np.array_equal(PP.intersect_sphere(o, d).numpy(), g['fg_far'])
    pts, dr = PP.depth2pts_outside(o[:, None].expand(40, 16, 3), d[:, None].expand(40, 16, 3), T(g['depth']))
    assert np.array_equal(pts.numpy(), g['pts']) and np.array_equal(dr.numpy(), g['depth_real'])
    assert np.array_equal(PP.sample_pdf(T(g['bins']), T(g['w']), 128, None).numpy(), g['s_det'])
    assert np.array_equal(PP.sample_pdf(T(g['bins']), T(g['w']), 128, T(g['u'])).numpy(), g['s_u'])
    try:
        PP.intersect_sphere(torch.tensor([[2.0, 0, 0]]), torch.tensor([[0.0, 1.0, 0]]))
        assert False, 'camera outside the unit sphere must raise (ddp_train_nerf.py:65-66)'
    except Exception as e:
        assert 'unit sphere' in str(e)


def test_param_layout(golden_dir):
    levels = load_levels(golden_dir)
    fg, bg = levels[0]
    assert list(fg.keys()) == [n for n, _ in PP.mlpnet_param_shapes(63)]
    assert list(bg.keys()) == [n for n, _ in PP.mlpnet_param_shapes(84)]
    for sd, ic in ((fg, 63), (bg, 84)):
        for n, shp in PP.mlpnet_param_shapes(ic):
            assert tuple(sd[n].shape) == shp
    assert sum(v.numel() for v in fg.values()) + sum(v.numel() for v in bg.values()) == 1202440


def test_forward_and_cascade_step(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_pp_step.npz'))
    levels = load_levels(golden_dir)
    ro, rd, tgt = T(g['ro']), T(g['rd']), T(g['target'])
    with torch.no_grad():
        ret = PP.nerfnet_forward(levels[0][0], levels[0][1], ro, rd, T(g['fg_far']), T(g['l0.fg_z']), T(g['l0.bg_z']))
    for k in ('rgb', 'fg_weights', 'bg_weights', 'fg_rgb', 'bg_rgb', 'bg_lambda', 'fg_depth', 'bg_depth'):
        assert np.allclose(ret[k].numpy(), g['l0.' + k], rtol=1e-5, atol=2e-6), k
    rand = [{'fg_t': T(g['fg_t']), 'bg_t': T(g['bg_t'])}, {'fg_u': T(g['fg_u']), 'bg_u': T(g['bg_u'])}]
    outs = PP.cascade_step(levels, ro, rd, tgt, [64, 128], rand)
    assert np.allclose(outs[1][2]['rgb'].numpy(), g['rgb_pred'], atol=2e-6)
    for m in range(2):
        names = ['fg_net.' + n for n, _ in PP.mlpnet_param_shapes(63)] + ['bg_net.' + n for n, _ in PP.mlpnet_param_shapes(84)]
        for n, gr in zip(names, outs[m][1]):
            ref = g[f'grad.l{m}.{n}']
            got = gr.numpy() if gr.numel() <= 40000 else gr.numpy()[:16]
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-6 * max(1.0, np.abs(ref).max()) + 1e-9), (m, n)
