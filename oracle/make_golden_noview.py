#!/usr/bin/env python3
"""G19: the model WITHOUT view directions (`use_viewdirs=False`: nerf-ours/model.py:35-36,60-61, run_nerf.py:73-77,
render.py:59-78,218) recorded from the REFERENCE (build container only).

create_nerf(use_viewdirs=False, N_importance=32) -> two nets with `output_linear` 256 -> 5.  Their trunks take the weights
of the G7 nets (tests/golden/g7_weights.npz, `pts_linears.*`), so this fixture only adds the small tensors: the two
`output_linear` layers as initialised under torch.manual_seed(0); `views_linears.0` (present but never used by this model)
is zeroed.  Recorded: a train-mode render of 64 rays (32 + 32 samples, pytest draws, white background) with retraw (five raw
channels), the two-term loss, and the gradients of every bias, both `output_linear.weight` and `pts_linears.{0,7}.weight`.
Data-only -> tests/golden/g19_noview.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF, install_stubs, pose_spherical_np  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import render as R
    import run_nerf as RN
    import run_nerf_helpers as H

    class A:
        pass
    args = A()
    args.multires, args.multires_views, args.i_embed = 10, 4, 0
    args.use_viewdirs, args.N_importance, args.netdepth, args.netwidth = False, 32, 8, 256
    args.netdepth_fine, args.netwidth_fine, args.netchunk = 8, 256, 65536
    args.lrate, args.basedir, args.expname, args.ft_path, args.no_reload = 5e-4, '/tmp', 'golden_tmp', None, True
    args.perturb, args.N_samples, args.white_bkgd, args.raw_noise_std = 1.0, 32, True, 0.0
    args.dataset_type, args.no_ndc, args.lindisp = 'blender', False, False
    os.makedirs('/tmp/golden_tmp', exist_ok=True)
    torch.manual_seed(0)
    kw_train, kw_test, _, _, grad_vars, optim = RN.create_nerf(args)
    assert kw_train['use_viewdirs'] is False
    g7 = np.load(os.path.join(OUT, 'g7_weights.npz'))
    rec = {}
    for pre, key in (('c.', 'network_fn'), ('f.', 'network_fine')):
        net = kw_train[key]
        sd = net.state_dict()
        names = [k.replace('module.', '') for k in sd.keys()]
        assert names[-2:] == ['output_linear.weight', 'output_linear.bias'] and 'views_linears.0.weight' in names
        new = {}
        for k, v in sd.items():
            n = k.replace('module.', '')
            if n.startswith('pts_linears.'):
                new[k] = torch.from_numpy(g7[pre + n])
            elif n.startswith('views_linears.'):
                new[k] = torch.zeros_like(v)
            else:
                new[k] = v
                rec['w.' + pre + n] = v.detach().numpy().copy()
        net.load_state_dict(new)
        rec['names.' + pre[0]] = np.array(names)
        rec['shape.' + pre + 'views_linears.0.weight'] = np.array(sd[[k for k in sd if 'views_linears.0.weight' in k][0]].shape)

    g = torch.Generator().manual_seed(4321)
    c2w = pose_spherical_np(30.0, -30.0, 4.0)[:3, :4]
    focal = 0.5 * 800 / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 400.0], [0, focal, 400.0], [0, 0, 1]])
    o_b, d_b = H.get_rays(800, 800, K, c2w)
    sel = torch.randint(0, 800 * 800, (64,), generator=g)
    ro = o_b.reshape(-1, 3)[sel].contiguous()
    rd = d_b.reshape(-1, 3)[sel].contiguous()
    tgt = torch.rand(64, 3, generator=g)
    rgb, disp, acc, ex = R.render(800, 800, K, chunk=32768, rays=torch.stack([ro, rd], 0), retraw=True, near=2.0, far=6.0,
                                  pytest=True, **kw_train)
    optim.zero_grad()
    l1 = H.img2mse(rgb, tgt)
    l0 = H.img2mse(ex['rgb0'], tgt)
    (l1 + l0).backward()
    for pre, key in (('c.', 'network_fn'), ('f.', 'network_fine')):
        for n, p in kw_train[key].named_parameters():
            n = n.replace('module.', '')
            if n.startswith('views_linears.'):
                assert p.grad is None          # never touched by the forward pass
            elif n.endswith('bias') or n in ('output_linear.weight', 'pts_linears.0.weight', 'pts_linears.7.weight'):
                rec['grad.' + pre + n] = p.grad.detach().numpy().copy()
    np.random.seed(0)
    t_rand = np.random.rand(64, 32).astype(np.float32)
    np.random.seed(0)
    u = np.random.rand(64, 32).astype(np.float32)
    # test-mode render (perturb 0, deterministic inverse-CDF) of the same rays
    rgb_t, disp_t, acc_t, ex_t = R.render(800, 800, K, chunk=32768, rays=torch.stack([ro, rd], 0), near=2.0, far=6.0, **kw_test)
    np.savez_compressed(os.path.join(OUT, 'g19_noview.npz'), ro=ro.numpy(), rd=rd.numpy(), target=tgt.numpy(), t_rand=t_rand, u=u,
                        K=K, rgb=rgb.detach().numpy(), disp=disp.detach().numpy(), acc=acc.detach().numpy(),
                        raw=ex['raw'].detach().numpy(), rgb0=ex['rgb0'].detach().numpy(), disp0=ex['disp0'].detach().numpy(),
                        acc0=ex['acc0'].detach().numpy(), z_std=ex['z_std'].detach().numpy(), loss=float(l1), loss0=float(l0),
                        test_rgb=rgb_t.detach().numpy(), test_disp=disp_t.detach().numpy(), test_acc=acc_t.detach().numpy(),
                        **rec)
    print('wrote g19_noview.npz; raw', tuple(ex['raw'].shape), 'loss', float(l1), float(l0))


if __name__ == '__main__':
    main()
