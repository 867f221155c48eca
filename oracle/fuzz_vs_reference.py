#!/usr/bin/env python3
"""Differential check of the oracle against the REFERENCE itself over random render configurations (build container
only: needs /root/reference; nothing here travels or is imported by tests).  For every draw of (N_samples, N_importance,
lindisp, white_bkgd, perturb, raw_noise_std, use_viewdirs, ndc, shared fine net) both sides render the same rays with the
same injected randoms (the reference's `pytest=True` hook = np.random.seed(0) draws) and the same weights; outputs and the
gradients of the two-term loss must agree to fp32 round-off.  Prints one line per configuration; exit code 1 on mismatch.

Run:  python oracle/fuzz_vs_reference.py [n_configs]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from make_golden import REF, install_stubs, pose_spherical_np  # noqa: E402


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    install_stubs()
    sys.path.insert(0, REF)
    import render as R
    import run_nerf as RN
    import run_nerf_helpers as H
    from oracle import nerf_oracle as O
    rng = np.random.RandomState(7)
    c2w = pose_spherical_np(30.0, -30.0, 4.0)[:3, :4]
    bad = 0
    for ci in range(n_cfg):
        use_viewdirs = bool(rng.rand() < 0.75)
        Ns = int(rng.choice([2, 3, 8, 17, 32, 64]))
        Ni = int(rng.choice([0, 1, 5, 16, 33, 64]))
        if Ns < 3:
            Ni = 0        # the reference itself fails for N_importance > 0 with two coarse samples (empty inner weights)
        lindisp, white = bool(rng.rand() < 0.4), bool(rng.rand() < 0.5)
        perturb = float(rng.choice([0.0, 1.0]))
        noise_std = float(rng.choice([0.0, 0.0, 1.0]))
        ndc = bool(rng.rand() < 0.3)
        if ndc:
            lindisp = False   # near = 0 in NDC: sampling linearly in 1 / depth divides by zero on both sides
        shared = bool(Ni > 0 and rng.rand() < 0.25)

        class A:
            pass
        args = A()
        args.multires, args.multires_views, args.i_embed = 10, 4, 0
        args.use_viewdirs, args.N_importance, args.netdepth, args.netwidth = use_viewdirs, Ni, 8, 256
        args.netdepth_fine, args.netwidth_fine, args.netchunk = 8, 256, 65536
        args.lrate, args.basedir, args.expname, args.ft_path, args.no_reload = 5e-4, '/tmp', 'golden_tmp', None, True
        args.perturb, args.N_samples, args.white_bkgd, args.raw_noise_std = perturb, Ns, white, noise_std
        args.dataset_type, args.no_ndc, args.lindisp = ('llff' if ndc else 'blender'), False, lindisp
        os.makedirs('/tmp/golden_tmp', exist_ok=True)
        torch.manual_seed(100 + ci)
        so = sys.stdout
        sys.stdout = open(os.devnull, 'w')
        try:
            kw, _, _, _, grad_vars, _ = RN.create_nerf(args)
        finally:
            sys.stdout = so
        if shared:
            kw['network_fine'] = None
        Hh, Ww, focal = (378, 504, 400.0) if ndc else (800, 800, 1111.1)
        K = np.array([[focal, 0, 0.5 * Ww], [0, focal, 0.5 * Hh], [0, 0, 1]])
        pose = torch.tensor([[1.0, 0, 0, 0.05], [0, 1.0, 0, -0.03], [0, 0, 1.0, 0.1]]) if ndc else c2w
        o_b, d_b = H.get_rays(Hh, Ww, K, pose)
        g = torch.Generator().manual_seed(ci)
        sel = torch.randint(0, Hh * Ww, (24,), generator=g)
        ro, rd = o_b.reshape(-1, 3)[sel].contiguous(), d_b.reshape(-1, 3)[sel].contiguous()
        tgt = torch.rand(24, 3, generator=g)
        near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
        rgb, disp, acc, ex = R.render(Hh, Ww, K, chunk=32768, rays=torch.stack([ro, rd], 0), retraw=True, near=near, far=far,
                                      pytest=True, **kw)
        loss = H.img2mse(rgb, tgt) + (H.img2mse(ex['rgb0'], tgt) if 'rgb0' in ex else 0.0)
        nets = [kw['network_fn']] + ([kw['network_fine']] if kw['network_fine'] is not None else [])
        ps = [(n, p) for net in nets for n, p in net.named_parameters()]
        gr_ref = torch.autograd.grad(loss, [p for _, p in ps], allow_unused=True)
        # ---- oracle on the same weights / draws ----
        sds = [{k.replace('module.', ''): v.detach().clone() for k, v in net.state_dict().items()} for net in nets]
        for sd in sds:
            for v in sd.values():
                v.requires_grad_(True)
        rb = O.make_ray_batch(ro, rd, near, far, Hh, Ww, focal, ndc=ndc, use_viewdirs=use_viewdirs)

        def draw(shape):
            np.random.seed(0)
            return torch.from_numpy(np.random.rand(*shape).astype(np.float32))
        t_rand = draw((24, Ns)) if perturb > 0 else None
        u = draw((24, Ni)) if (perturb > 0 and Ni > 0) else None
        n0 = draw((24, Ns)) * noise_std if noise_std > 0 else None
        n1 = draw((24, Ns + Ni)) * noise_std if (noise_std > 0 and Ni > 0) else None
        ret = O.render_rays(rb, sds[0], sds[1] if len(sds) > 1 else None, Ns, Ni, lindisp, white, t_rand, u, n0, n1, retraw=True)
        lo = O.img2mse(ret['rgb_map'], tgt) + (O.img2mse(ret['rgb0'], tgt) if 'rgb0' in ret else 0.0)
        names = [n.replace('module.', '') for n, _ in ps]
        flat = [sds[0][n] for n in names[:len(list(nets[0].named_parameters()))]]
        if len(sds) > 1:
            flat += [sds[1][n] for n in names[len(flat):]]
        gr_or = torch.autograd.grad(lo, flat, allow_unused=True)
        worst, n_finite = 0.0, 0

        def cmp(a, b, what):
            nonlocal worst
            a, b = a.detach().numpy(), b.detach().numpy()
            if not np.array_equal(np.isnan(a), np.isnan(b)):
                return what + ':nan-pattern'
            m = ~np.isnan(a)
            nonlocal n_finite
            n_finite += int(m.sum())
            if m.any():
                e = np.abs(a[m] - b[m]).max() / max(1.0, np.abs(b[m]).max())
                worst = max(worst, float(e))
                if e > 2e-5:
                    return f'{what}:{e:.1e}'
            return None
        errs = [cmp(ret['rgb_map'], rgb, 'rgb'), cmp(ret['disp_map'], disp, 'disp'), cmp(ret['acc_map'], acc, 'acc'),
                cmp(ret['raw'], ex['raw'], 'raw')]
        if 'rgb0' in ex:
            errs += [cmp(ret['rgb0'], ex['rgb0'], 'rgb0'), cmp(ret['z_std'], ex['z_std'], 'z_std')]
        gw = 0.0
        for (n, _), a, b in zip(ps, gr_or, gr_ref):
            if (a is None) != (b is None):
                errs.append('grad-none:' + n)
            elif a is not None:
                e = float((a - b).abs().max() / max(float(b.abs().max()), 1e-8))
                gw = max(gw, e)
                if e > 5e-4:
                    errs.append(f'grad {n}:{e:.1e}')
        errs = [e for e in errs if e]
        bad += bool(errs)
        print(f'cfg {ci:2d} viewdirs={int(use_viewdirs)} Ns={Ns:2d} Ni={Ni:2d} lindisp={int(lindisp)} white={int(white)} perturb={perturb:.0f} '
              f'noise={noise_std:.0f} ndc={int(ndc)} shared={int(shared)}: out {worst:.1e} ({n_finite} finite values) grad {gw:.1e} ' + ('OK' if not errs else 'MISMATCH ' + ' '.join(errs)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
