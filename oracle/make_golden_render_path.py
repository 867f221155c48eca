#!/usr/bin/env python3
"""G18: the evaluation loop render_path (nerf-ours/render.py:94-146) recorded from the REFERENCE (build container only):
two 6x8 frames of the G7 nets (weights in tests/golden/g7_weights.npz) from two pose_spherical cameras, test-mode kwargs
(perturb 0, deterministic inverse-CDF), ground-truth images given -> the returned rgbs / disps, the per-frame PSNR / SSIM
the reference prints into results.txt, and the 8-bit frames it writes.  LPIPS is an external package (absent): stood in
for by a constant, and not part of the fixture.  Data-only -> tests/golden/g18_render_path.npz."""
import os
import re
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF, install_stubs, pose_spherical_np  # noqa: E402


def main():
    install_stubs()
    lp = type(sys)('lpips')

    class _LP:
        def __init__(self, **kw): pass
        def eval(self): return self
        def cuda(self): return self
        def __call__(self, a, b, normalize=True): return torch.zeros(1)
    lp.LPIPS = _LP
    sys.modules['lpips'] = lp
    written = {}
    sys.modules['imageio'].imwrite = lambda path, arr: written.__setitem__(os.path.basename(path), np.array(arr))
    sys.path.insert(0, REF)
    import render as R
    import run_nerf as RN

    class A:
        pass
    args = A()
    args.multires, args.multires_views, args.i_embed = 10, 4, 0
    args.use_viewdirs, args.N_importance, args.netdepth, args.netwidth = True, 128, 8, 256
    args.netdepth_fine, args.netwidth_fine, args.netchunk = 8, 256, 65536
    args.lrate, args.basedir, args.expname, args.ft_path, args.no_reload = 5e-4, '/tmp', 'golden_tmp', None, True
    args.perturb, args.N_samples, args.white_bkgd, args.raw_noise_std = 1.0, 64, True, 0.0
    args.dataset_type, args.no_ndc, args.lindisp = 'blender', False, False
    os.makedirs('/tmp/golden_tmp', exist_ok=True)
    kw_train, kw_test, _, _, _, _ = RN.create_nerf(args)
    w = np.load(os.path.join(OUT, 'g7_weights.npz'))
    kw_test['network_fn'].load_state_dict({'module.' + k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith('c.')})
    kw_test['network_fine'].load_state_dict({'module.' + k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith('f.')})
    kw_test.update(near=2.0, far=6.0)
    H, W, focal = 6, 8, 9.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([pose_spherical_np(30.0, -30.0, 4.0), pose_spherical_np(-70.0, -20.0, 3.5)], 0)
    g = torch.Generator().manual_seed(18)
    gt = torch.rand(2, H, W, 3, generator=g).numpy()
    with tempfile.TemporaryDirectory() as d:
        with torch.no_grad():
            rgbs, disps = R.render_path(poses, [H, W, focal], K, 1024, kw_test, gt_imgs=gt, savedir=d)
        results = open(os.path.join(d, 'results.txt')).read()
    # per-frame metrics, recomputed the way render.py:116-121 does (the file only holds the means)
    import run_nerf_helpers as RH
    psnr = [float(-10. * np.log10(np.mean(np.square(rgbs[i] - gt[i])))) for i in range(2)]
    ssim = [float(RH.compute_ssim(torch.tensor(gt[i]).float(), torch.tensor(rgbs[i])).item()) for i in range(2)]
    m = re.search(r'mean PSNR: (\S+)\nmean SSIM: (\S+)', results)
    np.savez(os.path.join(OUT, 'g18_render_path.npz'), poses=poses.numpy(), K=K, hwf=np.array([H, W, focal]), gt=gt,
             rgbs=rgbs, disps=disps, psnr=np.array(psnr), ssim=np.array(ssim), mean_psnr=float(m.group(1)),
             mean_ssim=float(m.group(2)), png0=written['000.png'], png1=written['001.png'])
    print(results, psnr, ssim)


if __name__ == '__main__':
    main()
