#!/usr/bin/env python3
"""Golden vectors for nerf++'s full-image evaluation path, render_single_image (ddp_test_nerf.py:126-227), recorded from
the REFERENCE itself (build container only; stubs as in make_golden.py): a 6x8 image, two cascade levels with the G10
weights, ragged chunks.  Data-only fixture -> tests/golden/g14_pp_render.npz."""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import install_stubs, OUT  # noqa: E402
from make_golden_pp import Args, REF  # noqa: E402


def main():
    install_stubs()
    torch.cuda.empty_cache = lambda: None
    lp = type(sys)('lpips')   # the module builds an LPIPS net at import time; never used by render_single_image

    class _LP:
        def __init__(self, **kw): pass
        def eval(self): return self
        def cuda(self): return self
    lp.LPIPS = _LP
    sys.modules['lpips'] = lp
    sys.path.insert(0, REF)
    import ddp_model as M
    import nerf_sample_ray_split as RS
    import ddp_test_nerf as T
    wts = np.load(os.path.join(OUT, 'g10_pp_weights.npz'))
    nets = [M.NerfNetWithAutoExpo(Args, optim_autoexpo=False) for _ in range(2)]
    for m, nt in enumerate(nets):
        nt.nerf_net.load_state_dict({k[3:]: torch.from_numpy(wts[k]) for k in wts.files if k.startswith('l%d.' % m)})
    H, W = 6, 8
    intr = np.array([[30.0, 0, 4.0, 0], [0, 30.0, 3.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    c2w = np.eye(4); c2w[:3, 3] = [0.1, -0.2, 0.3]
    ro, rd, _ = RS.get_rays_single_image(H, W, intr, c2w)
    ro, rd = ro.astype(np.float32), rd.astype(np.float32)   # RaySamplerSingleImage.get_all hands out float32

    class Sampler:
        pass
    s = Sampler()
    s.H, s.W = H, W
    s.get_all = lambda: OrderedDict([('ray_o', torch.from_numpy(ro)), ('ray_d', torch.from_numpy(rd)), ('depth', None),
                                     ('rgb', None), ('mask', None),
                                     ('min_depth', torch.from_numpy(1e-4 * np.ones_like(rd[..., 0])))])
    models = {'cascade_level': 2, 'cascade_samples': [64, 128], 'net_0': nets[0], 'net_1': nets[1]}
    ret = T.render_single_image(models, s, 20)   # chunks 20, 20, 8
    rec = {'intr': intr, 'c2w': c2w, 'ray_o': ro, 'ray_d': rd}
    for m in range(2):
        for k, v in ret[m].items():
            rec['l%d.%s' % (m, k)] = v.numpy()
    np.savez(os.path.join(OUT, 'g14_pp_render.npz'), **rec)
    print({k: v.shape for k, v in rec.items()})


if __name__ == '__main__':
    main()
