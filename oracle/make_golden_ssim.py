#!/usr/bin/env python3
"""Golden vector for the evaluation metric compute_ssim, recorded from the REFERENCE itself
(build container only).  Data-only fixture -> tests/golden/g13_ssim.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import install_stubs, OUT  # noqa: E402

REF = '/root/reference/nerf-ours'


def main():
    assert os.path.isdir(REF)
    install_stubs()
    sys.path.insert(0, REF)
    import run_nerf_helpers as RH
    g = torch.Generator().manual_seed(1313)
    a = torch.rand(37, 29, 3, generator=g)
    b = (a + 0.1 * torch.randn(37, 29, 3, generator=g)).clamp(0, 1)
    c = torch.rand(2, 16, 12, 3, generator=g)       # batched, smaller than twice the window
    d = torch.rand(2, 16, 12, 3, generator=g)
    np.savez(os.path.join(OUT, 'g13_ssim.npz'),
             a=a.numpy(), b=b.numpy(), c=c.numpy(), d=d.numpy(),
             ssim_ab=RH.compute_ssim(a, b).numpy(), map_ab=RH.compute_ssim(a, b, return_map=True).numpy(),
             ssim_aa=RH.compute_ssim(a, a).numpy(), ssim_cd=RH.compute_ssim(c, d).numpy(),
             ssim_cd_k=RH.compute_ssim(c, d, max_val=2.0, filter_size=7, filter_sigma=1.0, k1=0.02, k2=0.05).numpy())
    print('wrote g13_ssim.npz')


if __name__ == '__main__':
    main()
