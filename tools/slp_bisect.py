#!/usr/bin/env python3
"""Bisection of the SLP-vectoriser corruption of the split-bf16 training forward.

Witness: at bench size (4096 x 192 points) the saving forward must produce the same raw outputs as the non-saving one,
bit for bit, on every launch.  Run once per library variant (FASTNERF_LIB=...), prints one line:
    <lib> launches=<n> bad_launches=<k> bad_points=<total> first_bad=<tile, lane pattern>
Variants are built by tools/slp_bisect.sh: SLP on; SLP on + 34 idle states at every k-loop exit (BF_DBG_DRAIN=1);
SLP on + s_waitcnt vmcnt(0) lgkmcnt(0) at every k-loop exit (BF_DBG_DRAIN=2); SLP on without the permlane32_swap
stores (BF_W16=0); SLP off (the product build)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf  # noqa: E402


def main():
    n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    fastnerf.ops.set_math('bf16x3')
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'g7_weights.npz'))
    names = [k for k in g.files if k.startswith('f.')]
    from fastnerf.model import param_slices
    flat = torch.cat([torch.from_numpy(g['f.' + n]).reshape(-1) for n, _, _ in param_slices()]).cuda()
    pf, pb = fastnerf.ops.mlp_pack(flat)
    gen = torch.Generator().manual_seed(0)
    n, S = 4096, 192
    ro = torch.randn(n, 3, generator=gen) * 0.3
    rd = torch.randn(n, 3, generator=gen)
    rays = torch.zeros(n, 11)
    rays[:, 0:3], rays[:, 3:6] = ro, rd
    rays[:, 6], rays[:, 7] = 2.0, 6.0
    rays[:, 8:11] = rd / rd.norm(dim=-1, keepdim=True)
    rays = rays.cuda()
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.cuda()
    act = torch.empty(fastnerf.ops.act_floats(n * S)).cuda()
    ref = fastnerf.ops.mlp_fwd(rays, z, flat, pf).clone()
    bad_l = bad_p = 0
    first = None
    for it in range(n_launch):
        r = fastnerf.ops.mlp_fwd(rays, z, flat, pf, act=act)
        ne = (r != ref).any(-1).reshape(-1)
        k = int(ne.sum())
        if k:
            bad_l += 1
            bad_p += k
            if first is None:
                p = int(torch.nonzero(ne)[0])
                first = 'point %d (tile %d row %d) delta %.3e' % (p, p // 64, p % 64, float((r.reshape(-1, 4)[p] - ref.reshape(-1, 4)[p]).abs().max()))
        r2 = fastnerf.ops.mlp_fwd(rays, z, flat, pf)
        assert torch.equal(r2, ref), 'the non-saving forward is not reproducible either'
    print('SLPBISECT %s launches=%d bad_launches=%d bad_points=%d first_bad=%s' % (
        os.environ.get('FASTNERF_LIB', 'product'), n_launch, bad_l, bad_p, first))


if __name__ == '__main__':
    main()
