#!/usr/bin/env python3
"""Round-3 profiling workloads (SURVEY 8(d) throughput protocol: random-init nets, uniformly drawn rays, U[0,1) targets, plain
backward -- what bench.py's `value` runs), for rocprofv3:
  prof_r03.py steps <mode> <k>     k optimisation steps of 4096 rays x (64+128) samples in math mode <mode> (after 3 warm-up steps
                                   whose kernels are marked off by a `pack_rays_kernel` count in the summariser)
  prof_r03.py kernels <mode> <r>   stand-alone fine-pass launches: forward without saving, saving forward, backward (r reps)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf  # noqa: E402
from fastnerf import ops  # noqa: E402
import bench as B  # noqa: E402

N, NS, NI = 4096, 64, 128
WARM = 3


def main():
    what, mode, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dev = torch.device('cuda')
    ops.set_math(mode)
    fastnerf.render.set_compact('0')
    K = np.array([[B.FOCAL, 0, 400.0], [0, B.FOCAL, 400.0], [0, 0, 1]])
    poses = torch.stack([fastnerf.synthetic.pose_spherical(-180.0 + 3.6 * i, -30.0, 4.0)[:3, :4] for i in range(100)], 0).to(dev)
    gen = torch.Generator().manual_seed(1000)
    batches = []
    for _ in range(8):
        pix = torch.stack([torch.randint(0, 100, (N,), generator=gen), torch.randint(0, 800, (N,), generator=gen),
                           torch.randint(0, 800, (N,), generator=gen)], 1).int()
        ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
        tag = torch.stack([pix[:, 0], (pix[:, 1] // 50) * 16 + pix[:, 2] // 50], 1).int().to(dev).contiguous()
        batches.append((ro, rd, torch.rand(N, 3, generator=gen).to(dev), tag))
    args = fastnerf.run_nerf.make_args(N_importance=NI, N_samples=NS, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4,
                                       lrate_decay=500)
    torch.manual_seed(0)
    tr = fastnerf.run_nerf.Trainer(fastnerf.run_nerf.create_nerf(args, device=dev)[0], 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    table = torch.zeros(100 * 256, device=dev, dtype=torch.int32)
    if what == 'steps':
        for i in range(WARM + k):
            ro, rd, tgt, tag = batches[i % 8]
            loss2, _ = tr.step(ro, rd, tgt, leaf_tag=tag, table=table, max_leaves=256)
        torch.cuda.synchronize()
        print('PROFSTEP mode=%s steps=%d (+%d warm-up) loss=%s' % (mode, k, WARM, loss2.tolist()))
        return
    ro, rd, tgt, tag = batches[0]
    rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
    S1 = NS + NI
    z = torch.sort(torch.rand(N, S1, device=dev) * 4 + 2, -1).values
    P = N * S1
    act = torch.empty(ops.act_floats(P), device=dev)
    dact = torch.empty(ops.dact_floats(P), device=dev)
    partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
    grads = torch.empty(ops.NET_PARAMS, device=dev)
    raw = torch.empty(N, S1, 4, device=dev)
    draw = torch.randn(N, S1, 4, device=dev) * 1e-4
    for _ in range(k):
        ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], raw=raw)
        ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], act=act, raw=raw)
        ops.mlp_bwd(draw, act, tr.net_f.flat, tr.pf[1], dact, partial, grads)
    torch.cuda.synchronize()
    print('PROFKERNELS mode=%s reps=%d' % (mode, k))


if __name__ == '__main__':
    main()
