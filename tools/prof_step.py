#!/usr/bin/env python3
"""Whole optimisation steps at the bench size for rocprofv3 (kernel trace / PMC passes), in the stationary state of training the
analytic solid-body scene -- the regime bench.py's `value` is measured in.

  prof_step.py train <state.pt> [n]     n (600) optimisation steps from random init on the analytic scene; saves the trainer state
  prof_step.py steps <state.pt> <k>     loads the state, runs exactly k steps (FASTNERF_COMPACT = 1 / 0 / auto picks the backward)
  prof_step.py kernels <state.pt> [r]   stand-alone fine-pass launches on the trained fine net: forward without saving, saving
                                        forward, saving forward over the live list, backward plain / over the live list (r reps)
Same cameras, batches and tags as bench.py (seed 1000)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf  # noqa: E402
from fastnerf import ops, synthetic  # noqa: E402

N_RAYS, NS, NI = 4096, 64, 128
SCENE_CUTOFF = 1.5   # bench.py's solid-body scene


def setup():
    dev = torch.device('cuda')
    args = fastnerf.run_nerf.make_args(N_importance=NI, N_samples=NS, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4,
                                       lrate_decay=500)
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
    gen = torch.Generator().manual_seed(1000)
    batches = []
    for _ in range(64):
        pix = torch.stack([torch.randint(0, 100, (N_RAYS,), generator=gen), torch.randint(0, H, (N_RAYS,), generator=gen),
                           torch.randint(0, W, (N_RAYS,), generator=gen)], 1).int()
        ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
        tag = torch.stack([pix[:, 0], (pix[:, 1] // 50) * 16 + pix[:, 2] // 50], 1).int().to(dev).contiguous()
        torch.rand(N_RAYS, 3, generator=gen)   # (bench.py draws its noise targets here: keep the streams aligned)
        batches.append((ro, rd, synthetic.render_rays(ro, rd, cutoff=SCENE_CUTOFF).contiguous(), tag))
    torch.manual_seed(0)
    ktr, _, _, _, _, _ = fastnerf.run_nerf.create_nerf(args, device=dev)
    tr = fastnerf.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    table = torch.zeros(100 * 256, device=dev, dtype=torch.int32)
    return tr, batches, table


def main():
    what, path = sys.argv[1], sys.argv[2]
    tr, batches, table = setup()

    def step(i):
        ro, rd, tgt, tag = batches[i % 64]
        return tr.step(ro, rd, tgt, leaf_tag=tag, table=table, max_leaves=256)
    if what == 'train':
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 600
        for i in range(n):
            loss2, _ = step(i)
        torch.cuda.synchronize()
        torch.save({'flat': tr.flat.cpu(), 'm': tr.m.cpu(), 'v': tr.v.cpu(), 'adam_t': tr.adam_t, 'global_iter': tr.global_iter,
                    'lr': tr.lr, 'n': n}, path)
        print('trained', n, 'steps, loss', loss2.tolist())
        return
    st = torch.load(path)
    with torch.no_grad():
        tr.flat.copy_(st['flat'].cuda()); tr.m.copy_(st['m'].cuda()); tr.v.copy_(st['v'].cuda())
    tr.adam_t, tr.global_iter, tr.lr = st['adam_t'], st['global_iter'], st['lr']
    tr.repack()
    if what == 'steps':
        k = int(sys.argv[3])
        for i in range(k):
            loss2, _ = step(st['n'] + i)
        torch.cuda.synchronize()
        c = tr.live_counts.cpu().tolist()
        print('PROFSTEP steps=%d compact=%s live=%s loss=%s' % (k, tr.last_step_live, c, loss2.tolist()))
        return
    # stand-alone fine-pass launches on a real batch of the trained nets
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ro, rd, tgt, tag = batches[0]
    rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
    out, _ = fastnerf.render._forward_core(rays11, tr.net_c, tr.net_f, NS, NI, False, 1.0, True, None, None, None, None, save=False,
                                           packed_c=tr.pc, packed_f=tr.pf)
    z = out['z_vals'].contiguous()
    loss2, g, g0 = ops.mse_leafmax(out['rgb_map'], out['rgb0'], tgt)
    draw = ops.raw2outputs_bwd(out['raw'], z, rays11, g, None, True)
    idx, cnt = ops.compact_live(draw)
    P = N_RAYS * (NS + NI)
    act = torch.empty(ops.act_floats(P), device='cuda')
    dact = torch.empty(ops.dact_floats(P), device='cuda')
    partial = torch.empty(ops.mlp_bwd_partial_floats(), device='cuda')
    grads = torch.empty(ops.NET_PARAMS, device='cuda')
    raw = torch.empty(N_RAYS, NS + NI, 4, device='cuda')
    for _ in range(reps):
        ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], raw=raw)
        ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], act=act, raw=raw)
        ops.mlp_bwd(draw, act, tr.net_f.flat, tr.pf[1], dact, partial, grads)
        ops.mlp_fwd_live(rays11, z, tr.net_f.flat, tr.pf[0], act, idx, cnt)
        ops.mlp_bwd_live(draw, act, tr.net_f.flat, tr.pf[1], dact, partial, grads, idx, cnt)
    torch.cuda.synchronize()
    print('PROFKERNELS live=%s' % cnt.cpu().tolist())


if __name__ == '__main__':
    main()
