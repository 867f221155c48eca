"""Live-sample fraction, backward kind and step time along a long optimisation of the bench's analytic scene (the bench's own
batches and trainer): how long does the compaction of the backward keep paying?   python tools/live_trajectory.py [steps] [every]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf  # noqa: E402
from fastnerf import ops, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device('cuda:0')
N = 4096
args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4,
                                   lrate_decay=500)
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
poses = torch.stack([synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
gen = torch.Generator().manual_seed(1000)
batches = []
for _ in range(64):
    pix = torch.stack([torch.randint(0, 100, (N,), generator=gen), torch.randint(0, H, (N,), generator=gen),
                       torch.randint(0, W, (N,), generator=gen)], 1).int()
    ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
    batches.append((ro, rd, synthetic.render_rays(ro, rd).contiguous()))
torch.manual_seed(0)
ktr = fastnerf.run_nerf.create_nerf(args, device=dev)[0]
tr = fastnerf.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
counts = torch.zeros(4, device=dev, dtype=torch.int32)
t0 = time.perf_counter()
for i in range(steps):
    ro, rd, tgt = batches[i % 64]
    loss2, _ = tr.step(ro, rd, tgt)
    if (i + 1) % every == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / every
        # an explicit compacted probe step for the exact live counts of this state (not timed)
        old = fastnerf.render.get_compact()
        fastnerf.render.set_compact('1')
        tr.forward_backward(ro, rd, tgt)
        fastnerf.render.set_compact(old)
        c = tr.live_counts.cpu().tolist()
        print(f'step {i + 1:6d}: {1e3 * dt:6.2f} ms/step  backward {"compacted" if tr.last_step_live else "plain":9s} policy.on={tr.live.on}  '
              f'live fine {c[0] / max(1, c[1]):.3f} coarse {c[2] / max(1, c[3]):.3f}  psnr {-10 * np.log10(float(loss2[0])):.1f} dB', flush=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
