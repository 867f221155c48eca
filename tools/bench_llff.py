"""BASELINE configs[3] shape on one GPU: LLFF-like forward-facing cameras, 1008x756, NDC rays, 64 + 64 samples,
raw_noise_std = 1, 4096 rays / step.  Not the headline bench (bench.py is); DESIGN.md cites it."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf as fn
from fastnerf import ops
N = 4096
dev = torch.device('cuda')
H, W, focal = 756, 1008, 815.13
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
for mode in (sys.argv[1:] or ['fp32', 'bf16x3']):
    ops.set_math(mode)
    torch.manual_seed(0)
    args = fn.run_nerf.make_args(N_importance=64, N_samples=64, perturb=1.0, raw_noise_std=1.0, no_reload=True, dataset_type='llff',
                                 lrate=5e-4, lrate_decay=250)
    ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args, device=dev)
    poses = torch.eye(4)[None, :3, :4].repeat(20, 1, 1)
    poses[:, 0, 3] = torch.linspace(-0.3, 0.3, 20)
    poses = poses.to(dev)
    g = torch.Generator().manual_seed(1)
    pix = torch.stack([torch.randint(0, 20, (N,), generator=g), torch.randint(0, H, (N,), generator=g),
                       torch.randint(0, W, (N,), generator=g)], 1).int().to(dev)
    ro, rd = ops.gen_rays_pixels(pix, poses, K)
    tgt = torch.rand(N, 3, generator=g).to(dev)
    tr = fn.run_nerf.Trainer(ktr, H, W, K, 0.0, 1.0, lrate=5e-4, lrate_decay=250)
    for _ in range(3): tr.step(ro, rd, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Kst = 20
    for _ in range(Kst): loss, _ = tr.step(ro, rd, tgt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / Kst
    c = tr.live_counts.cpu().tolist()
    print('%-7s ndc=%s 64+64: %.2f ms/step  %.0f rays/s  loss %s  backward %s live %s' % (mode, tr.ndc, dt * 1e3, N / dt, [round(float(x), 5) for x in loss],
          'compacted' if tr.last_step_live else 'plain', [round(c[0] / max(1, c[1]), 3), round(c[2] / max(1, c[3]), 3)]))
