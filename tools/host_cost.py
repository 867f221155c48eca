"""Host cost of one optimisation step: time to ENQUEUE k steps after a synchronise (short bursts, so that the HIP queue
never fills and the host is never throttled by the GPU), fused one-call route vs the call-by-call route."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fastnerf
from fastnerf import ops, synthetic
dev = torch.device('cuda:0')
K_BURST = int(os.environ.get('K_BURST', '3'))
for N in (512, 4096):
    args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
    H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
    gen = torch.Generator().manual_seed(1000)
    pix = torch.stack([torch.randint(0, 100, (N,), generator=gen), torch.randint(0, H, (N,), generator=gen), torch.randint(0, W, (N,), generator=gen)], 1).int()
    ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
    tgt = torch.rand(N, 3, generator=gen).to(dev)
    tag = torch.stack([pix[:, 0], (pix[:, 1] // 50) * 16 + pix[:, 2] // 50], 1).int().to(dev).contiguous()
    table = torch.zeros(100 * 256, device=dev, dtype=torch.int32)
    for mode in ('0', '1'):
        for fused in (True, False):
            fastnerf.render.set_compact(mode)
            torch.manual_seed(0)
            tr = fastnerf.run_nerf.Trainer(fastnerf.run_nerf.create_nerf(args, device=dev)[0], H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
            tr.fused = fused
            for i in range(10): tr.step(ro, rd, tgt, leaf_tag=tag, table=table, max_leaves=256)
            host, wall = [], []
            for rep in range(15):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(K_BURST): tr.step(ro, rd, tgt, leaf_tag=tag, table=table, max_leaves=256)
                t1 = time.perf_counter()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                host.append((t1 - t0) / K_BURST); wall.append((t2 - t0) / K_BURST)
            print(f'N={N} math={ops.get_math()} compact={mode} fused={fused}: host enqueue {1e3 * np.median(host):.3f} ms/step '
                  f'(min {1e3 * min(host):.3f}), wall {1e3 * np.median(wall):.3f} ms/step', flush=True)
