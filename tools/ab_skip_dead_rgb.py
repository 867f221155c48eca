import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fastnerf
from fastnerf import ops, synthetic
dev = torch.device('cuda:0'); N = 4096
args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
poses = torch.stack([synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
gen = torch.Generator().manual_seed(1000); batches = []
for _ in range(64):
    pix = torch.stack([torch.randint(0, 100, (N,), generator=gen), torch.randint(0, H, (N,), generator=gen), torch.randint(0, W, (N,), generator=gen)], 1).int()
    ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
    batches.append((ro, rd, synthetic.render_rays(ro, rd, cutoff=1.5).contiguous()))
torch.manual_seed(0)
tr = fastnerf.run_nerf.Trainer(fastnerf.run_nerf.create_nerf(args, device=dev)[0], H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
for i in range(600): tr.step(*batches[i % 64])
for rep in range(3):
    for skip in (False, True):
        tr.skip_dead_rgb = skip
        for i in range(10): tr.step(*batches[i % 64])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100): tr.step(*batches[i % 64])
        torch.cuda.synchronize()
        print(f'skip={skip}: {(time.perf_counter() - t0) * 10:.3f} ms/step live={tr.last_step_live}', flush=True)
# host-side enqueue time of a step (the GPU must never wait for the host): 100 steps issued without synchronising
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): tr.step(*batches[i % 64])
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f'host enqueue {t_host * 10:.3f} ms/step, GPU-bound wall {t_all * 10:.3f} ms/step', flush=True)
