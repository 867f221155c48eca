"""Per-seed PSNR of the toy 300-iteration protocol of tests/test_gpu_train.py::test_psnr_300_iterations_both_modes for fp32,
bf16x3 and fp32 with 1-ulp-jittered initial weights: what does the distribution of per-seed differences look like?"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fastnerf as fn
from oracle import nerf_oracle as O
imgs, poses, focal = fn.synthetic.make_dataset(n_images=6, H=24, W=24)
H = W = 24
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
rays = [O.get_rays(H, W, K, poses[i]) for i in range(6)]
ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3).cuda()
rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3).cuda()
tgt_all = imgs.reshape(-1, 3).cuda()
n_iters, N = int(os.environ.get('N_ITERS', 300)), 192
n_seeds = int(os.environ.get('N_SEEDS', 64))
fn.render.set_compact('0')
out = {'fp32': [], 'bf16x3': [], 'fp32_jit': []}
for seed in range(n_seeds):
    gen = torch.Generator().manual_seed(100 + seed)
    sched = [(torch.randint(0, ro_all.shape[0], (N,), generator=gen).cuda(), torch.rand(N, 16, generator=gen).cuda(), torch.rand(N, 16, generator=gen).cuda())
             for it in range(n_iters)]
    for key in out:
        fn.ops.set_math('bf16x3' if key == 'bf16x3' else 'fp32')
        torch.manual_seed(seed)
        args = fn.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
        ktr = fn.run_nerf.create_nerf(args)[0]
        tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
        if key == 'fp32_jit':
            g = torch.Generator(device='cuda').manual_seed(7 + seed)
            with torch.no_grad():
                bits = tr.flat.view(torch.int32)
                bits += torch.randint(-1, 2, bits.shape, generator=g, device='cuda', dtype=torch.int32)
            tr.repack()
        ls = torch.stack([tr.step(ro_all[s], rd_all[s], tgt_all[s], t_rand=t, u=u)[0][0] for s, t, u in sched]).cpu().numpy()
        out[key].append([float(-10 * np.log10(np.mean(ls[a:b]))) for a, b in ((50, 100), (150, 200), (250, 300))])
json.dump(out, open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'psnr_seed_study.json'), 'w'))
for k in out:
    print(k, np.round(np.array(out[k])[:, 2], 2).tolist())
