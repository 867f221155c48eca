"""Is the GPU-vs-CPU PSNR gap at 200 iterations a bias or trajectory divergence?  (round 3)

 lockstep : before every iteration the GPU trainer takes over the CPU oracle's weights and Adam moments, both take the same
            step on the same batch; reports per-step loss differences and the relative difference of the updates.
 free     : both start from the same weights and run freely; per-iteration fine losses side by side (GPU fp32, GPU bf16x3, CPU).
Runs ON THE GPU BOX (needs both)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fastnerf
from fastnerf import ops, synthetic
from oracle import nerf_oracle as O
import bench as B

dev = torch.device('cuda:0')
N = int(os.environ.get('PS_RAYS', '512'))
ITERS = int(os.environ.get('PS_ITERS', '40'))
torch.set_num_threads(32)
K = np.array([[B.FOCAL, 0, 400.0], [0, B.FOCAL, 400.0], [0, 0, 1]])
poses = torch.stack([synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
g = torch.Generator().manual_seed(2)
batches = []
for it in range(ITERS):
    pix = torch.stack([torch.randint(0, 100, (N,), generator=g), torch.randint(0, 800, (N,), generator=g), torch.randint(0, 800, (N,), generator=g)], 1).int()
    ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
    batches.append((ro, rd, synthetic.render_rays(ro, rd, cutoff=0.0), torch.rand(N, 64, generator=g).to(dev), torch.rand(N, 128, generator=g).to(dev)))
args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)


def new_gpu(mode):
    ops.set_math(mode); fastnerf.render.set_compact('0')
    torch.manual_seed(0)
    k = fastnerf.run_nerf.create_nerf(args, device=dev)[0]
    return fastnerf.run_nerf.Trainer(k, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500), k


def cpu_state(k):
    sdc = {n: v.detach().cpu().clone() for n, v in k['network_fn'].state_dict().items()}
    sdf = {n: v.detach().cpu().clone() for n, v in k['network_fine'].state_dict().items()}
    return sdc, sdf, O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)


def flat_of(sdc, sdf):
    return torch.cat([v.reshape(-1) for v in list(sdc.values()) + list(sdf.values())])


what = sys.argv[1] if len(sys.argv) > 1 else 'lockstep'
if what == 'lockstep':
    for mode in ('fp32', 'bf16x3'):
        tr, k = new_gpu(mode)
        sdc, sdf, opt = cpu_state(k)
        print(f'--- lockstep, GPU mode {mode}: it, loss_gpu, loss_cpu, rel dloss, |dW_gpu - dW_cpu| / |dW_cpu| (L2), max|dw| / lr, grad rel L2', flush=True)
        for it, (ro, rd, tgt, t_rand, u) in enumerate(batches):
            w0 = flat_of(sdc, sdf)
            with torch.no_grad():
                tr.flat.copy_(w0.to(dev))
                tr.m.copy_(torch.cat([m.reshape(-1) for m in opt.m]).to(dev)); tr.v.copy_(torch.cat([v.reshape(-1) for v in opt.v]).to(dev))
            tr.adam_t = opt.t
            tr.repack()
            lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4
            tr.lr = opt.lr = lr
            l2, _ = tr.step(ro, rd, tgt, t_rand=t_rand, u=u, decay=False)
            l1, l0, _, grads = O.train_step(sdc, sdf, opt, O.make_ray_batch(ro.cpu(), rd.cpu(), 2.0, 6.0), tgt.cpu(), 64, 128, True, t_rand=t_rand.cpu(), u=u.cpu())
            w1c = flat_of(sdc, sdf)
            w1g = tr.flat.cpu()
            dc, dg = w1c - w0, w1g - w0
            gc = torch.cat([x.reshape(-1) for x in grads]); gg = tr.grad.cpu()
            print(it, f'{float(l2[0]):.6f} {float(l1):.6f} {abs(float(l2[0]) - float(l1)) / float(l1):.2e} '
                  f'{float((dg - dc).norm() / dc.norm()):.3e} {float((dg - dc).abs().max() / lr):.3f} {float((gg - gc).norm() / gc.norm()):.3e} '
                  f'sum(dg)/sum(dc) {float(dg.sum() / dc.sum()):.6f}', flush=True)
else:
    res = {}
    for mode in ('fp32', 'bf16x3'):
        tr, k = new_gpu(mode)
        if mode == 'fp32':
            sdc, sdf, opt = cpu_state(k)
        ls = []
        for ro, rd, tgt, t_rand, u in batches:
            ls.append(tr.step(ro, rd, tgt, t_rand=t_rand, u=u)[0][0])
        res[mode] = torch.stack(ls).cpu().numpy()
    lc = []
    for it, (ro, rd, tgt, t_rand, u) in enumerate(batches):
        opt.lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4
        l1 = O.train_step(sdc, sdf, opt, O.make_ray_batch(ro.cpu(), rd.cpu(), 2.0, 6.0), tgt.cpu(), 64, 128, True, t_rand=t_rand.cpu(), u=u.cpu())[0]
        lc.append(float(l1))
        print(it, f'fp32 {res["fp32"][it]:.6f} bf16x3 {res["bf16x3"][it]:.6f} cpu {lc[-1]:.6f}  rel(fp32-cpu) {(res["fp32"][it] - lc[-1]) / lc[-1]:+.2e} '
              f'rel(bf16x3-fp32) {(res["bf16x3"][it] - res["fp32"][it]) / res["fp32"][it]:+.2e}', flush=True)
