import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import fastnerf as fn
from oracle import nerf_oracle as O
imgs, poses, focal = fn.synthetic.make_dataset(n_images=6, H=24, W=24)
H = W = 24
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
rays = [O.get_rays(H, W, K, poses[i]) for i in range(6)]
ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3); rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3)
tgt_all = imgs.reshape(-1, 3)
res = {}
for mode in ('cpu', 'fp32', 'bf16x6', 'bf16x3'):
    torch.manual_seed(0)
    args = fn.run_nerf.make_args(N_importance=16, N_samples=16, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
    if mode != 'cpu': fn.ops.set_math(mode)
    ktr = fn.run_nerf.create_nerf(args)[0]
    sdc = {k: v.detach().cpu().clone() for k, v in ktr['network_fn'].state_dict().items()}
    sdf = {k: v.detach().cpu().clone() for k, v in ktr['network_fine'].state_dict().items()}
    tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
    gen = torch.Generator().manual_seed(1)
    ls = []
    for it in range(40):
        sel = torch.randint(0, ro_all.shape[0], (192,), generator=gen)
        t_rand, u = torch.rand(192, 16, generator=gen), torch.rand(192, 16, generator=gen)
        ro, rd, tgt = ro_all[sel], rd_all[sel], tgt_all[sel]
        if mode == 'cpu':
            opt.lr = O.lr_schedule(5e-4, 500, it - 1) if it > 0 else 5e-4
            l1 = O.train_step(sdc, sdf, opt, O.make_ray_batch(ro, rd, 2.0, 6.0), tgt, 16, 16, True, t_rand=t_rand, u=u)[0]
            ls.append(float(l1))
        else:
            ls.append(float(tr.step(ro.cuda(), rd.cuda(), tgt.cuda(), t_rand=t_rand.cuda(), u=u.cuda())[0][0]))
    res[mode] = np.array(ls)
c = res['cpu']
for m in ('fp32', 'bf16x6', 'bf16x3'):
    d = np.abs(res[m] - c) / c
    print(m, 'rel diff at it 0,5,10,15,20,25,30,35,39:', ' '.join('%.1e' % d[i] for i in (0, 5, 10, 15, 20, 25, 30, 35, 39)), ' psnr(last5) %.3f vs cpu %.3f' % (-10 * np.log10(res[m][-5:].mean()), -10 * np.log10(c[-5:].mean())))
