"""How many 64-point tiles of a training batch have an exactly-zero upstream gradient (raw sigma < 0 everywhere: alpha = 0,
weights = 0, relu'(sigma) = 0) as training progresses?  Synthetic analytic scene, the whole train() loop; after every epoch
one 4096-ray batch is pushed through the forward and the compositing backward and the all-zero tiles are counted."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops, render
from oracle import nerf_oracle as O
H = W = 200
imgs, poses, focal = fn.synthetic.make_dataset(n_images=20, H=H, W=W)
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
dev = torch.device('cuda')
torch.manual_seed(0); np.random.seed(0)
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
ktr = fn.run_nerf.create_nerf(args)[0]
tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
rays = [O.get_rays(H, W, K, poses[i]) for i in range(20)]
ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3).to(dev)
rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3).to(dev)
tgt_all = torch.as_tensor(imgs).reshape(-1, 3).to(dev)
gen = torch.Generator(device='cpu').manual_seed(1)
def probe(tag):
    sel = torch.randint(0, ro_all.shape[0], (4096,), generator=gen).to(dev)
    rays11 = ops.pack_rays(ro_all[sel], rd_all[sel], 2.0, 6.0)
    out, saved = render._forward_core(rays11, tr.net_c, tr.net_f, 64, 128, False, 1.0, True, None, None, None, None, save=True,
                                      packed_c=tr.pc, packed_f=tr.pf)
    loss2, g, g0 = ops.mse_leafmax(out['rgb_map'], out.get('rgb0'), tgt_all[sel])
    res = []
    for name, raw, z, gg, noise in (('coarse', saved['raw0'], saved['z0'], g0, saved['noise0']), ('fine', saved['raw1'], saved['z1'], g, saved.get('noise1'))):
        draw = ops.raw2outputs_bwd(raw, z, rays11, gg, noise, True)          # [n, S, 4]
        nz = (draw != 0).any(-1).reshape(-1)                                  # per point
        P = nz.numel()
        tiles = nz[:P // 64 * 64].view(-1, 64).any(-1)
        res.append('%s: %.1f %% of points, %.1f %% of 64-point tiles carry gradient' % (name, 100 * nz.float().mean().item(), 100 * tiles.float().mean().item()))
    print('%-12s loss %.5f | %s' % (tag, float(loss2[0]), ' | '.join(res)))
probe('init')
for ep in range(6):
    for it in range(200):
        sel = torch.randint(0, ro_all.shape[0], (4096,), generator=gen).to(dev)
        tr.step(ro_all[sel], rd_all[sel], tgt_all[sel])
    probe('iter %d' % ((ep + 1) * 200))
