"""Gradient agreement of the two math modes at bench scale: the SAME weights (after a short training so that they are
not at the init distribution), the same 4096-ray batch and jitter; per parameter tensor ||g_bf16x3 - g_fp32|| / ||g_fp32||."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops
from oracle import nerf_oracle as O
H = W = 200
imgs, poses, focal = fn.synthetic.make_dataset(n_images=8, H=H, W=W)
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
dev = torch.device('cuda')
ops.set_math('fp32')
torch.manual_seed(0)
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
ktr = fn.run_nerf.create_nerf(args)[0]
tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
rays = [O.get_rays(H, W, K, poses[i]) for i in range(8)]
ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3).to(dev)
rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3).to(dev)
tgt_all = torch.as_tensor(imgs).reshape(-1, 3).to(dev)
gen = torch.Generator(device='cpu').manual_seed(1)
n_pre = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(n_pre):
    sel = torch.randint(0, ro_all.shape[0], (4096,), generator=gen).to(dev)
    tr.step(ro_all[sel], rd_all[sel], tgt_all[sel])
sel = torch.randint(0, ro_all.shape[0], (4096,), generator=gen).to(dev)
t_rand = torch.rand(4096, 64, generator=gen).to(dev); u = torch.rand(4096, 128, generator=gen).to(dev)
grads = {}
for mode in ('fp32', 'bf16x3', 'fp32'):
    ops.set_math(mode)
    tr.repack()
    loss2, out = tr.forward_backward(ro_all[sel], rd_all[sel], tgt_all[sel], t_rand=t_rand, u=u)
    torch.cuda.synchronize()
    grads.setdefault(mode, []).append((tr.grad.clone(), loss2.clone(), out['rgb_map'].clone()))
g32, l32, rgb32 = grads['fp32'][0]
g32b = grads['fp32'][1][0]
gbf, lbf, rgbbf = grads['bf16x3'][0]
print('loss fp32 %s  bf16x3 %s   max |rgb diff| %.3e' % (l32.tolist(), lbf.tolist(), (rgb32 - rgbbf).abs().max().item()))
print('fp32 run-to-run: ||dg||/||g|| = %.3e' % ((g32 - g32b).norm() / g32.norm()).item())
print('whole gradient: ||g_bf - g_32|| / ||g_32|| = %.3e   cosine %.8f' % (((gbf - g32).norm() / g32.norm()).item(),
      (torch.dot(gbf, g32) / (gbf.norm() * g32.norm())).item()))
off = 0
names = [n for n, _ in O.nerf_param_shapes()]
shapes = [s for _, s in O.nerf_param_shapes()]
for net in ('coarse', 'fine'):
    for n, s in zip(names, shapes):
        k = int(np.prod(s))
        a, b = gbf[off:off + k], g32[off:off + k]
        print('%-6s %-28s rel diff %.3e   max|g| %.3e  max|dg| %.3e  sum ratio %.6f' % (net, n, ((a - b).norm() / (b.norm() + 1e-30)).item(),
              b.abs().max().item(), (a - b).abs().max().item(), (a.sum() / b.sum()).item()))
        off += k
