"""Run-to-run reproducibility: the same 150 optimisation steps (4096 rays, 64+128 samples, fixed seeds) twice in one
process and report whether parameters and losses are bit-identical; also scans every step's loss for non-finite values."""
import sys, hashlib, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf as fn
from fastnerf import ops
H = W = 200
imgs, poses, focal = fn.synthetic.make_dataset(n_images=8, H=H, W=W)
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
dev = torch.device('cuda')
from oracle import nerf_oracle as O
rays = [O.get_rays(H, W, K, poses[i]) for i in range(8)]
ro_all = torch.stack([r[0] for r in rays], 0).reshape(-1, 3).to(dev)
rd_all = torch.stack([r[1] for r in rays], 0).reshape(-1, 3).to(dev)
tgt_all = torch.as_tensor(imgs).reshape(-1, 3).to(dev)
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
out = []
for run in range(2):
    torch.manual_seed(0)
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
    ktr = fn.run_nerf.create_nerf(args)[0]
    tr = fn.run_nerf.Trainer(ktr, H, W, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    gen = torch.Generator(device='cpu').manual_seed(1)
    losses = []
    for it in range(n_steps):
        sel = torch.randint(0, ro_all.shape[0], (4096,), generator=gen).to(dev)
        t_rand = torch.rand(4096, 64, generator=gen).to(dev); u = torch.rand(4096, 128, generator=gen).to(dev)
        loss2, _ = tr.step(ro_all[sel], rd_all[sel], tgt_all[sel], t_rand=t_rand, u=u)
        losses.append(loss2.clone())
    torch.cuda.synchronize()
    L = torch.stack(losses).cpu()
    assert torch.isfinite(L).all(), 'non-finite loss'
    h = hashlib.sha256(tr.flat.cpu().numpy().tobytes()).hexdigest()[:16]
    out.append((h, L))
    print('run %d: %d steps, final loss %s, parameter hash %s' % (run, n_steps, L[-1].tolist(), h))
same_p = out[0][0] == out[1][0]
same_l = torch.equal(out[0][1], out[1][1])
print('parameters bit-identical across runs:', same_p, '  per-step losses bit-identical:', same_l)
if not same_l:
    d = (out[0][1] - out[1][1]).abs()
    print('first differing step', int((d.sum(1) > 0).nonzero()[0]), 'max |dloss|', float(d.max()))
