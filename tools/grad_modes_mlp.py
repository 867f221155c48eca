"""MLP-only logit / gradient check with TRAINED weights against an fp64 autograd reference: same points, same cotangent, every math
mode.  Reports per parameter tensor the relative L2 error of each mode vs fp64 and the ratio of sums (bias check)."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    sel = torch.randint(0, ro_all.shape[0], (4096,), generator=gen).to(dev)
    tr.step(ro_all[sel], rd_all[sel], tgt_all[sel])
net = ktr['network_fine']
n, S = 96, 128     # 12288 points: fp64 autograd on the CPU finishes in seconds
sel = torch.randint(0, ro_all.shape[0], (n,), generator=gen).to(dev)
rays11 = ops.pack_rays(ro_all[sel], rd_all[sel], 2.0, 6.0)
z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.to(dev)
cot = (torch.randn(n, S, 4, generator=gen) * 1e-3).to(dev)
P = n * S
res = {}
MODES = ('fp32', 'bf16x6', 'f16x3', 'bf16x3')
for mode in MODES:
    ops.set_math(mode)
    pf, pb = net.packed(refresh=True)
    act = torch.empty(ops.act_floats(P), device=dev)
    raw = ops.mlp_fwd(rays11, z, net.flat, pf, act=act)
    dact = torch.empty(ops.dact_floats(P), device=dev)
    partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
    grads = torch.zeros(ops.NET_PARAMS, device=dev)
    ops.mlp_bwd(cot, act, net.flat, pb, dact, partial, grads)
    torch.cuda.synchronize()
    res[mode] = (raw.cpu().double(), grads.cpu().double())
# fp64 reference
sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
r11 = rays11.cpu().double()
pts = r11[:, None, 0:3] + r11[:, None, 3:6] * z.cpu().double()[:, :, None]
vd = r11[:, 8:11]
x = torch.cat([O.posenc(pts.reshape(-1, 3), 10), O.posenc(vd[:, None].expand(n, S, 3).reshape(-1, 3), 4)], -1)
raw64 = O.nerf_forward(sd, x).reshape(n, S, 4)
(raw64 * cot.cpu().double()).sum().backward()
names = [nm for nm, _ in O.nerf_param_shapes()]
g64 = torch.cat([sd[nm].grad.reshape(-1) for nm in names])
for mode in MODES:
    raw, g = res[mode]
    e = (raw - raw64.detach()).abs()
    print('%-7s raw: max abs err %.3e  mean abs err %.3e  rms %.3e (max |raw| %.2f)   grad: ||g - g64|| / ||g64|| = %.3e' % (
        mode, e.max().item(), e.mean().item(), e.pow(2).mean().sqrt().item(), raw64.abs().max().item(), ((g - g64).norm() / g64.norm()).item()))
off = 0
for nm, s in O.nerf_param_shapes():
    k = int(np.prod(s))
    r = g64[off:off + k]
    line = '%-28s' % nm
    for mode in MODES:
        a = res[mode][1][off:off + k]
        line += '  %s rel %.2e sum-ratio %.6f' % (mode, ((a - r).norm() / r.norm()).item(), (a.sum() / r.sum()).item())
    print(line)
    off += k
