import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from oracle import nerf_oracle as O
gd = '/root/repo/tests/golden'
wts = np.load(gd + '/g7_weights.npz')
g8 = np.load(gd + '/g8_train_step.npz')
sd = {k[2:]: torch.from_numpy(wts[k]).clone().requires_grad_(True) for k in wts.files if k.startswith('c.')}
flat = torch.cat([sd[n].detach().reshape(-1) for n, _ in O.nerf_param_shapes()]).cuda()
pf, pb = fn.ops.mlp_pack(flat)
ro, rd, tgt = (torch.from_numpy(g8[k]) for k in ('ro', 'rd', 'target'))
rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
z = O.coarse_z(rb[:, 6:7], rb[:, 7:8], 64, False, torch.from_numpy(g8['t_rand']))
n, S = z.shape; P = n * S
# oracle with per-layer pre-activation capture
pts = (rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]).reshape(-1, 3)
emb = torch.cat([O.posenc(pts, 10), O.posenc(rb[:, None, 8:11].expand(n, S, 3).reshape(-1, 3), 4)], -1)
F = torch.nn.functional
pe, views = emb[:, :63], emb[:, 63:]
h = pe; pre = []; hs = []
for i in range(8):
    y = F.linear(h, sd[f'pts_linears.{i}.weight'], sd[f'pts_linears.{i}.bias']); y.retain_grad(); pre.append(y)
    h = torch.relu(y); hs.append(h)
    if i == 4: h = torch.cat([pe, h], -1)
alpha = F.linear(h, sd['alpha_linear.weight'], sd['alpha_linear.bias'])
feat = F.linear(h, sd['feature_linear.weight'], sd['feature_linear.bias']); feat.retain_grad()
yv = F.linear(torch.cat([feat, views], -1), sd['views_linears.0.weight'], sd['views_linears.0.bias']); yv.retain_grad()
rgb = F.linear(torch.relu(yv), sd['rgb_linear.weight'], sd['rgb_linear.bias'])
raw = torch.cat([rgb, alpha], -1).reshape(n, S, 4)
rgbm = O.raw2outputs(raw, z, rb[:, 3:6], None, True)[0]
loss = O.img2mse(rgbm, tgt); loss.backward()
# gpu
act = torch.empty(P * fn.ops.ACT_FLOATS).cuda()
rawg = fn.ops.mlp_fwd(rb.cuda(), z.cuda(), flat, pf, act=act)
print('raw err', (rawg.cpu() - raw.detach()).abs().max().item())
g_rgb = (2 * (rgbm.detach() - tgt) / (3 * n)).cuda()
draw = fn.ops.raw2outputs_bwd(rawg, z.cuda(), rb.cuda(), g_rgb, None, True)
dact = torch.empty(P * fn.ops.DACT_FLOATS).cuda()
partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
grads = torch.empty(fn.ops.NET_PARAMS).cuda()
fn.ops.mlp_bwd(draw, act, flat, pb, dact, partial, grads)
dact = dact.cpu(); actc = act.cpu()
for l in range(7, -1, -1):
    got = dact[l * P * 256:(l + 1) * P * 256].view(P, 256)
    ref = pre[l].grad
    err = (got - ref).abs()
    hg = actc[P * 64 + l * P * 256: P * 64 + (l + 1) * P * 256].view(P, 256)
    flips = ((hg > 0) != (hs[l].detach() > 0)).sum().item()
    big = (err > 1e-3 * ref.abs().max()).sum().item()
    print('dY%d scale %.3e maxerr %.3e rel %.2e  n(err>1e-3 max) %d  mask flips %d  h maxerr %.2e' % (l, ref.abs().max(), err.max(), err.max() / ref.abs().max(), big, flips, (hg - hs[l].detach()).abs().max()))
got = dact[8 * P * 256: 9 * P * 256].view(P, 256); print('dfeat rel', ((got - feat.grad).abs().max() / feat.grad.abs().max()).item())
got = dact[9 * P * 256: 9 * P * 256 + P * 128].view(P, 128); print('dYv rel', ((got - yv.grad).abs().max() / yv.grad.abs().max()).item())
