import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from oracle import nerf_oracle as O
gd = '/root/repo/tests/golden'
g = np.load(gd + '/g6_sample_pdf.npz')
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for name in ('rand', 'flat', 'spike'):
    det = fn.ops.sample_pdf(G(g['bins']), G(g['w_' + name]), 128, det=True).cpu().numpy()
    us = fn.ops.sample_pdf(G(g['bins']), G(g['w_' + name]), 128, u=G(g['u'])).cpu().numpy()
    print(name, 'det err', np.abs(det - g[name + '_det']).max(), 'u err', np.abs(us - g[name + '_u']).max())
sys.path.insert(0, '/root/repo/tests')
import test_gpu_render as tr
K = np.load(gd + '/g1_get_rays.npz')['K']
g7 = np.load(gd + '/g7_render.npz')
ktr, kte, gv, opt = tr.build(fn, gd)
rays = torch.stack([torch.from_numpy(g7['ro']), torch.from_numpy(g7['rd'])], 0).cuda()
with torch.no_grad():
    for tag, kw, extra in (('a', kte, {}), ('b', ktr, {'pytest': True}), ('c', kte, {'N_samples': 32, 'N_importance': 0})):
        kk = dict(kw); kk.update(extra)
        rgb, disp, acc, ex = fn.render.render(800, 800, K, chunk=32768, rays=rays, retraw=True, near=2.0, far=6.0, **kk)
        res = {'rgb': rgb, 'disp': disp, 'acc': acc}; res.update(ex)
        for k in ('rgb', 'acc', 'rgb0', 'acc0', 'raw', 'z_std', 'disp', 'disp0'):
            if f'{tag}.{k}' in g7.files:
                ref = g7[f'{tag}.{k}']; print(tag, k, 'maxerr', np.abs(res[k].cpu().numpy() - ref).max(), 'scale', np.abs(ref).max())
g8 = np.load(gd + '/g8_train_step.npz')
T = fn.run_nerf.Trainer(ktr, 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
ro, rd, tgt = (torch.from_numpy(g8[k]).cuda() for k in ('ro', 'rd', 'target'))
loss2, out = T.forward_backward(ro, rd, tgt, t_rand=torch.from_numpy(g8['t_rand']).cuda(), u=torch.from_numpy(g8['u']).cuda())
print('loss', loss2.tolist(), float(g8['loss']), float(g8['loss0']))
grad = T.grad.cpu()
names = ['c.' + n for n, _ in O.nerf_param_shapes()] + ['f.' + n for n, _ in O.nerf_param_shapes()]
shapes = [s for _, s in O.nerf_param_shapes()] * 2
off = 0
for n, shp in zip(names, shapes):
    ref = g8['grad.' + n]; k = ref.size
    got = grad[off:off + k].view(shp).numpy(); off += k
    print('%-28s scale %.3e maxerr %.3e rel %.2e' % (n, np.abs(ref).max(), np.abs(got - ref).max(), np.abs(got - ref).max() / np.abs(ref).max()))
# compare against the oracle run with the GPU's own z (isolates kernel error from input sensitivity)
wts = np.load(gd + '/g7_weights.npz')
sdc = {k[2:]: torch.from_numpy(wts[k]).clone() for k in wts.files if k.startswith('c.')}
sdf = {k[2:]: torch.from_numpy(wts[k]).clone() for k in wts.files if k.startswith('f.')}
z1 = out['z_vals'].cpu()
rb = O.make_ray_batch(ro.cpu(), rd.cpu(), 2.0, 6.0)
for v in sdf.values(): v.requires_grad_(True)
pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z1[..., None]
raw = O.run_network(sdf, pts, rb[:, 8:11])
rgbm = O.raw2outputs(raw, z1, rb[:, 3:6], None, True)[0]
l = O.img2mse(rgbm, tgt.cpu())
gr = torch.autograd.grad(l, list(sdf.values()))
off = fn.ops.NET_PARAMS
print('--- fine net vs oracle evaluated at the GPU z_vals')
for (n, shp), gg in zip(O.nerf_param_shapes(), gr):
    k = gg.numel(); got = grad[off:off + k].view(shp); off += k
    print('%-28s scale %.3e maxerr %.3e rel %.2e' % (n, gg.abs().max(), (got - gg).abs().max(), (got - gg).abs().max() / gg.abs().max()))
print('rgb err vs oracle@gpu-z', (out['rgb_map'].cpu() - rgbm.detach()).abs().max().item(), 'raw err', (out['raw'].cpu() - raw.detach()).abs().max().item())
