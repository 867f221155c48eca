import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import fastnerf as fn
from oracle import nerf_oracle as O
gd = '/root/repo/tests/golden'
g = np.load(gd + '/g7_weights.npz')
weights = {k[2:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith('c.')}
flat = torch.cat([weights[n].reshape(-1) for n, _ in O.nerf_param_shapes()]).cuda()
gen = torch.Generator().manual_seed(77)
for (n, S) in ((9, 50), (16, 64)):
    ro = torch.randn(n, 3, generator=gen) * 0.5; rd = torch.randn(n, 3, generator=gen)
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values
    cot = torch.randn(n, S, 4, generator=gen)
    sd = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
    out = O.run_network(sd, pts, rb[:, 8:11])
    grads_ref = torch.autograd.grad((out * cot).sum(), list(sd.values()))
    pf, pb = fn.ops.mlp_pack(flat)
    P = n * S
    act = torch.empty(fn.ops.act_floats(P)).cuda()
    raw = fn.ops.mlp_fwd(rb.cuda(), z.cuda(), flat, pf, act=act)
    dact = torch.empty(P * fn.ops.DACT_FLOATS).cuda()
    partial = torch.empty(fn.ops.mlp_bwd_partial_floats()).cuda()
    grads = torch.full((fn.ops.NET_PARAMS,), float('nan')).cuda()
    fn.ops.mlp_bwd(cot.cuda(), act, flat, pb, dact, partial, grads)
    grads = grads.cpu(); off = 0
    print('P', P)
    for (name, shape), gr in zip(O.nerf_param_shapes(), grads_ref):
        k = gr.numel(); got = grads[off:off + k].view(shape); off += k
        print('  %-26s scale %.3e maxerr %.3e' % (name, gr.abs().max(), (got - gr).abs().max()))
    # masks vs activations
    actc = act.cpu()
    m64 = actc[P * 2528:].view(torch.int64)
    for l in (0, 7):
        h = actc[P * 64 + l * P * 256: P * 64 + (l + 1) * P * 256].view(P, 256)
        bad = 0; tot = 0
        for tile in range((P + 127) // 128):
            for wave in range(8):
                wm, wn = wave >> 2, wave & 3
                words = m64[(tile * 64 + l * 8 + wave) * 64:(tile * 64 + l * 8 + wave) * 64 + 64]
                for idx in (0, 17, 63):
                    nt, mt, r = idx // 32, (idx // 16) % 2, idx % 16
                    w = int(words[idx].item()) & ((1 << 64) - 1)
                    for lane in (0, 5, 40, 63):
                        m = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        nn = (wn * 2 + nt) * 32 + (lane & 31)
                        p = tile * 128 + m
                        if p < P:
                            tot += 1; bad += int(((w >> lane) & 1) != int(h[p, nn] > 0))
        print('  mask check layer', l, 'bad', bad, 'of', tot)
