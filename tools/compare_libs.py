#!/usr/bin/env python3
"""Gradients of one backward pass from two builds of the library (product vs a variants/<name>.so), same inputs, same math mode:
   python tools/compare_libs.py <mode> <variant.so> [n S]
Each build runs in its own process (the library is loaded once per process)."""
import os, subprocess, sys
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == '--worker':
    mode, n, S, out = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import fastnerf
    from fastnerf import ops
    ops.set_math(mode)
    gen = torch.Generator().manual_seed(5)
    args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
    torch.manual_seed(0)
    net = fastnerf.run_nerf.create_nerf(args, device=torch.device('cuda'))[0]['network_fine']
    flat = net.flat
    pf, pb = ops.mlp_pack(flat)
    ro = (torch.randn(n, 3, generator=gen) * 0.5).cuda(); rd = torch.randn(n, 3, generator=gen).cuda()
    rb = ops.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, -1).values.cuda()
    cot = torch.randn(n, S, 4, generator=gen).cuda()
    P = n * S
    act = torch.empty(ops.act_floats(P)).cuda()
    ops.mlp_fwd(rb, z, flat, pf, act=act)
    dact = torch.empty(ops.dact_floats(P)).cuda(); partial = torch.empty(ops.mlp_bwd_partial_floats()).cuda()
    g = torch.full((ops.NET_PARAMS,), float('nan')).cuda()
    ops.mlp_bwd(cot, act, flat, pb, dact, partial, g)
    np.save(out, g.cpu().numpy())
    sys.exit(0)

mode, var = sys.argv[1], sys.argv[2]
n, S = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1367, 193)
res = []
for lib in (None, var):
    env = dict(os.environ)
    if lib: env['FASTNERF_LIB'] = os.path.abspath(lib)
    else: env.pop('FASTNERF_LIB', None)
    out = '/tmp/cmp_%s.npy' % ('var' if lib else 'base')
    subprocess.check_call([sys.executable, __file__, '--worker', mode, str(n), str(S), out], env=env)
    res.append(np.load(out).astype(np.float64))
a, b = res
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import nerf_oracle as O
off = 0
print('mode %s  P = %d   product vs %s' % (mode, n * S, var))
for name, shape in O.nerf_param_shapes():
    k = int(np.prod(shape))
    x, y = a[off:off + k], b[off:off + k]
    print('  %-28s max|diff| %.3e  of max %.3e   rel L2 %.3e' % (name, np.abs(x - y).max(), np.abs(y).max(), np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30)))
    off += k
