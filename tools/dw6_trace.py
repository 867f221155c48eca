#!/usr/bin/env python3
"""Timing study (variant builds with -DDW6_TRACE=1): per-wave phase clocks of workgroup 0 of the 256 x 256 bf16x6 dW job."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf
from fastnerf import ops, _lib
ops.set_math('bf16x6')
dev = torch.device('cuda')
N, S1 = 4096, 192
P = N * S1
torch.manual_seed(0)
args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
net = fastnerf.run_nerf.create_nerf(args, device=dev)[0]['network_fine']
flat = net.flat
pf = net.packed(refresh=True)
ro = torch.randn(N, 3, device=dev) * 0.1 + torch.tensor([0., 0., 4.], device=dev); rd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
rays = ops.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S1, device=dev) * 4 + 2, -1).values
act = torch.empty(ops.act_floats(P), device=dev); dact = torch.empty(ops.dact_floats(P), device=dev)
partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev); grads = torch.empty(ops.NET_PARAMS, device=dev)
raw = torch.empty(N, S1, 4, device=dev); draw = torch.randn(N, S1, 4, device=dev) * 1e-4
for _ in range(2):
    ops.mlp_fwd(rays, z, flat, pf[0], act=act, raw=raw)
    ops.mlp_bwd(draw, act, flat, pf[1], dact, partial, grads)
torch.cuda.synchronize()
lib = _lib.lib() if hasattr(_lib, 'lib') else ctypes.CDLL(os.environ['FASTNERF_LIB'])
out = (ctypes.c_longlong * 48)()
f = lib.fastnerf_debug_dw6_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
print('rc', f(out))
a = np.array(list(out)).reshape(8, 6)
print('wave  wait-loads  multiply  split+Swrite  barrier  issue-loads   (clock64 ticks per k-step)')
for w in range(8):
    n = max(a[w, 5], 1)
    print('%4d  %10.0f %9.0f %11.0f %8.0f %10.0f  total %6.0f   k-steps %d' % (w, a[w,0]/n, a[w,1]/n, a[w,2]/n, a[w,3]/n, a[w,4]/n, a[w,:5].sum()/n, a[w,5]))
