import sys, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops
N, S = 4096, 192
torch.manual_seed(0)
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
net = ktr['network_fine']; pf, pb = net.packed(); dev = torch.device('cuda')
ro = torch.randn(N, 3, device=dev) * 0.1; rd = torch.randn(N, 3, device=dev)
rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values
P = N * S
act = torch.empty(ops.act_floats(P), device=dev); raw = torch.empty(N, S, 4, device=dev)
dact = torch.empty(ops.dact_floats(P), device=dev); partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
grads = torch.empty(ops.NET_PARAMS, device=dev); draw = torch.randn(N, S, 4, device=dev) * 1e-3
ops.mlp_fwd(rays11, z, net.flat, pf, act=act, raw=raw)
for _ in range(2): ops.mlp_bwd(draw, act, net.flat, pb, dact, partial, grads)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.mlp_bwd(draw, act, net.flat, pb, dact, partial, grads)
e1.record(); torch.cuda.synchronize()
print('TIME mlp_bwd (dx + all dW) ms %.3f' % (e0.elapsed_time(e1) / 5))
