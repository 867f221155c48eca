"""dX and dW timed separately (rocprof-free): the backward launcher with FASTNERF_DX_WGS / FASTNERF_DW_WGS overrides
(-DBF_EXPERIMENT builds): how do the two halves of the backward scale with the number of CUs they get?"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops
torch.manual_seed(0)
dev = torch.device('cuda')
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
net = fn.run_nerf.create_nerf(args)[0]['network_fine']
N, S = 4096, 192
P = N * S
ro = torch.randn(N, 3, device=dev) * 0.1; rd = torch.randn(N, 3, device=dev)
rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values
cot = torch.randn(N, S, 4, device=dev)
pf, pb = net.packed(refresh=True)
act = torch.empty(ops.act_floats(P), device=dev)
ops.mlp_fwd(rays11, z, net.flat, pf, act=act)
dact = torch.empty(ops.dact_floats(P), device=dev)
partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
grads = torch.empty(ops.NET_PARAMS, device=dev)
def t(n=5):
    for _ in range(2): ops.mlp_bwd(cot, act, net.flat, pb, dact, partial, grads)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.mlp_bwd(cot, act, net.flat, pb, dact, partial, grads)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('dx_wgs %s dw_wgs %s : bwd %.3f ms' % (os.environ.get('FASTNERF_DX_WGS', '512'), os.environ.get('FASTNERF_DW_WGS', '256'), t()))
