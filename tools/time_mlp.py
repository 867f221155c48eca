import sys, os, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops
torch.manual_seed(0)
dev = torch.device('cuda')
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
net = ktr['network_fine']
N, S = 4096, 192
P = N * S
ro = torch.randn(N, 3, device=dev) * 0.1; rd = torch.randn(N, 3, device=dev)
rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values
cot = torch.randn(N, S, 4, device=dev)
def timeit(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pf, pb = net.packed(refresh=True)
raw = torch.empty(N, S, 4, device=dev)
act = torch.empty(ops.act_floats(P), device=dev)
dact = torch.empty(ops.dact_floats(P), device=dev)
partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
grads = torch.empty(ops.NET_PARAMS, device=dev)
t_inf = timeit(lambda: ops.mlp_fwd(rays11, z, net.flat, pf, raw=raw))
t_sav = timeit(lambda: ops.mlp_fwd(rays11, z, net.flat, pf, act=act, raw=raw))
t_bwd = timeit(lambda: ops.mlp_bwd(cot, act, net.flat, pb, dact, partial, grads))
print('%s fwd %.3f  fwd+save %.3f  bwd %.3f ms' % (os.environ.get('FASTNERF_CFLAGS', ''), t_inf, t_sav, t_bwd))
