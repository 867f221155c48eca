"""Harness: split-bf16 vs fp32-MFMA MLP kernels (accuracy + time of fwd / fwd+save / bwd).  GPU only."""
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
res = {}
def timeit(f, n=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mode in ('fp32', 'bf16x3'):
    ops.set_math(mode)
    pf, pb = net.packed(refresh=True)
    raw = torch.empty(N, S, 4, device=dev)
    act = torch.empty(ops.act_floats(P), device=dev)
    dact = torch.empty(ops.dact_floats(P), device=dev)
    partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev)
    grads = torch.empty(ops.NET_PARAMS, device=dev)
    t_inf = timeit(lambda: ops.mlp_fwd(rays11, z, net.flat, pf, raw=raw))
    t_sav = timeit(lambda: ops.mlp_fwd(rays11, z, net.flat, pf, act=act, raw=raw))
    t_bwd = timeit(lambda: ops.mlp_bwd(cot, act, net.flat, pb, dact, partial, grads))
    res[mode] = (raw.clone(), grads.clone())
    fl = P * 2 * 593408 / 1e9
    print('%-7s fwd %.3f ms (%.0f TF)  fwd+save %.3f ms (%.0f TF)  bwd %.3f ms (%.0f TF)' % (
        mode, t_inf, fl / t_inf, t_sav, fl / t_sav, t_bwd, 2 * fl / t_bwd))
    del act, dact
d = (res['fp32'][0] - res['bf16x3'][0]).abs()
g = (res['fp32'][1] - res['bf16x3'][1]).abs()
print('raw max diff %.3e (max |raw| %.3e)   grad max diff %.3e (max |grad| %.3e)' % (
    d.max().item(), res['fp32'][0].abs().max().item(), g.max().item(), res['fp32'][1].abs().max().item()))

off = 0
from oracle import nerf_oracle as O
for name, shape in O.nerf_param_shapes():
    k = 1
    for d_ in shape: k *= d_
    a = res['fp32'][1][off:off + k]; b = res['bf16x3'][1][off:off + k]
    print('  %-28s max|g| %.3e  max diff %.3e  rel %.2e' % (name, a.abs().max().item(), (a - b).abs().max().item(),
          (a - b).abs().max().item() / a.abs().max().item()))
    off += k
