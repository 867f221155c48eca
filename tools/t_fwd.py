"""Scratch harness for csrc/experimental/mlp_t.hip: builds it into tools/libt.so (git-ignored) on the GPU box and
compares / times it against the shipped split-bf16 forward."""
import ctypes, os, subprocess, sys, torch
sys.path.insert(0, '/root/repo')
ROOT = '/root/repo'
so = os.path.join(ROOT, 'tools', 'libt.so')
csrc = os.path.join(ROOT, 'fast-learning-nerf_amd', 'csrc')
extra = os.environ.get('T_CFLAGS', '').split()
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize',
                       '-fPIC', '-shared', '-Wno-unused-result', '-I', csrc, '-I', os.path.join(ROOT, 'include')] + extra +
                      [os.path.join(csrc, 'experimental', 'mlp_t.hip'), os.path.join(csrc, 'rays.hip'), '-o', so])
import fastnerf as fn
from fastnerf import ops
from fastnerf._lib import ptr, stream
L = ctypes.CDLL(so)
L.fastnerf_mlp_t_floats.restype = ctypes.c_int64
L.fastnerf_mlp_t_floats.argtypes = [ctypes.c_int, ctypes.c_int]
VP = ctypes.c_void_p
L.fastnerf_mlp_t_pack.argtypes = [ctypes.c_int, VP, VP, VP]
L.fastnerf_mlp_t_fwd.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, VP, VP, VP, VP, VP, VP]
torch.manual_seed(0)
dev = torch.device('cuda')
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
ktr, _, _, _, _, _ = fn.run_nerf.create_nerf(args)
net = ktr['network_fine']
with torch.no_grad():
    net.flat.mul_(1.0 + 0.3 * torch.rand_like(net.flat))
ops.set_math('bf16x3')
pf, pb = net.packed(refresh=True)
pt = torch.empty(L.fastnerf_mlp_t_floats(0, 1), device=dev)
assert L.fastnerf_mlp_t_pack(0, ptr(net.flat), ptr(pt), stream()) == 0
def tfwd(rays11, z, raw):
    n, S = z.shape
    assert L.fastnerf_mlp_t_fwd(0, n, S, ptr(rays11), ptr(z), ptr(net.flat), ptr(pt), ptr(raw), stream()) == 0
    return raw
for (N, S) in ((3, 5), (64, 64), (4096, 192)):
    ro = torch.randn(N, 3, device=dev) * 0.1; rd = torch.randn(N, 3, device=dev)
    rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values
    r0 = ops.mlp_fwd(rays11, z, net.flat, pf)
    r1 = tfwd(rays11, z, torch.empty(N, S, 4, device=dev))
    torch.cuda.synchronize()
    d = (r0 - r1).abs()
    print('N %d S %d  max|raw| %.3e  max diff %.3e  mean diff %.3e' % (N, S, r0.abs().max().item(), d.max().item(), d.mean().item()))
raw = torch.empty(N, S, 4, device=dev)
for name, f in (('bf16x3', lambda: ops.mlp_fwd(rays11, z, net.flat, pf, raw=raw)), ('tchain', lambda: tfwd(rays11, z, raw))):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print('%s ms %.3f' % (name, e0.elapsed_time(e1) / 5))
