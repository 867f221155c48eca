"""Where a wave of the bf16x6 forward spends its cycles (library built with -DX6_TIMING, selected through FASTNERF_LIB): fine-pass launch of
786 432 points, saving and not saving.  Prints cycles per layer iteration (six plain 256 x 256 layers per tile are instrumented)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fastnerf
from fastnerf import ops, _lib
import bench as B
dev = torch.device('cuda')
N, S1 = 4096, 192
ops.set_math('bf16x6')
args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
K = np.array([[B.FOCAL, 0, 400.0], [0, B.FOCAL, 400.0], [0, 0, 1]])
torch.manual_seed(0)
tr = fastnerf.run_nerf.Trainer(fastnerf.run_nerf.create_nerf(args, device=dev)[0], 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
poses = torch.stack([fastnerf.synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
g = torch.Generator().manual_seed(1000)
pix = torch.stack([torch.randint(0, 100, (N,), generator=g), torch.randint(0, 800, (N,), generator=g), torch.randint(0, 800, (N,), generator=g)], 1).int()
ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S1, device=dev) * 4 + 2, -1).values
P = N * S1
act = torch.empty(ops.act_floats(P), device=dev)
raw = torch.empty(N, S1, 4, device=dev)
fn = _lib.lib().fastnerf_dbg_x6_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
for name, call in (('saving forward', lambda: ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], act=act, raw=raw)),
                   ('forward', lambda: ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], raw=raw))):
    for _ in range(2): call()
    torch.cuda.synchronize(); fn(None, 1)
    ms = B.time_launch(call, 3)
    torch.cuda.synchronize(); fn(None, 1)
    call(); torch.cuda.synchronize(); fn(out, 1)
    k, b1, e, b2, nl, tot, pro, ng = [int(v) for v in out]
    tiles = nl / 6 / 1.0   # wave-layer-iterations / 6 = wave-tiles
    print(f'{name}: {ms:.2f} ms/launch; per wave and plain layer: k-loop {k / nl:.0f} (768 MFMAs = 12288 pipe cycles; prologue {pro / ng:.0f} per gemm call)  '
          f'barrier-1 {b1 / nl:.0f}  epilogue {e / nl:.0f}  barrier-2 {b2 / nl:.0f}   whole tile {tot / tiles:.0f} cycles per wave '
          f'({tot / tiles / 9.06:.0f} per layer equivalent)', flush=True)
