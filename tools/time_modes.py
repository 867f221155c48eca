"""Fine-pass launch times and step time (SURVEY 8(d) protocol, plain backward) per math mode."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fastnerf
from fastnerf import ops
import bench as B
dev = torch.device('cuda')
N, S1 = 4096, 192
modes = sys.argv[1:] or ['fp32', 'bf16x6', 'bf16x3']
K = np.array([[B.FOCAL, 0, 400.0], [0, B.FOCAL, 400.0], [0, 0, 1]])
poses = torch.stack([fastnerf.synthetic.pose_spherical(-180.0 + 3.6 * k, -30.0, 4.0)[:3, :4] for k in range(100)], 0).to(dev)
g = torch.Generator().manual_seed(1000)
pix = torch.stack([torch.randint(0, 100, (N,), generator=g), torch.randint(0, 800, (N,), generator=g), torch.randint(0, 800, (N,), generator=g)], 1).int()
ro, rd = ops.gen_rays_pixels(pix.to(dev), poses, K)
tgt = torch.rand(N, 3, generator=g).to(dev)
args = fastnerf.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, lrate=5e-4, lrate_decay=500)
fastnerf.render.set_compact(os.environ.get('TM_COMPACT', '0'))
for mode in modes:
    ops.set_math(mode)
    torch.manual_seed(0)
    tr = fastnerf.run_nerf.Trainer(fastnerf.run_nerf.create_nerf(args, device=dev)[0], 800, 800, K, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
    rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(N, S1, device=dev) * 4 + 2, -1).values
    P = N * S1
    act = torch.empty(ops.act_floats(P), device=dev); dact = torch.empty(ops.dact_floats(P), device=dev)
    partial = torch.empty(ops.mlp_bwd_partial_floats(), device=dev); gt = torch.empty(ops.NET_PARAMS, device=dev)
    raw = torch.empty(N, S1, 4, device=dev); draw = torch.randn(N, S1, 4, device=dev) * 1e-4
    ms_save = B.time_launch(lambda: ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], act=act, raw=raw), 5)
    ms_inf = B.time_launch(lambda: ops.mlp_fwd(rays11, z, tr.net_f.flat, tr.pf[0], raw=raw), 5)
    ms_bwd = B.time_launch(lambda: ops.mlp_bwd(draw, act, tr.net_f.flat, tr.pf[1], dact, partial, gt), 5)
    for _ in range(5): tr.step(ro, rd, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(ro, rd, tgt)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 50
    fl = P * B.FWD_FLOP_PER_POINT / 1e9
    print(f'{mode:7s} fwd-save {ms_save:6.2f} ms ({fl / ms_save:5.0f} TF)  fwd {ms_inf:6.2f} ms ({fl / ms_inf:5.0f} TF)  bwd {ms_bwd:6.2f} ms ({P * B.BWD_FLOP_PER_POINT / 1e9 / ms_bwd:5.0f} TF)  '
          f'step {ms:6.2f} ms = {N / ms:6.1f} k rays/s', flush=True)
