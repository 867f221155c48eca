"""Stand-alone inference forward at the coarse (64 samples) and fine (192 samples) sizes of the bench batch, and at 3 x coarse:
is the coarse launch slower per point?   python tools/time_fwd_sizes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf as fn  # noqa: E402
from fastnerf import ops  # noqa: E402

torch.manual_seed(0)
dev = torch.device('cuda')
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
net = fn.run_nerf.create_nerf(args)[0]['network_fine']
pf, _ = net.packed(refresh=True)


def timeit(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for N, S in ((1024, 64), (2048, 64), (3072, 64), (3584, 64), (4000, 64), (4096, 64), (4200, 64), (4608, 64), (5120, 64), (6144, 64), (8192, 64), (4096, 192), (4096, 64), (4096, 63), (4096, 65)):
    ro = torch.randn(N, 3, device=dev) * 0.1
    rd = torch.randn(N, 3, device=dev)
    rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values
    raw = torch.empty(N, S, 4, device=dev)
    t = timeit(lambda: ops.mlp_fwd(rays11, z, net.flat, pf, raw=raw))
    print(f'N={N:6d} S={S:4d}: {t * 1e3:8.1f} us = {t * 1e6 / (N * S):.3f} ns/point, {N * S // 64} tiles', flush=True)
