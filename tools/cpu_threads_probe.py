import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, numpy as np
from oracle import nerf_oracle as O
gen = torch.Generator().manual_seed(0)
sdc, sdf = O.init_nerf_params(gen), O.init_nerf_params(gen)
opt = O.Adam(list(sdc.values()) + list(sdf.values()), lr=5e-4)
c2w = O.pose_spherical(30.0, -30.0, 4.0)[:3, :4]
K = O.intrinsics(800, 800, 1111.111)
ro, rd = O.get_rays(800, 800, K, c2w)
def run(n, threads, reps):
    torch.set_num_threads(threads)
    sel = torch.randint(0, 640000, (n,), generator=gen)
    rb = O.make_ray_batch(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], 2.0, 6.0)
    tgt = torch.rand(n, 3, generator=gen)
    ts = []
    for it in range(reps):
        t_rand, u = torch.rand(n, 64, generator=gen), torch.rand(n, 128, generator=gen)
        t0 = time.time()
        O.train_step(sdc, sdf, opt, rb, tgt, 64, 128, True, t_rand=t_rand, u=u)
        ts.append(time.time() - t0)
    print(f'n={n} threads={threads}: ' + ' '.join(f'{t:.2f}' for t in ts) + f' s  -> {n/min(ts):.1f} rays/s', flush=True)
for n, th, reps in ((1024, 32, 2), (1024, 64, 2), (1024, 128, 2), (16, 256, 2), (64, 256, 1), (256, 128, 2), (256, 64, 2)):
    run(n, th, reps)
