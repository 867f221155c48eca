"""BASELINE configs[2] flow on one GPU: the whole epoch loop of run_nerf.train() with quadtree ray selection (per-epoch
ray generation from the leaf plans, fused steps feeding the on-device leaf-error table, tree adjustment) on a synthetic
multi-view scene.  Reports end-to-end rays/s per epoch (host quadtree work included).  DESIGN.md cites it."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import fastnerf as fn
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_images = int(sys.argv[2]) if len(sys.argv) > 2 else 20
imgs, poses, focal = fn.synthetic.make_dataset(n_images=n_images, H=H, W=W, device='cuda' if H >= 400 else None)
torch.manual_seed(0); np.random.seed(0)
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, N_rand=4096,
                             n_epoch=5, init_level=2, subdivide_every=1, subdivide_thres=0.02, lrate=5e-4, lrate_decay=500)
logs = []
t0 = time.time()
ktr, kte, trainer, mgr, hist = fn.run_nerf.train(imgs, poses, H, W, focal, args, log=logs.append, compat_rng=False)
torch.cuda.synchronize()
print('\n'.join(logs))
for (ep, it, mse, psnr, sec) in hist:
    print('epoch %d: %d steps of 4096 rays in %.2f s = %.0f rays/s end to end (psnr %.2f, leaves max %d)' % (
        ep, it, sec, it * 4096 / sec, psnr, mgr.max_leaves()))
print('total %.1f s' % (time.time() - t0))
