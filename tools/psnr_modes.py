"""PSNR at equal iterations, split-bf16 default vs exact-fp32 math mode, at a BASELINE-like scale on one GPU:
the whole run_nerf.train() epoch loop (quadtree ray selection) on a synthetic multi-view scene, same seeds in both
modes, then held-out views rendered with render_path (perturb = 0).  north_star: PSNR within 0.1 dB.
usage: python tools/psnr_modes.py [H=W] [n_train_views] [n_epoch] [n_seeds] [first_seed]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_epoch = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n_test = 4
n_seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 1
seed0 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
imgs, poses, focal = fn.synthetic.make_dataset(n_images=n_train + n_test, H=H, W=W)
sel_test = np.arange(n_test) * ((n_train + n_test) // n_test)
sel_train = np.array([i for i in range(n_train + n_test) if i not in set(sel_test.tolist())])
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
res = {'bf16x3': [], 'fp32': []}
for seed in range(seed0, seed0 + n_seeds):
  for mode in ('bf16x3', 'fp32'):
    ops.set_math(mode)
    torch.manual_seed(seed); np.random.seed(seed)
    args = fn.run_nerf.make_args(N_importance=128, N_samples=64, perturb=1.0, white_bkgd=True, no_reload=True, N_rand=4096,
                                 n_epoch=n_epoch, init_level=2, subdivide_every=1, subdivide_thres=0.02, lrate=5e-4, lrate_decay=500)
    t0 = time.time()
    ktr, kte, trainer, mgr, hist = fn.run_nerf.train(imgs[sel_train], poses[sel_train], H, W, focal, args, log=lambda *a: None,
                                                     compat_rng=False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    iters = sum(h[1] for h in hist)
    with torch.no_grad():
        rgbs, _ = fn.render.render_path(torch.as_tensor(poses[sel_test]).cuda(), [H, W, focal], K, 32768, kte,
                                        gt_imgs=np.asarray(imgs[sel_test]))
    ps = fn.render.render_path.last_psnrs
    res[mode].append(float(np.mean(ps)))
    tp = [float(h[3]) for h in hist]
    print('seed %d %-7s %d iterations of 4096 rays in %.1f s; train psnr per epoch %s; held-out PSNR %.3f dB (%s)' % (
        seed, mode, iters, dt, ' '.join('%.2f' % p for p in tp), res[mode][-1], ' '.join('%.2f' % p for p in ps)))
a, b = np.array(res['bf16x3']), np.array(res['fp32'])
print('held-out PSNR over %d seeds: bf16x3 %.3f +- %.3f dB, fp32 %.3f +- %.3f dB, difference of the means %+.3f dB' % (
    n_seeds, a.mean(), a.std(), b.mean(), b.std(), a.mean() - b.mean()))
