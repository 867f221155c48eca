#!/usr/bin/env python3
"""Cross-evaluation behind profiles/r06_psnr_null.md ("explain or dissolve G23's +0.33 dB held-out offset"): is any part of
a GPU - CPU PSNR difference at 1000 iterations due to the RENDERERS (the HIP render() and the CPU oracle evaluate the same weights differently)
rather than to the two free training trajectories?

  python tools/psnr_cross_eval.py gpu [K]    ON THE GPU BOX.  (a) every final-weight file of the CPU null members under oracle/_study/weights/
                                            (long_m1_<seed>.npz, recorded by tools/record_null_members.py with the CPU's own held-out PSNR in
                                            oracle/_study/parts/long_m1_<seed>.json) is rendered with the HIP render(): CPU-trained weights, GPU
                                            evaluation.  (b) the first K (default 8) G23 seeds are trained on the GPU (bf16x6, the paired
                                            protocol) and their final weights + the GPU's own held-out PSNR go to gpurun_out/cross/.
  python tools/psnr_cross_eval.py cpu        IN THE BUILD CONTAINER: the GPU-trained weights of (b) evaluated by the CPU oracle.
Writes gpurun_out/cross/cross_gpu.json / cross_cpu.json (the report is tools/psnr_null_report.py)."""
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import psnr_protocol as P      # noqa: E402  (a study tool, not the product)
OUT = os.path.join(ROOT, 'gpurun_out', 'cross')


def main():
    mode = sys.argv[1]
    os.makedirs(OUT, exist_ok=True)
    if mode == 'gpu':
        import fastnerf as fn
        K_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        dev = torch.device('cuda')
        data = P.inputs(lambda o, d: fn.synthetic.render_rays(o, d, cutoff=0.0), iters=P.LONG_ITERS, batch_seed=3)
        dd = {k: v.to(dev) for k, v in data.items()}
        Kmat = np.array([[P.FOCAL, 0, 0.5 * P.W], [0, P.FOCAL, 0.5 * P.H], [0, 0, 1]])
        args = fn.run_nerf.make_args(N_importance=P.N_IMPORTANCE, N_samples=P.N_SAMPLES, perturb=1.0, white_bkgd=True, no_reload=True,
                                     lrate=5e-4, lrate_decay=500)
        fn.ops.set_math('bf16x6')
        fn.render.set_compact('0')

        def held_out(kte):
            with torch.no_grad():
                rgb = fn.render.render(P.H, P.W, Kmat, chunk=P.HELD_OUT, rays=(dd['ho_ro'], dd['ho_rd']), near=2.0, far=6.0, **kte)[0]
                return P.psnr(float(torch.mean((rgb - dd['ho_tgt']) ** 2)))
        res = {'cpu_weights_gpu_eval': {}, 'gpu_runs': {}}
        for f in sorted(glob.glob(os.path.join(ROOT, 'oracle', '_study', 'weights', 'long_m1_*.npz'))):
            seed = int(os.path.basename(f)[len('long_m1_'):-4])
            z = np.load(f)
            ktr, kte, _, _, _, _ = fn.run_nerf.create_nerf(args, device=dev)
            ktr['network_fn'].load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('c.')})
            ktr['network_fine'].load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('f.')})
            part = json.load(open(os.path.join(ROOT, 'oracle', '_study', 'parts', 'long_m1_%d.json' % seed)))
            g = held_out(kte)
            res['cpu_weights_gpu_eval'][seed] = {'cpu_eval_db': part['held'], 'gpu_eval_db': g, 'diff_db': g - part['held']}
            print('seed %d (CPU-trained weights): CPU evaluation %.4f dB, HIP render() %.4f dB, difference %+.5f' % (seed, part['held'], g, g - part['held']), flush=True)
        z23 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g23_psnr_cpu_long.npz'))
        seeds = [int(s) for s, h in zip(z23['seeds'], z23['held_out_psnr_db']) if h > 15.0][:K_seeds]
        for seed in seeds:
            ktr, kte, _, _, _, _ = fn.run_nerf.create_nerf(args, device=dev)
            sdc, sdf = P.init_weights(seed)
            ktr['network_fn'].load_state_dict(sdc)
            ktr['network_fine'].load_state_dict(sdf)
            tr = fn.run_nerf.Trainer(ktr, P.H, P.W, Kmat, 2.0, 6.0, lrate=5e-4, lrate_decay=500)
            tr.repack()
            for it in range(P.LONG_ITERS):
                tr.step(dd['ro'][it], dd['rd'][it], dd['tgt'][it], t_rand=dd['t_rand'][it], u=dd['u'][it])
            g = held_out(kte)
            np.savez(os.path.join(OUT, 'gpu_final_%d.npz' % seed),
                     **{'c.' + k: v.detach().cpu().numpy() for k, v in ktr['network_fn'].state_dict().items()},
                     **{'f.' + k: v.detach().cpu().numpy() for k, v in ktr['network_fine'].state_dict().items()})
            res['gpu_runs'][seed] = {'gpu_eval_db': g}
            print('seed %d (GPU-trained): HIP render() %.4f dB' % (seed, g), flush=True)
        json.dump(res, open(os.path.join(OUT, 'cross_gpu.json'), 'w'), indent=1)
    else:
        synthetic = __import__('importlib').import_module('fast-learning-nerf_amd.synthetic')
        torch.set_num_threads(int(os.environ.get('G22_THREADS', '2')))
        data = P.inputs(lambda o, d: synthetic.render_rays(o, d, cutoff=0.0), iters=P.LONG_ITERS, batch_seed=3)
        gj = json.load(open(os.path.join(OUT, 'cross_gpu.json')))
        res = {}
        for f in sorted(glob.glob(os.path.join(OUT, 'gpu_final_*.npz'))):
            seed = int(os.path.basename(f)[len('gpu_final_'):-4])
            z = np.load(f)
            sdc = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('c.')}
            sdf = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('f.')}
            c = P.held_out_psnr(sdc, sdf, data)
            g = gj['gpu_runs'][str(seed)]['gpu_eval_db']
            res[seed] = {'gpu_eval_db': g, 'cpu_eval_db': c, 'diff_db': g - c}
            print('seed %d (GPU-trained weights): HIP render() %.4f dB, CPU evaluation %.4f dB, difference %+.5f' % (seed, g, c, g - c), flush=True)
        json.dump(res, open(os.path.join(OUT, 'cross_cpu.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
