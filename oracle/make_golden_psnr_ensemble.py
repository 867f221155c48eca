"""Records tests/golden/g22_psnr_cpu_ensemble.npz: the CPU oracle's PSNR after 200 iterations of the paired protocol of
oracle/psnr_protocol.py for initialisation seeds 0 .. N-1 (one free run each; ~2.5 minutes of 8 host cores per seed).  Run in the
build container (no GPU needed):   python -m oracle.make_golden_psnr_ensemble [N] [first_seed]
The file is rewritten after every seed, so a partial ensemble is usable; the GPU side is tests/test_gpu_train.py::
test_psnr_paired_with_the_cpu_ensemble_g22."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import psnr_protocol as P      # noqa: E402
import importlib                            # noqa: E402
synthetic = importlib.import_module('fast-learning-nerf_amd.synthetic')   # the analytic scene (pure torch; runs on CPU tensors)

OUT = os.path.join(ROOT, 'tests', 'golden', 'g22_psnr_cpu_ensemble.npz')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.set_num_threads(int(os.environ.get('G22_THREADS', '8')))
    data = P.inputs(lambda o, d: synthetic.render_rays(o, d, cutoff=0.0))
    done = {}
    if os.path.exists(OUT):
        z = np.load(OUT)
        done = {int(s): (float(a), float(b), float(c)) for s, a, b, c in zip(z['seeds'], z['train_psnr_db'], z['held_out_psnr_db'], z['first_loss'])}
    for seed in range(first, first + n):
        if seed in done:
            continue
        t0 = time.time()
        done[seed] = P.cpu_run(seed, data)
        seeds = sorted(done)
        np.savez(OUT, seeds=np.array(seeds), train_psnr_db=np.array([done[s][0] for s in seeds]),
                 held_out_psnr_db=np.array([done[s][1] for s in seeds]), first_loss=np.array([done[s][2] for s in seeds]),
                 protocol=np.array([P.ITERS, P.RAYS, P.HELD_OUT, P.WINDOW, P.N_SAMPLES, P.N_IMPORTANCE]),
                 input_digest=np.array([float(data['ro'].double().sum()), float(data['tgt'].double().sum()), float(data['u'].double().sum())]),
                 torch_version=np.array(torch.__version__))
        print('seed %d: train %.3f dB, held-out %.3f dB, first loss %.5f  (%.0f s)' % ((seed,) + done[seed] + (time.time() - t0,)), flush=True)


if __name__ == '__main__':
    main()
