"""Records tests/golden/g22_psnr_cpu_ensemble.npz (and g23_psnr_cpu_long.npz): the CPU oracle's PSNR after 200 (1000) iterations of the
paired protocol of oracle/psnr_protocol.py, one free run per initialisation seed.  No GPU needed; the oracle itself is the generator (its
step is pinned to the reference by G8), so these files are a yardstick recorded from the oracle, not reference-held vectors.

  python -m oracle.make_golden_psnr_ensemble [N] [first_seed]                   seeds first .. first+N-1 into the golden file (rewritten after
                                                                               every seed: a partial ensemble is usable)
  python -m oracle.make_golden_psnr_ensemble --parts DIR --seeds A B [--long]   one JSON per seed under DIR (parallel workers, any host:
                                                                               the build container or the GPU box's host cores)
  python -m oracle.make_golden_psnr_ensemble --merge DIR [--long]               fold DIR/*.json into the golden file
  python -m oracle.make_golden_psnr_ensemble --check K                          regenerate K recorded seeds with the recorded thread count and
                                                                               compare at 1e-3 dB (oracle/check_goldens.sh)
G22_THREADS (default 8) = torch CPU threads per run; recorded per seed (a free run is chaotic: another thread count regroups the GEMMs' sums
and gives another member of the same distribution, so --check replays with the recorded count).
~2.5 minutes of 8 cores (11 minutes of 1 core) per 200-iteration seed.  GPU side: tests/test_gpu_train.py::test_psnr_paired_with_the_cpu_ensemble_g22."""
import argparse
import glob
import json
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

GOLD = os.environ.get('FASTNERF_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')


def out_path(long, member=0):
    if member:      # the NULL members (CPU' = the CPU oracle from weights x (1 + 1e-6 N(0, 1))): G24 (200 iterations) / G25 (1000)
        return os.path.join(GOLD, ('g25_psnr_cpu_null_long_m%d.npz' if long else 'g24_psnr_cpu_null_m%d.npz') % member)
    return os.path.join(GOLD, 'g23_psnr_cpu_long.npz' if long else 'g22_psnr_cpu_ensemble.npz')


def part_name(long, member, seed):
    return ('long_' if long else 'seed_') + ('m%d_' % member if member else '') + '%d.json' % seed


def make_inputs(long):
    scene = lambda o, d: synthetic.render_rays(o, d, cutoff=0.0)   # noqa: E731
    return P.inputs(scene, iters=P.LONG_ITERS, batch_seed=3) if long else P.inputs(scene)


def load(path):
    if not os.path.exists(path):
        return {}
    z = np.load(path)
    th = z['threads'] if 'threads' in z.files else np.full(len(z['seeds']), 8)
    return {int(s): (float(a), float(b), float(c), int(t)) for s, a, b, c, t in
            zip(z['seeds'], z['train_psnr_db'], z['held_out_psnr_db'], z['first_loss'], th)}


def save(path, done, data, long):
    seeds = sorted(done)
    iters = P.LONG_ITERS if long else P.ITERS
    np.savez(path, seeds=np.array(seeds), train_psnr_db=np.array([done[s][0] for s in seeds]),
             held_out_psnr_db=np.array([done[s][1] for s in seeds]), first_loss=np.array([done[s][2] for s in seeds]),
             threads=np.array([done[s][3] for s in seeds]),
             protocol=np.array([iters, P.RAYS, P.HELD_OUT, P.WINDOW, P.N_SAMPLES, P.N_IMPORTANCE]),
             input_digest=np.array([float(data['ro'].double().sum()), float(data['tgt'].double().sum()), float(data['u'].double().sum())]),
             torch_version=np.array(torch.__version__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('n', nargs='?', type=int, default=40)
    ap.add_argument('first', nargs='?', type=int, default=0)
    ap.add_argument('--parts')
    ap.add_argument('--seeds', nargs=2, type=int)
    ap.add_argument('--seed-list', type=int, nargs='*')
    ap.add_argument('--merge')
    ap.add_argument('--check', type=int)
    ap.add_argument('--long', action='store_true')
    ap.add_argument('--member', type=int, default=0, help='0 = the recorded ensemble; k > 0 = the k-th jittered NULL member (G24 / G25)')
    ap.add_argument('--weights', help='directory for the final weights of every run (cross-evaluation study; not a golden)')
    a = ap.parse_args()
    threads = int(os.environ.get('G22_THREADS', '8'))
    torch.set_num_threads(threads)
    path = out_path(a.long, a.member)
    data = make_inputs(a.long)
    if a.merge:
        done = load(path)
        for f in sorted(glob.glob(os.path.join(a.merge, ('long_' if a.long else 'seed_') + ('m%d_' % a.member if a.member else '') + '*.json'))):
            r = json.load(open(f))
            if int(r.get('member', 0)) != a.member:
                continue
            done.setdefault(int(r['seed']), (r['train'], r['held'], r['first'], int(r['threads'])))
        save(path, done, data, a.long)
        print('%d seeds in %s' % (len(done), path))
        return
    if a.check:
        done = load(os.path.join(ROOT, 'tests', 'golden', os.path.basename(path)))
        own = [s for s in sorted(done) if done[s][3] == threads]
        pick = own[:: max(1, len(own) // a.check)][:a.check]
        bad = 0
        for s in pick:
            t0 = time.time()
            r = P.cpu_run(s, data, member=a.member)
            same = abs(r[0] - done[s][0]) < 1e-3 and abs(r[1] - done[s][1]) < 1e-3
            print('seed %d (%d threads): train %.4f vs %.4f, held-out %.4f vs %.4f  %s  (%.0f s)' % (
                s, threads, r[0], done[s][0], r[1], done[s][1], 'ok' if same else 'DIFFERS', time.time() - t0), flush=True)
            bad += not same
        sys.exit(1 if bad or not pick else 0)
    if a.parts:
        os.makedirs(a.parts, exist_ok=True)
        todo = a.seed_list if a.seed_list else range(a.seeds[0], a.seeds[1])
        for seed in todo:
            f = os.path.join(a.parts, part_name(a.long, a.member, seed))
            if os.path.exists(f):
                continue
            t0 = time.time()
            r = P.cpu_run(seed, data, member=a.member, return_weights=bool(a.weights))
            if a.weights:
                os.makedirs(a.weights, exist_ok=True)
                np.savez(os.path.join(a.weights, part_name(a.long, a.member, seed).replace('.json', '.npz')),
                         **{'c.' + k: v.detach().numpy() for k, v in r[3][0].items()}, **{'f.' + k: v.detach().numpy() for k, v in r[3][1].items()})
            json.dump({'seed': seed, 'member': a.member, 'train': r[0], 'held': r[1], 'first': r[2], 'threads': threads, 'seconds': time.time() - t0,
                       'iters': int(data['ro'].shape[0])}, open(f + '.tmp', 'w'))
            os.replace(f + '.tmp', f)
            print('seed %d: train %.3f dB, held-out %.3f dB  (%.0f s)' % (seed, r[0], r[1], time.time() - t0), flush=True)
        return
    done = load(path)
    for seed in range(a.first, a.first + a.n):
        if seed in done:
            continue
        t0 = time.time()
        done[seed] = P.cpu_run(seed, data) + (threads,)
        save(path, done, data, a.long)
        print('seed %d: train %.3f dB, held-out %.3f dB, first loss %.5f  (%.0f s)' % ((seed,) + done[seed][:3] + (time.time() - t0,)), flush=True)


if __name__ == '__main__':
    main()
