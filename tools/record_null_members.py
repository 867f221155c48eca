"""Records the NULL members of the paired PSNR design in the build container: for seeds already in G22 / G23 a second
CPU-oracle run from weights x (1 + 1e-6 N(0, 1)) (oracle/psnr_protocol.py jitter_weights).  One thread per run, WORKERS runs at a time, the
1000-iteration runs first (they also keep their final weights under oracle/_study/ for the cross-evaluation).  CPU only.
  python tools/record_null_members.py [workers] [n_short] [n_long]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_short = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n_long = int(sys.argv[3]) if len(sys.argv) > 3 else 20
parts = os.path.join(ROOT, 'oracle', '_study', 'parts')
z22 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g22_psnr_cpu_ensemble.npz'))
z23 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g23_psnr_cpu_long.npz'))
long_seeds = [int(s) for s, h in zip(z23['seeds'], z23['held_out_psnr_db']) if h > 15.0][:n_long]
short_seeds = [int(s) for s, h, t in zip(z22['seeds'], z22['held_out_psnr_db'], z22['threads']) if h > 15.0 and t == 1][:n_short]
jobs = [(s, True) for s in long_seeds] + [(s, False) for s in short_seeds]
env = dict(os.environ, G22_THREADS='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')


def run(job):
    seed, long = job
    cmd = ['nice', '-n', '10', sys.executable, '-m', 'oracle.make_golden_psnr_ensemble', '--parts', parts, '--seed-list', str(seed), '--member', '1']
    if long:
        cmd += ['--long', '--weights', os.path.join(ROOT, 'oracle', '_study', 'weights')]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)


with ThreadPoolExecutor(workers) as ex:
    list(ex.map(run, jobs))
