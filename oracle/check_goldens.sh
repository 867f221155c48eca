#!/bin/bash
# (build container only: needs /root/reference)  Regenerate EVERY reference-derived fixture of tests/golden/ from the reference into a
# scratch directory and compare with the committed files: arrays bit for bit (npz members), other files byte for byte.  The two PSNR
# ensembles (G22, G23) are recorded from the ORACLE, not from the reference (hours of host time; free runs are chaotic): for those, K
# recorded seeds of G22 are re-run with the recorded thread count and compared at 1e-3 dB (G22_CHECK_SEEDS, default 2; 0 skips).  G23's seeds
# (1000 iterations, ~1 h of one core each) are not re-run here: `G22_THREADS=1 python -m oracle.make_golden_psnr_ensemble --long --check 1` does it.
# G24 / G25 (round 6) are the NULL members of the same two ensembles -- the same generator from weights x (1 + 1e-6 N(0, 1)), one thread per run:
# G24_CHECK_SEEDS=k re-runs k of G24's seeds (11 minutes of one core each; default 0), `... --long --member 1 --check 1` one of G25's.
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d /tmp/goldens.XXXXXX)
export FASTNERF_GOLDEN_OUT="$D"
for g in make_golden make_golden_pp make_golden_pp_render make_golden_ssim make_golden_loaders make_golden_treepkl \
         make_golden_pp_loader make_golden_render_path make_golden_noview make_golden_prob make_golden_pp_ckpt; do
  python oracle/$g.py > "$D/$g.log" 2>&1 || { echo "FAILED $g (see $D/$g.log)"; exit 1; }
done
python - "$D" <<'PY'
import os, sys
import numpy as np
new, old = sys.argv[1], 'tests/golden'
bad = 0
ORACLE_RECORDED = ('g22_psnr_cpu_ensemble.npz', 'g23_psnr_cpu_long.npz',         # checked by make_golden_psnr_ensemble --check below
                   'g24_psnr_cpu_null_m1.npz', 'g25_psnr_cpu_null_long_m1.npz')  # their NULL members (same generator, --member 1): G24_CHECK_SEEDS
names = sorted(f for f in os.listdir(old) if not f.startswith('.') and f not in ORACLE_RECORDED)
for f in names:
    a, b = os.path.join(old, f), os.path.join(new, f)
    if not os.path.exists(b):
        print('NOT REGENERATED', f); bad += 1; continue
    if f.endswith('.npz'):
        x, y = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
        same = sorted(x.files) == sorted(y.files) and all(
            x[k].dtype == y[k].dtype and x[k].shape == y[k].shape and x[k].tobytes() == y[k].tobytes() for k in x.files)
    else:
        same = open(a, 'rb').read() == open(b, 'rb').read()
    print('ok ' if same else 'DIFFERS', f)
    bad += not same
print(f'{len(names) - bad} of {len(names)} fixtures reproduce bit for bit')
sys.exit(1 if bad else 0)
PY
K=${G22_CHECK_SEEDS:-2}
if [ "$K" != "0" ]; then
  unset FASTNERF_GOLDEN_OUT
  G22_THREADS=8 python -m oracle.make_golden_psnr_ensemble --check "$K" || { echo "G22: recorded seeds do not reproduce"; exit 1; }
  echo "G22: $K recorded seeds reproduce at 1e-3 dB"
fi
K4=${G24_CHECK_SEEDS:-0}
if [ "$K4" != "0" ]; then
  unset FASTNERF_GOLDEN_OUT
  G22_THREADS=1 python -m oracle.make_golden_psnr_ensemble --member 1 --check "$K4" || { echo "G24: recorded null members do not reproduce"; exit 1; }
  echo "G24: $K4 recorded null members reproduce at 1e-3 dB"
fi
