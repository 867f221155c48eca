#!/bin/bash
# Builds the variants of tools/slp_bisect.py (here, on the CPU box: hipcc cross-compiles) or runs them (on the GPU box).
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  tools/build_variants.sh slp_on "-fslp-vectorize" slp_on_drain "-fslp-vectorize -DBF_DBG_DRAIN=1" \
    slp_on_wait "-fslp-vectorize -DBF_DBG_DRAIN=2" slp_on_now16 "-fslp-vectorize -DBF_W16=0" slp_off_drain "-DBF_DBG_DRAIN=1"
else
  python tools/slp_bisect.py 40
  for v in slp_on slp_on_drain slp_on_wait slp_on_now16 slp_off_drain; do
    FASTNERF_LIB=fast-learning-nerf_amd/variants/$v.so python tools/slp_bisect.py 40
  done
fi
