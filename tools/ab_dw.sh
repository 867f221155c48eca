#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants on the same box: tools/ab_dw.sh <mode> <variant>...   ("base" = the product library)
mode=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then unset FASTNERF_LIB; else export FASTNERF_LIB=$GRAFT_REPO_ROOT/fast-learning-nerf_amd/variants/$v.so; fi
  echo "== $v"; python tools/time_modes.py $mode 2>&1 | tail -1
done
done
