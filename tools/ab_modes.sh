#!/bin/bash
# runs ON THE GPU BOX: tools/time_modes.py bf16x6 with every library under variants/ (two rounds: box drift shows)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for so in fast-learning-nerf_amd/variants/*.so; do
  echo -n "$(basename $so) : "; FASTNERF_LIB=$PWD/$so timeout 300 python tools/time_modes.py ${MODES:-bf16x6} 2>&1 | tail -1
done; done
