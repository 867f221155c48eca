#!/bin/bash
# runs ON THE GPU BOX: tools/time_step_sizes.py (the fused step at 512 / 1024 / 1920 / 4096 rays) with every library under variants/, two rounds
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for so in fast-learning-nerf_amd/variants/*.so; do
  echo "== $(basename $so) (round $rep)"; FASTNERF_LIB=$PWD/$so timeout 400 python tools/time_step_sizes.py 2>&1 | tail -5
done; done
