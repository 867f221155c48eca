#!/bin/bash
# runs ON THE GPU BOX: time the MLP kernels (tools/time_mlp.py) with every library under variants/
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for so in fast-learning-nerf_amd/variants/*.so; do
  echo -n "$(basename $so) : "; FASTNERF_LIB=$PWD/$so python tools/time_mlp.py 2>&1 | tail -1
done; done
