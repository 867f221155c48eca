#!/bin/bash
# usage: tools/build_variants.sh name1 "flags1" name2 "flags2" ...   -> fast-learning-nerf_amd/variants/<name>.so
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  FASTNERF_VARIANT="$1" FASTNERF_CFLAGS="$2" python fast-learning-nerf_amd/build.py --force > /dev/null || echo "FAILED $1"
  shift 2
done
ls -la fast-learning-nerf_amd/variants/
