#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace of bench.py in both math modes and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel-trace only) on the stand-alone MLP launches of tools/prof_kernels.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r1
mkdir -p $O
for mode in bf16x3 fp32; do
  FASTNERF_MATH=$mode timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_$mode -o bench -- \
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench_$mode.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    FASTNERF_MATH=$mode timeout 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${mode}_$c -o pmc -- \
      python tools/prof_kernels.py 2 < /dev/null > $O/pmc_${mode}_$c.log 2>&1
  done
done
ls -R $O | head -50
