#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  bash tools/collect_profiles.sh <round, e.g. r06>  [MODES="bf16x6 fp32 bf16x3"]
# The round's profiles (summarised by `python tools/summarize_prof.py <round>` into profiles/<round>_*; one script for every round since r06,
# the per-round copies r02 .. r05 were folded into it):
#   1. rocprofv3 --kernel-trace --stats of bench.py (the command whose roofline block is reported), CPU legs off
#   2. per math mode: kernel trace of 6 optimisation steps of the headline protocol
#   3. per math mode: PMC passes, ONE counter per run, kernel-trace only: FETCH_SIZE / WRITE_SIZE of 4 steps -> HBM bytes per kernel / step
#   4. per math mode: SQ counter groups on the stand-alone fine-pass launches (matrix-pipe busy share, instruction mix)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=${1:-r06}
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- \
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --psnr-iters 0 < /dev/null > $O/bench.log 2>&1
for mode in ${MODES:-bf16x6 fp32 bf16x3}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/steps_$mode -o steps -- \
    python tools/prof_r03.py steps $mode 6 < /dev/null > $O/steps_$mode.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${mode}_$c -o pmc -- \
      python tools/prof_r03.py steps $mode 4 < /dev/null > $O/pmc_${mode}_$c.log 2>&1
  done
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
             "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq_${mode}_$i -o pmc -- \
      python tools/prof_r03.py kernels $mode 1 < /dev/null > $O/sq_${mode}_$i.log 2>&1
  done
done
rm -f $O/bench/bench_kernel_trace.csv $O/sq_*/pmc_kernel_trace.csv $O/pmc_*/pmc_kernel_trace.csv
find $O -name "*.csv" | wc -l
du -sh $O
grep -h "metric" $O/bench.log | cut -c1-300
