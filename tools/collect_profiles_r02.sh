#!/bin/bash
# Runs ON THE GPU BOX (via gpurun).  Round-2 profiles:
#   1. rocprofv3 --kernel-trace --stats of bench.py (the command whose roofline block is reported)
#   2. kernel trace of 10 steady-state steps, compacted and plain backward (per-kernel time inside the step)
#   3. PMC passes, one counter per run, kernel-trace only: FETCH_SIZE / WRITE_SIZE of 4 steady-state steps (compacted, plain)
#      -> HBM bytes per step and per kernel
#   4. SQ counter groups on the stand-alone fine-pass launches (matrix-pipe busy share, instruction mix)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r2
mkdir -p $O
python tools/prof_step.py train $O/scene.pt 600 > $O/train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- \
  python bench.py --steps 20 --warmup 5 --sustained-steps 50 --no-cpu-baseline < /dev/null > $O/bench.log 2>&1
for mode in 1 0; do
  FASTNERF_COMPACT=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/steps_c$mode -o steps -- \
    python tools/prof_step.py steps $O/scene.pt 10 < /dev/null > $O/steps_c$mode.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    FASTNERF_COMPACT=$mode timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_c${mode}_$c -o pmc -- \
      python tools/prof_step.py steps $O/scene.pt 4 < /dev/null > $O/pmc_c${mode}_$c.log 2>&1
  done
done
# 3b. kernel trace of the stand-alone fine-pass launches (full work: no tile skips its colour branch) -- the launch bench.py's
#     roofline leg times with HIP events
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernels -o kernels -- python tools/prof_step.py kernels $O/scene.pt 5 \
  < /dev/null > $O/kernels.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq$i -o pmc -- python tools/prof_step.py kernels $O/scene.pt 1 \
    < /dev/null > $O/sq$i.log 2>&1
done
rm -f $O/scene.pt $O/bench/bench_kernel_trace.csv $O/sq*/pmc_kernel_trace.csv $O/pmc_*/pmc_kernel_trace.csv   # (not read by the summariser; gpurun copies back at most 64 MiB)
find $O -name "*.csv" | head -40
du -sh $O
