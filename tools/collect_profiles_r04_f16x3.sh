#!/bin/bash
# Runs ON THE GPU BOX (via gpurun).  Round-4 evidence of the f16x3 mode (summarised into profiles/r04_f16x3_*):
#   1. the whole GPU suite (four math modes incl. f16x3; G22 paired PSNR in three)      -> gputest.log
#   2. bench.py with its defaults (the driver's command): the line with the f16x3_mode sibling block  -> bench_line.json
#   3. rocprofv3 --kernel-trace --stats of 6 optimisation steps in f16x3; PMC FETCH_SIZE / WRITE_SIZE (one counter per run);
#      SQ counter groups on the stand-alone fine-pass launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f16x3
rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -s ) > $O/gputest.log 2>&1
timeout 900 python bench.py < /dev/null > $O/bench.log 2>&1
grep -h '"metric"' $O/bench.log | tail -1 > $O/bench_line.json
mode=f16x3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/steps_$mode -o steps -- \
  python tools/prof_r03.py steps $mode 6 < /dev/null > $O/steps_$mode.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${mode}_$c -o pmc -- \
    python tools/prof_r03.py steps $mode 4 < /dev/null > $O/pmc_${mode}_$c.log 2>&1
done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq_${mode}_$i -o pmc -- \
    python tools/prof_r03.py kernels $mode 1 < /dev/null > $O/sq_${mode}_$i.log 2>&1
done
rm -f $O/steps_*/*/*kernel_trace.csv
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
tail -5 $O/gputest.log
cut -c1-400 $O/bench_line.json
