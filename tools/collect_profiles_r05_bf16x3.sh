cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r5; mkdir -p $O; mode=bf16x3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/steps_$mode -o steps -- python tools/prof_r03.py steps $mode 6 < /dev/null > $O/steps_$mode.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${mode}_$c -o pmc -- python tools/prof_r03.py steps $mode 4 < /dev/null > $O/pmc_${mode}_$c.log 2>&1
done
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq_${mode}_$i -o pmc -- python tools/prof_r03.py kernels $mode 1 < /dev/null > $O/sq_${mode}_$i.log 2>&1
done
rm -f $O/sq_*/pmc_kernel_trace.csv $O/pmc_*/pmc_kernel_trace.csv
find $O -name "*.csv" | wc -l
