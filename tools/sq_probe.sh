#!/bin/bash
# Runs ON THE GPU BOX: SQ counter groups (one rocprofv3 --pmc pass per group, kernel-trace only) over the stand-alone fine-pass
# launches of one math mode ($1); prints per-kernel sums.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq_probe_$1; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o pmc -- python tools/prof_r03.py kernels $1 1 < /dev/null > $O/g$i.log 2>&1
done
python - "$O" <<'PY'
import csv,glob,sys,collections
T=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+"/g*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        T[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in T.items():
    if 'mlp' not in k: continue
    print(k)
    for c in sorted(v): print("   %-32s %.4g" % (c, v[c]))
PY
rm -rf $O/g*/*/*kernel_trace.csv
