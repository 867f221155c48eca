#!/bin/bash
# Runs ON THE GPU BOX: SQ counter groups (one rocprofv3 --pmc pass per group, kernel-trace only) over the stand-alone fine-pass
# launches of one math mode ($1); prints per-kernel sums.  $2 = optional tag of the output directory (FASTNERF_LIB selects the library).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq_probe_$1$2; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o pmc -- python tools/prof_r03.py kernels $1 1 < /dev/null > $O/g$i.log 2>&1
done
python - "$O" <<'PY'
import csv,glob,sys,collections
T=collections.defaultdict(lambda: collections.defaultdict(float))
D=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/g*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        T[r["Kernel_Name"][:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
for f in glob.glob(sys.argv[1]+"/g*/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        D[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in T.items():
    if 'mlp' not in k: continue
    d=D.get(k,[0]); print(k, ' us/launch (profiled passes): %.0f' % (sum(d)/len(d)))
    for c in sorted(v): print("   %-32s %.4g" % (c, v[c]))
    if 'SQ_WAVE_CYCLES' in v and v['SQ_WAVE_CYCLES']:
        print("   matrix pipe busy (2 waves/SIMD)   %.1f %%" % (100*v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(2*v['SQ_WAVE_CYCLES'])))
PY
rm -rf $O/g*/*/*kernel_trace.csv
