#!/bin/bash
# Runs ON THE GPU BOX: average shader clock per kernel = GRBM_GUI_ACTIVE / duration (rocprofv3 --pmc with the kernel trace), over the
# stand-alone fine-pass launches:  tools/clk_probe.sh <mode> <variant>...   ("base" = the product library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mode=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset FASTNERF_LIB; else export FASTNERF_LIB=$GRAFT_REPO_ROOT/fast-learning-nerf_amd/variants/$v.so; fi
  O=gpurun_out/clk_$v; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O -o k -- python tools/prof_r03.py kernels $mode 3 < /dev/null > $O/log 2>&1
  python - "$O" "$v" <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
tr={r["Dispatch_Id"]:(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]))}
A=collections.defaultdict(lambda:[0,0.0,0.0])
for r in csv.DictReader(open(glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0])):
    if r["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
    k=r["Kernel_Name"][:64]; a=A[k]; a[0]+=1; a[1]+=float(r["Counter_Value"]); a[2]+=tr[r["Dispatch_Id"]]
for k,a in A.items():
    if "mlp" in k: print("%-8s %-64s n=%3d  %8.1f us  %6.0f MHz (x XCD count if summed)" % (sys.argv[2],k,a[0],a[2]/a[0]/1e3,a[1]/a[2]*1e3))
PY
  rm -rf $O
done
