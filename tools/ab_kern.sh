#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average of the stand-alone fine-pass launches, per library variant:
#   tools/ab_kern.sh <mode> <kernel substring> <variant>...     ("base" = the product library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mode=$1; pat=$2; shift; shift
for v in "$@"; do
  if [ $v = base ]; then unset FASTNERF_LIB; else export FASTNERF_LIB=$GRAFT_REPO_ROOT/fast-learning-nerf_amd/variants/$v.so; fi
  O=gpurun_out/abk_$v; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python tools/prof_r03.py kernels $mode 3 < /dev/null > $O/log 2>&1
  python - "$O" "$pat" "$v" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]:
        print("%-8s %-70s %4s x %9.1f us" % (sys.argv[3], r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf $O
done
