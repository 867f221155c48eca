#!/bin/bash
# Runs ON THE GPU BOX: per-kernel stats of K optimisation steps of N rays ($1 = N, $2 = tag), top rows printed.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; TAG=${2:-base}
O=gpurun_out/kstats_n${N}_$TAG; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o steps -- python tools/time_step_sizes.py steps $N 12 < /dev/null > $O/log 2>&1
python - "$O" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=int(r["TotalDurationNs"])
    print("%-110s %5s %9.1f us avg %8.3f ms tot" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, int(r["TotalDurationNs"])/1e6))
print("total ms", tot/1e6, "per step", tot/1e6/12)
PY
cp $(find $O -name '*kernel_stats.csv' | head -1) gpurun_out/kstats_n${N}_$TAG.csv
rm -f $O/*/*kernel_trace.csv
