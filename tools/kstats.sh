#!/bin/bash
# Runs ON THE GPU BOX: per-kernel stats of 6 optimisation steps in one math mode ($1), top rows printed.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/kstats_$1; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o steps -- python tools/prof_r03.py steps $1 6 < /dev/null > $O/log 2>&1
python - "$O" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=int(r["TotalDurationNs"])
    print("%-100s %5s %10.1f us avg %8.3f ms tot" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, int(r["TotalDurationNs"])/1e6))
print("total ms", tot/1e6)
PY
rm -f $O/*/*kernel_trace.csv
