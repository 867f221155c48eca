#!/bin/bash
# Runs ON THE GPU BOX: per-kernel stats (rocprofv3 --kernel-trace --stats) of an arbitrary python command: tools/kstats_cmd.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT="gpurun_out/kstats_$TAG"
mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o steps -- python "$@" < /dev/null > "$OUT/log" 2>&1
tail -3 "$OUT/log"
python - "$OUT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-100s %6s %9.1f us avg %8.2f ms tot %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, int(r["TotalDurationNs"])/1e6, 100*int(r["TotalDurationNs"])/tot))
print("total ms", tot/1e6)
PY
