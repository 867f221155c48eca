#!/bin/bash
# runs ON THE GPU BOX: per-kernel averages (rocprofv3 --kernel-trace --stats over 12 steps of $1 rays, default 4096) with every library under variants/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-4096}; PAT=${2:-mlp_bwd_dw6}
for so in fast-learning-nerf_amd/variants/*.so; do
  v=$(basename $so .so); O=gpurun_out/abk_$v; rm -rf $O; mkdir -p $O
  FASTNERF_LIB=$PWD/$so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o steps -- python tools/time_step_sizes.py steps $N 12 < /dev/null > $O/log 2>&1
  echo "== $v"
  python - "$O" "$PAT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=int(r["TotalDurationNs"])
    if sys.argv[2] in r["Name"]:
        print("  %-90s %5s %9.1f us avg" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
print("  per step ms", tot/1e6/12)
PY
  rm -rf $O
done
