#!/bin/bash
# runs ON THE GPU BOX: socket power and shader clock while the bench step loops (is the chip power-limited?)
cd "$GRAFT_REPO_ROOT"
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|fclk\|mclk" | head -12
python bench.py --steps 400 --warmup 3 > /tmp/bench_long.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
tail -c 400 /tmp/bench_long.log
