#!/bin/bash
# runs ON THE GPU BOX: the whole GPU suite while the box's idle host cores extend the CPU PSNR ensemble G22 (one thread per seed);
# usage: tools/gpu_suite_with_g22.sh FIRST_SEED N_WORKERS [pytest args]
first=$1; n=$2; shift 2
mkdir -p gpurun_out/g22_parts
free -g | head -2 > gpurun_out/box_mem.txt; nproc >> gpurun_out/box_mem.txt
# a worker holds ~1.7 GB (256 rays x 192 samples of autograd state): never plan more than the box's free memory / 3 GB or its cores - 24
avail_gb=$(awk '/MemAvailable/ {print int($2 / 1048576)}' /proc/meminfo)
cap=$(( avail_gb / 3 )); cores=$(( $(nproc) - 24 - 4 * $(echo $G23_SEEDS | wc -w) ))
[ $cap -lt $n ] && n=$cap
[ $cores -lt $n ] && n=$cores
[ $n -lt 0 ] && n=0
echo "workers: $n (MemAvailable ${avail_gb} GB)" >> gpurun_out/box_mem.txt
pids=()
for ((w = 0; w < n; w++)); do
  s=$((first + w))
  G22_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 nice -n 10 python -m oracle.make_golden_psnr_ensemble --parts gpurun_out/g22_parts --seeds $s $((s + 1)) \
    > gpurun_out/g22_parts/w_$s.log 2>&1 &
  pids+=($!)
done
if [ -n "$G23_SEEDS" ]; then    # the longer horizon (G23): 1000 iterations, 4 threads per seed
  for s in $G23_SEEDS; do
    G22_THREADS=4 OMP_NUM_THREADS=4 MKL_NUM_THREADS=4 nice -n 10 python -m oracle.make_golden_psnr_ensemble --long --parts gpurun_out/g22_parts --seeds $s $((s + 1)) \
      > gpurun_out/g22_parts/wl_$s.log 2>&1 &
    pids+=($!)
  done
fi
python -m pytest tests -m gpu -q "$@" > gpurun_out/gputest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gputest.log
tail -5 gpurun_out/gputest.log
t0=$(date +%s)
# the box's GPU minutes are what the round budgets: wait at most G22_WAIT seconds (default 120) for the workers once the GPU work is done,
# then stop the rest (a seed's JSON appears only when the seed is complete: nothing partial is kept)
deadline=$(( t0 + ${G22_WAIT:-120} ))
for p in "${pids[@]}"; do
  while kill -0 $p 2>/dev/null && [ $(date +%s) -lt $deadline ]; do sleep 5; done
  kill $p 2>/dev/null
done
wait
echo "waited $(( $(date +%s) - t0 )) s more for the ensemble workers; parts: $(ls gpurun_out/g22_parts/*.json 2>/dev/null | wc -l)"
