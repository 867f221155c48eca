#!/bin/bash
# runs ON THE GPU BOX: an arbitrary GPU command line while the box's idle host cores extend the CPU PSNR ensemble G22 (one thread per worker, G22_PER_WORKER
# seeds per worker; G23_SEEDS="a b c": 1000-iteration runs of the longer horizon, 4 threads each).  The workers never outlive the GPU work by more than
# G22_WAIT seconds (default 120): GPU-box minutes are the budget.
#   usage: tools/with_g22_workers.sh FIRST_SEED N_WORKERS 'gpu command line'
first=$1; n=$2; cmd=$3
mkdir -p gpurun_out/g22_parts
free -g | head -2 > gpurun_out/box_mem.txt; nproc >> gpurun_out/box_mem.txt
avail_gb=$(awk '/MemAvailable/ {print int($2 / 1048576)}' /proc/meminfo)
cap=$(( avail_gb / 3 )); cores=$(( $(nproc) / 2 - 24 - 4 * $(echo $G23_SEEDS | wc -w) ))   # physical cores: with one worker per hardware thread (192 on 256) not one seed finished in 40 minutes
[ $cap -lt $n ] && n=$cap
[ $cores -lt $n ] && n=$cores
[ $n -lt 0 ] && n=0
echo "workers: $n (MemAvailable ${avail_gb} GB)" >> gpurun_out/box_mem.txt
pids=()
k=${G22_PER_WORKER:-3}      # seeds per worker, one after the other (a worker is stopped when the GPU work is done: the later ones may not happen)
for ((w = 0; w < n; w++)); do
  s=$((first + k * w))
  G22_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 nice -n 10 python -m oracle.make_golden_psnr_ensemble --parts gpurun_out/g22_parts --seeds $s $((s + k)) \
    > gpurun_out/g22_parts/w_$s.log 2>&1 &
  pids+=($!)
done
for s in $G23_SEEDS; do
  G22_THREADS=4 OMP_NUM_THREADS=4 MKL_NUM_THREADS=4 nice -n 10 python -m oracle.make_golden_psnr_ensemble --long --parts gpurun_out/g22_parts --seeds $s $((s + 1)) \
    > gpurun_out/g22_parts/wl_$s.log 2>&1 &
  pids+=($!)
done
bash -c "$cmd"
echo "gpu command rc=$?"
t0=$(date +%s)
deadline=$(( t0 + ${G22_WAIT:-120} ))
for p in "${pids[@]}"; do
  while kill -0 $p 2>/dev/null && [ $(date +%s) -lt $deadline ]; do sleep 5; done
  kill $p 2>/dev/null
done
wait
echo "waited $(( $(date +%s) - t0 )) s more for the ensemble workers; parts: $(ls gpurun_out/g22_parts/*.json 2>/dev/null | wc -l)"
