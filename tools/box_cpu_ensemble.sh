#!/bin/bash
# runs ON THE GPU BOX (round 5: the spare GPU-box minutes used for their HOST cores): W workers x T threads of the CPU PSNR ensemble for D seconds.
#   usage: tools/box_cpu_ensemble.sh FIRST_SEED WORKERS THREADS SECONDS
first=$1; W=$2; T=$3; D=$4
mkdir -p gpurun_out/g22_parts
nproc > gpurun_out/box_mem.txt; free -g | head -2 >> gpurun_out/box_mem.txt
pids=()
for ((w = 0; w < W; w++)); do
  s=$((first + 40 * w))
  G22_THREADS=$T OMP_NUM_THREADS=$T MKL_NUM_THREADS=$T python -m oracle.make_golden_psnr_ensemble --parts gpurun_out/g22_parts --seeds $s $((s + 40)) \
    > gpurun_out/g22_parts/w_$s.log 2>&1 &
  pids+=($!)
done
sleep $D
for p in "${pids[@]}"; do kill $p 2>/dev/null; done
wait
ls gpurun_out/g22_parts/*.json 2>/dev/null | wc -l
cat gpurun_out/g22_parts/w_*.log | grep -c "seed"
