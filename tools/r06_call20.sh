#!/bin/bash
# round 6, call 20 (measurement): hardware-queue sharing between a handle's streams — GPU_MAX_HW_QUEUES (ROCm default 4) on the host-fed lines
mkdir -p gpurun_out
run() { python bench.py $1 --no-cpu-baseline --sustained-seconds 0 --no-power-sensitivity --no-fc-split3-leg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'GPU_MAX_HW_QUEUES=$3', 'host-fed', d['ms_per_step'], 'resident', round(1e3*d['config'].get('rois',0)/d['value_inputs_resident'],4) if 0 else round(d['value']/d['value_inputs_resident']*d['ms_per_step'],4))"; }
for rep in 1 2; do for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  run "" headline $q
  run "--config c3 --steps 20 --warmup 5" c3 $q
  if [ $rep = 1 ]; then run "--config c4 --dtype bf16 --steps 20 --warmup 5" c4_bf16 $q; run "--config c5 --steps 10 --warmup 3" c5 $q; fi
done; done > gpurun_out/hwq.txt 2>&1
cat gpurun_out/hwq.txt
