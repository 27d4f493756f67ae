#!/bin/bash
# round 6, call 23: the pooling stream = the side stream (one stream fewer per handle) — MultiPathNet tests (pipelined forms, launch graphs, shards), then the host-fed configs[2] line
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_launch_graphs.py tests/test_gpu_shard.py tests/test_gpu_graphs_rigor.py -q -x -k "mpnet or multipathnet or MultiPathNet or packed or lanes or pooling_stream or mix or mpn or vggmpn or graph or shard" 2>&1 | tail -4 > gpurun_out/pool_on_side_tests.txt
cat gpurun_out/pool_on_side_tests.txt
for rep in 1 2; do python bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[2] host-fed', d['ms_per_step'], 'resident', round(d['value']/d['value_inputs_resident']*d['ms_per_step'],4))"; done > gpurun_out/pool_on_side.txt 2>&1
cat gpurun_out/pool_on_side.txt
