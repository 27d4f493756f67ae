#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite on the new code (symbol map, ROI bin rule, fence, gate, hand-off test) + the tower knock-out + the fence's cost
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gpu_suite_1.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gpu_suite_1.txt
tail -5 gpurun_out/r06_gpu_suite_1.txt
MPN_FLAVOUR=debug python tools/tower_knockout.py c5 c4 12 > gpurun_out/r06_tower_knockout.txt 2>&1; cat gpurun_out/r06_tower_knockout.txt | tail -12
( for f in 1 0; do for m in 300 1000; do echo "== MPN_FUSED_FENCE=$f M=$m"; MPN_FUSED_FENCE=$f python tools/bench_nms.py $m distinct,fewties,ties 0; done; done ) > gpurun_out/r06_nms_fence.txt 2>&1; cat gpurun_out/r06_nms_fence.txt | grep -v amdgpu.ids
python bench.py > gpurun_out/r06_bench_1.json 2> gpurun_out/r06_bench_1.err; tail -c 600 gpurun_out/r06_bench_1.json
