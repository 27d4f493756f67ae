#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_alexnet.py tests/test_gpu_resnet.py tests/test_gpu_inception.py tests/test_gpu_shard.py tests/test_gpu_launch_graphs.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r06_call2_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call2_tests.txt
tail -4 gpurun_out/r06_call2_tests.txt
MPN_FLAVOUR=debug python tools/tower_lanes_ab.py c3 c4 c5 12 > gpurun_out/r06_tower_lanes_ab.txt 2>&1; grep -v amdgpu gpurun_out/r06_tower_lanes_ab.txt | tail -12
python -m pytest tests/test_gpu_graphs_rigor.py -m gpu -q -s -p no:cacheprovider -k "bf16" > gpurun_out/r06_call2_rigor_bf16.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call2_rigor_bf16.txt
grep "bf16 device vs plain\|passed\|failed" gpurun_out/r06_call2_rigor_bf16.txt | cut -c1-900
