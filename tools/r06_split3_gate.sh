#!/bin/bash
# VERDICT r5 task 2's gate: every existing fp32 parity test of the plain VGG pipeline, UNCHANGED, with fc6 forced onto the three-plane split
# (MPN_FC_ARITH=split3 makes models.FastRCNN / MultiPathNet build every VGG pipeline with MPN_FC_SPLIT3; the graph models have no fc6 / fc7 and ignore it)
# -> gpurun_out/r06_split3_gate.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
MPN_FC_ARITH=split3 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_shard.py tests/test_gpu_launch_graphs.py tests/test_gpu_graphs_rigor.py -m gpu -q -s -p no:cacheprovider -k 'not rn50 and not inc and not alexnet' > gpurun_out/r06_split3_gate.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r06_split3_gate.txt
tail -5 gpurun_out/r06_split3_gate.txt
