#!/bin/bash
# round 6, call 18: packed mix-GEMM rows (+ the XCD tile walk for any block count) — GEMM / MultiPathNet parity and invariance tests, then A/B on configs[2]
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_roipool.py tests/test_gpu_resnet.py -q -x -k "linear or gemm or mpnet or multipathnet or MultiPathNet or packed or lanes or pooling_stream or mix or adaptive or pointwise or resnet" 2>&1 | tail -6 > gpurun_out/packed_tests.txt
cat gpurun_out/packed_tests.txt
MPN_FLAVOUR=debug timeout 900 python tools/hook_ab.py c3 12 base mix_packed=0 > gpurun_out/packed_ab.txt 2>&1
cat gpurun_out/packed_ab.txt
