#!/bin/bash
# round 6, call 14: per-map range-max table builds on the pooling stream — bit-identity tests, A/B of the knob on configs[2], bench lines
mkdir -p gpurun_out
python -m pytest tests/test_gpu_roipool.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -x -k "range_max or pooling_stream or lanes or multipathnet or mpnet" 2>&1 | tail -5 > gpurun_out/tables_tests.txt
MPN_FLAVOUR=debug timeout 900 python tools/hook_ab.py c3 12 base tables_lazy=0 > gpurun_out/tables_ab.txt 2>&1
python bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
python bench.py --config c3 --fc-arith split3 > gpurun_out/bench_c3_split3.json 2>> gpurun_out/bench_c3.err
cat gpurun_out/tables_tests.txt gpurun_out/tables_ab.txt gpurun_out/bench_c3.json gpurun_out/bench_c3_split3.json
