#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gpu_suite_2.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gpu_suite_2.txt
tail -12 gpurun_out/r06_gpu_suite_2.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
