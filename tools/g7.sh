cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g7
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_graphs_rigor.py > gpurun_out/g7/suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/g7/suite.log
tail -n 15 gpurun_out/g7/suite.log
