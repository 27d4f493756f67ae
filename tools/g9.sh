cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g9
timeout 1200 python -m pytest tests/test_gpu_launch_graphs.py -m gpu -q > gpurun_out/g9/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g9/tests.log
tail -n 25 gpurun_out/g9/tests.log
