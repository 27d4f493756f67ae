cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/plain
timeout 200 python tools/bench_resnet.py 50 1000 bf16 > gpurun_out/plain/r04_resnet50_bf16_timing.txt 2>&1
timeout 200 python tools/bench_resnet.py 50 1000 > gpurun_out/plain/r04_resnet50_timing.txt 2>&1
timeout 200 python tools/bench_inception.py 2000 > gpurun_out/plain/r04_inception_bf16_timing.txt 2>&1
timeout 200 python tools/bench_inception.py 2000 mpn > gpurun_out/plain/r04_inception_mpn_bf16_timing.txt 2>&1
timeout 200 python tools/bench_mpn.py > gpurun_out/plain/r04_mpn_timing.txt 2>&1
for f in gpurun_out/plain/*.txt; do echo "== $f"; grep -E "ms/image|TFLOP" $f | head -3; done
