#!/bin/bash
# round 6, call 17 (measurement): per-kernel durations of configs[2]'s head with nothing overlapped (one lane, pooling on the launch stream)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_c3s && MPN_FLAVOUR=debug timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_c3s -o kt --output-format csv -- python $R/tools/hook_ab.py c3 6 tower_lanes=0,pool_overlap=0 > /tmp/kt_c3s.out 2>&1
cp $(find /tmp/kt_c3s -name "*kernel_stats.csv" | head -1) $R/gpurun_out/c3_serial_kernel_stats.csv
cp $(find /tmp/kt_c3s -name "*kernel_trace.csv" | head -1) $R/gpurun_out/c3_serial_kernel_trace.csv
tail -3 /tmp/kt_c3s.out
head -8 $R/gpurun_out/c3_serial_kernel_stats.csv | cut -c1-200
