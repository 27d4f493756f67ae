cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/g12
export MPN_FLAVOUR=debug
for v in 1 0; do
  rm -rf /tmp/kt$v
  MPN_HOOKS="bf16_bdir=$v" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt$v -o kt --output-format csv -- python $R/tools/bench_inception.py 2000 mpn > $R/gpurun_out/g12/inc_bdir$v.txt 2>&1
  cp $(find /tmp/kt$v -name "*kernel_stats.csv" | head -1) $R/gpurun_out/g12/inc_bdir${v}_kernel_stats.csv
  MPN_HOOKS="bf16_bdir=$v" timeout 300 python $R/tools/bench_inception.py 2000 mpn > $R/gpurun_out/g12/inc_plain_bdir$v.txt 2>&1
  MPN_HOOKS="bf16_bdir=$v" timeout 300 python $R/tools/bench_resnet.py 50 1000 mpn bf16 > $R/gpurun_out/g12/rn_plain_bdir$v.txt 2>&1
done
grep -h "ms/image" $R/gpurun_out/g12/*plain*.txt
head -6 $R/gpurun_out/g12/inc_bdir1_kernel_stats.csv | cut -c1-120,240-330
head -6 $R/gpurun_out/g12/inc_bdir0_kernel_stats.csv | cut -c1-120,240-330
