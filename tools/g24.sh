cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g24
for v in "bf16_bdir=2 bf16_bdir_ver=8" "bf16_bdir=2 bf16_bdir_ver=8 bf16_bdir_abl=1001" "bf16_bdir=2 bf16_bdir_ver=8 bf16_bdir_abl=1000"; do
  echo "=== variant: $v"
  timeout 300 python tools/bench_conv_bf16.py all --clk $v 2>&1 | grep -v amdgpu.ids
done > gpurun_out/g24/deep.txt
python - <<'PY'
import re
for l in open('gpurun_out/g24/deep.txt'):
    if 'variant' in l or 'tower' in l: print(l.strip()[:110])
PY
