#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider -k "three_plane" > gpurun_out/r06_call6_split3_full.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r06_call6_split3_full.txt
grep -v amdgpu gpurun_out/r06_call6_split3_full.txt | grep "N = \|passed\|failed\|rc \|^E " | cut -c1-400
for r in 2 8; do MPN_FLAVOUR=debug python - <<PY
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import multipathnet_amd
multipathnet_amd.load().mpn_debug_set_split3_ranges($r)
import subprocess
PY
done
bash tools/r06_split3_gate.sh
