for cfg in "0 0" "1 0" "1 248" "1 240" "1 224"; do set -- $cfg; python -c "
import sys; sys.argv=['bench.py','--no-cpu-baseline']
import multipathnet_amd; L=multipathnet_amd.load(); L.mpn_debug_set_conv_mode($1); L.mpn_debug_set_conv_persist_blocks($2)
import bench; bench.main()" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $1 P $2:', d['value'], d['ms_per_step'], d['kernels']['conv_wino']['ms_per_image'])"; done
