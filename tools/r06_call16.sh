#!/bin/bash
# round 6, call 16: the final binary — full GPU suite, smoke, the default bench line, then the evidence set (tools/collect_profiles_r06.sh)
bash tools/r06_call13.sh
bash tools/collect_profiles_r06.sh > gpurun_out/collect_r06.log 2>&1
tail -3 gpurun_out/collect_r06.log
