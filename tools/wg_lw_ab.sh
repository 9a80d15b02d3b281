#!/bin/bash
# A/B of the tile weight gradient's loader-wave form (DAM_WG_LW=0: plain form) on C2, C1, C3: ms per step, alternating runs on one box
mkdir -p gpurun_out/r5
out=gpurun_out/r5/wg_lw_ab.log
: > $out
for cfg in C2 C1 C3; do
  for lw in 1 0 1 0; do
    DAM_WG_LW=$lw timeout -k 10 300 python bench.py --config $cfg --steps 10 --no-cpu-baseline --no-host-stream --no-roofline 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('$cfg DAM_WG_LW=$lw ms_per_step %.4f median %.4f' % (j['ms_per_step'], j['repeat']['ms_per_step_median']))
" >> $out
  done
done
cat $out
