#!/bin/bash
# A/B of the strip kernel compiled for two workgroups per CU (tools/libdam_2wg{0,3}.so: -DDAM_STRIP_2WG=0 forward / EPI 0 forms only, =3 all
# 16-channel self-overlapped forms) against the shipped library, same box: whole C3 step and the roofline probe of the layer1 convolution.
mkdir -p gpurun_out/r5
out=${OUT:-gpurun_out/r5/strip_2wg_ab.log}
: > $out
for lib in ${LIBS:-default tools/libdam_2wg0.so tools/libdam_2wg3.so default}; do
  echo "== $lib" >> $out
  if [ "$lib" = default ]; then unset DAM_LIB_PATH; else export DAM_LIB_PATH=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --no-host-stream 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); r=j['roofline']
        print('ms_per_step %.4f (repeat median %.4f)  strip probe: 9-launch mean %.2f us (graph replay %.2f)  forward-only %.2f us  frac %.3f' % (j['ms_per_step'], j['repeat']['ms_per_step_median'], 1e6*r['avg_launch_s'], 1e6*r['avg_launch_graph_replay_s'], 1e6*r['forward_only']['avg_launch_s'], r['frac']))
" >> $out
done
unset DAM_LIB_PATH
cat $out
