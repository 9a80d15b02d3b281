#!/bin/bash
# A/B of the loader-wave tile convolution (conv_pipe_kernel) against conv_igemm_kernel on the thick layers (per-launch us,
# 200 launches each), and a sweep of its tile through the diagnostic switch DAM_TILE=MBxNB (which also disables split-K).
run() { python tools/conv_probe.py $1 200 2>/dev/null; }
for l in layer3 layer4 layer5 layer6 layer3s2 layer4s2 layer5s2 layer6s2; do
  echo "== $l"
  DAM_NO_PIPE=1 run $l | sed 's/^/  igemm, default tile: /'
  run $l | sed 's/^/  pipe,  default tile: /'
  case $l in layer4*) tiles="2x2 1x2 2x1 1x1";; *) tiles="1x4 2x2 1x2 1x1";; esac
  for t in $tiles; do
    DAM_TILE=$t run $l | sed "s/^/  pipe,  tile $t: /"
  done
done
