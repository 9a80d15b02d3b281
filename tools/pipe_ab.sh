#!/bin/bash
# A/B of the loader-wave tile convolution against conv_igemm_kernel on the thick layers (per-launch us, 200 launches each),
# and a sweep of its tile (DAM_TILE=MBxNB) and loader-wave count (DAM_PIPE_NL) through the diagnostic switches.
run() { python tools/conv_probe.py $1 200 2>/dev/null; }
for l in layer3 layer4 layer5 layer6; do
  echo "== $l"
  DAM_NO_PIPE=1 run $l | sed 's/^/  igemm default tile: /'
  run $l | sed 's/^/  pipe default tile:  /'
  case $l in layer4) tiles="1x2 2x2 4x2 2x1 4x1";; *) tiles="1x4 2x4 4x4 1x2 2x2 4x2";; esac
  for t in $tiles; do
    for nl in 1 2; do
      DAM_TILE=$t DAM_PIPE_NL=$nl run $l | sed "s/^/  pipe tile $t NL $nl: /"
    done
  done
done
