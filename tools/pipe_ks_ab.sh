#!/bin/bash
# tools/pipe_ks_ab.sh: conv_pipe_kernel on the deep stages, K split over the wave pairs (default) against DAM_PIPE_KS=1, plain /
# with the statistics epilogue / with the fused input affine (per-launch us, 200 launches)
for l in layer5 layer6 layer4; do
  for o in "" ":stats" ":affine" ":stats+affine"; do
    a=$(python tools/conv_probe.py $l$o 200 2>/dev/null | sed 's/.*: //')
    b=$(DAM_PIPE_KS=1 python tools/conv_probe.py $l$o 200 2>/dev/null | sed 's/.*: //')
    printf "%-22s default %s | DAM_PIPE_KS=1 %s\n" "$l$o" "$a" "$b"
  done
done
