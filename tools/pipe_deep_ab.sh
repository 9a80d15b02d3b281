#!/bin/bash
# tools/pipe_deep_ab.sh: per-launch us of conv_pipe_kernel on the deep stages for the shipped library and diagnostic builds
# (tools/build_variant.sh <name> dam_conv_pipe.hip -D...): usage  tools/pipe_deep_ab.sh "<variant names>" "<layers>"
variants=${1:-"now9 w0 now9_w0"}
layers=${2:-"layer4 layer5 layer6 layer5s2 layer6s2"}
for l in $layers; do
  printf "%-10s default: " $l; python tools/conv_probe.py $l 200 2>/dev/null | tr '\n' ' '; echo
  for v in $variants; do
    printf "%-10s %-10s " $l $v; DAM_LIB_PATH=tools/libdam_pipe_$v.so python tools/conv_probe.py $l 200 2>/dev/null | tr '\n' ' '; echo
  done
done
