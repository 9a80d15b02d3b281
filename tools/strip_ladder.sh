#!/bin/bash
# tools/strip_ladder.sh: builds the timing-only variants of dam_conv_strip.hip for the ablation ladder (run HERE: hipcc
# cross-compiles; the .so files travel to the GPU box), then on the box: python tools/strip_ladder.py
set -e
cd $(dirname $0)/..
python deep-audio-mixer_amd/build.py > /dev/null
v() { name=$1; shift; tools/build_variant.sh ladder_$name dam_conv_strip.hip "$@" > /dev/null; echo "  $name: $*"; }
v full
v noyield '-DDAM_STRIP_YIELD=do{}while(0)'
v nostats -DDAM_DIAG_NO_STATS
v nogeom -DDAM_DIAG_NO_GEOM
v nostore -DDAM_DIAG_NO_STORE
v nowriteout -DDAM_DIAG_NO_WRITEOUT
v noload -DDAM_DIAG_NO_LOAD
v mfma_only -DDAM_DIAG_NO_LOAD -DDAM_DIAG_NO_WRITEOUT -DDAM_DIAG_NO_GEOM
v mfma_only_noyield -DDAM_DIAG_NO_LOAD -DDAM_DIAG_NO_WRITEOUT -DDAM_DIAG_NO_GEOM '-DDAM_STRIP_YIELD=do{}while(0)'
v nomfma -DDAM_DIAG_NO_MFMA
v stamps -DDAM_STAMPS
