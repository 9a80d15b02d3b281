#!/bin/bash
# A/B: record-table budget a BatchNorm partial pass leaves its fused consumer (more records = more workgroups in the partial pass)
set -e
run() { env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$*', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run DAM_X=0
run DAM_BN_FA_PARTS_KB=80
run DAM_BN_FA_PARTS_KB=48
run DAM_BN_FA_PARTS_KB=32
run DAM_X=0
