#!/bin/bash
# A/B of the one-launch down-sampling kernels' switches on the C3 step (ms per step, 200 steps each; profiles/r04_s2_kernels_ab.txt)
set -e
mkdir -p gpurun_out/r4
run() { env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$*', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run DAM_X=0
run DAM_CS2_STREAM_MB=1
run DAM_CS2_STREAM_MB=2
run DAM_X=0
