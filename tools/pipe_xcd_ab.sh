#!/bin/bash
# A/B: XCD-contiguous unit walk of the loader-wave convolution (C3 step and C5 song)
set -e
run() { env "$1" python bench.py $2 --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$1 $2', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run DAM_X=0 "--config C3"
run DAM_PIPE_NO_XCD=1 "--config C3"
run DAM_X=0 "--config C3"
run DAM_PIPE_NO_XCD=1 "--config C3"
run DAM_X=0 "--config C5"
run DAM_PIPE_NO_XCD=1 "--config C5"
