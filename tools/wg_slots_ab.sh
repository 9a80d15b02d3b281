#!/bin/bash
# A/B: slots assumed by the tile weight gradient's makespan rule (fewer, longer workgroups = fewer slab bytes) on the C3 step
set -e
run() { env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print('$*', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
run DAM_X=0
run DAM_WG_SLOTS=256
run DAM_WG_SLOTS=768
run DAM_X=0
