#!/bin/bash
# tools/timeline_now.sh <tag> [env...]: kernel trace of bench.py (C3, 5 steps) -> gpurun_out/r5/<tag>_step_timeline.txt + summary
set -e -o pipefail
tag=$1; shift
out=gpurun_out/r5
mkdir -p $out
root=$(pwd)
export TMPDIR=/tmp
( cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace_$tag -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-host-stream ) > $out/last_$tag.log 2>&1 || { tail -5 $out/last_$tag.log; exit 1; }
python3 tools/step_timeline.py $out/trace_$tag > $out/${tag}_step_timeline.txt
python3 tools/trace_summary.py $out/trace_$tag 5 70 > $out/${tag}_kernel_trace_summary.txt
rm -rf $out/trace_$tag
tail -1 $out/${tag}_step_timeline.txt
