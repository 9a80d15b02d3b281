#!/bin/bash
# kernel + memory-copy trace of the reference-API loop (bench.py --via-trainer): per-step period, kernel time, largest gaps
set -e
out=gpurun_out/r04vt; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/trace -- python3 $root/bench.py --via-trainer --steps 96 ) > $out/last.log 2>&1 || { tail -5 $out/last.log; exit 1; }
python3 tools/step_gaps.py $out/trace > $out/via_trainer_step_gaps.txt
python3 tools/step_timeline.py $out/trace > $out/via_trainer_step_timeline.txt || true
rm -rf $out/trace
cat $out/via_trainer_step_gaps.txt
