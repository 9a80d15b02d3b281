#!/bin/bash
# tools/mini_collect2.sh <tag>: bench lines of the final build + C2 / C1 kernel traces -> gpurun_out/<tag>/
set -e -o pipefail
tag=${1:-r05g}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
run() { ( cd /tmp && timeout -k 10 300 rocprofv3 "$@" ) > $out/last.log 2>&1 || { tail -5 $out/last.log; exit 1; }; }
for cfg in C2 C1; do
  run --kernel-trace --stats --output-format csv -d $root/$out/trace_$cfg -- python3 $root/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-host-stream
  python3 tools/trace_summary.py $out/trace_$cfg 5 70 > $out/${cfg}_kernel_trace_summary.txt
  python3 tools/step_timeline.py $out/trace_$cfg > $out/${cfg}_step_timeline.txt
  cp $out/trace_$cfg/*/*_kernel_stats.csv $out/${cfg}_kernel_stats.csv
  rm -rf $out/trace_$cfg
done
python3 bench.py --steps 20 --warmup 3 > $out/bench_line_C3.json 2> $out/bench.err
python3 bench.py --config C2 --steps 10 > $out/bench_line_C2.json 2>> $out/bench.err
python3 bench.py --config C1 --steps 10 > $out/bench_line_C1.json 2>> $out/bench.err
python3 bench.py --via-trainer --steps 200 > $out/bench_line_via_trainer.json 2>> $out/bench.err
python3 bench.py --via-trainer --steps 200 --pcm-loader > $out/bench_line_via_trainer_pcm_loader.json 2>> $out/bench.err
python3 bench.py --via-trainer --dataloader-workers 6 --steps 96 > $out/bench_line_via_trainer_dataloader6.json 2>> $out/bench.err
python3 bench.py --via-trainer --dataloader-workers 6 --epoch-repeat 8 --steps 768 > $out/bench_line_via_trainer_dataloader6_long_epochs.json 2>> $out/bench.err
python3 - $out <<'P'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/bench_line_*.json')):
    d = json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], round(d['ms_per_step'], 4), round(d.get('pcie_inclusive', {}).get('ms_per_step', 0), 4), d.get('roofline', {}).get('frac'))
P
