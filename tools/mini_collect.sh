set -e
out=gpurun_out/${1:-r04m}; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace_C3 -- python3 $root/bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-host-stream ) > $out/last.log 2>&1
python3 tools/trace_summary.py $out/trace_C3 5 90 > $out/C3_kernel_trace_summary.txt
python3 tools/step_timeline.py $out/trace_C3 > $out/C3_step_timeline.txt
cp $out/trace_C3/*/*_kernel_stats.csv $out/C3_kernel_stats.csv
rm -rf $out/trace_C3
python3 bench.py --steps 20 --warmup 3 > $out/bench_line_C3.json 2> $out/bench.err
tail -3 $out/C3_kernel_trace_summary.txt
python3 -c "
import json; d=json.load(open('$out/bench_line_C3.json')); print(d['value'], d['ms_per_step'], d['repeat']['ms_per_step_median'], d['roofline']['frac'])"
