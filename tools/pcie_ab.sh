#!/bin/bash
# tools/pcie_ab.sh: the streamed (`pcie_inclusive`) leg of bench.py with and without the step mark that gates its uploads
# (include/dam_hip.h, dam_step_mark_*), and a kernel + memory-copy trace of the gated form -> gpurun_out/r5b/
set -e -o pipefail
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 120 python3 tools/ext_event_probe.py > $out/ext_event_probe.txt 2>&1 || { cat $out/ext_event_probe.txt; exit 1; }
cat $out/ext_event_probe.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_mark.json 2> $out/bench_mark.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-copy-mark > $out/bench_nomark.json 2> $out/bench_nomark.err
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/trace_pcie -- python3 $root/bench.py --steps 10 --warmup 1 --repeat 1 --no-cpu-baseline --no-roofline) > $out/pcie_trace.log 2>&1
python3 tools/pcie_trace.py $out/trace_pcie 8 > $out/pcie_trace_mark.txt
rm -rf $out/trace_pcie
cat $out/pcie_trace_mark.txt
python3 - <<'P'
import json
for f in ("mark", "nomark"):
    d = json.load(open("gpurun_out/r5b/bench_%s.json" % f))
    print(f, d["ms_per_step"], d["repeat"], d["pcie_inclusive"]["ms_per_step"])
P
