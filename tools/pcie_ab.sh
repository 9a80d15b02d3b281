#!/bin/bash
# tools/pcie_ab.sh: the streamed (`pcie_inclusive`) leg of bench.py with and without the step mark that times its uploads
# (include/dam_hip.h, dam_step_mark_*), the copy's position in the step, and the data check -> gpurun_out/r5b/
set -e -o pipefail
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 120 python3 tools/ext_event_probe.py > $out/ext_event_probe.txt 2>&1 || { cat $out/ext_event_probe.txt; exit 1; }
cat $out/ext_event_probe.txt
for m in gate end; do
  timeout -k 10 200 python3 tools/copy_gate_probe.py $m 8 > $out/copy_gate_probe_$m.txt 2>&1 || { cat $out/copy_gate_probe_$m.txt; exit 1; }
  grep -a "equal\|period\|step  [3-6]" $out/copy_gate_probe_$m.txt
done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_mark.json 2> $out/bench_mark.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-copy-mark > $out/bench_nomark.json 2> $out/bench_nomark.err
python3 - <<'P'
import json
for f in ("mark", "nomark"):
    d = json.load(open("gpurun_out/r5b/bench_%s.json" % f))
    print(f, d["ms_per_step"], d["repeat"], d["pcie_inclusive"]["ms_per_step"])
P
