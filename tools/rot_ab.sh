#!/bin/bash
# tools/rot_ab.sh: the device-side batch rotation (DAM_PCM_ROTATE) against the per-step re-pointing -> gpurun_out/r5b/
set -e -o pipefail
out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_host_gpu.py tests/test_features_gpu.py tests/test_ingest_gpu.py -x -q -m gpu > $out/tests_rot.log 2>&1 || { tail -40 $out/tests_rot.log; exit 1; }
tail -2 $out/tests_rot.log
for v in rot bind rot bind; do
  extra=""; [ $v = bind ] && extra="--bind-per-step"
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $extra > $out/bench_$v.json 2> $out/bench_$v.err
  python3 - $v <<'P'
import json, sys
d = json.load(open("gpurun_out/r5b/bench_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["ms_per_step"], 4), d["repeat"]["ms_per_step_median"], "pcie", round(d["pcie_inclusive"]["ms_per_step"], 4), "loss", d["config"]["final_loss"])
P
done
