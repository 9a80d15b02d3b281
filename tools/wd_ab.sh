#!/bin/bash
set -e -o pipefail
out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_conv_gpu.py tests/test_blocks_gpu.py tests/test_host_gpu.py -x -q -m gpu -k "wgrad or block or trainer or step" > $out/tests_wd.log 2>&1 || { tail -40 $out/tests_wd.log; exit 1; }
tail -2 $out/tests_wd.log
for v in on off on off; do
  if [ $v = off ]; then export DAM_WG_DIRECT_NO_BATCH=1; else unset DAM_WG_DIRECT_NO_BATCH; fi
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-host-stream > $out/bench_wd_$v.json 2> $out/bench_wd_$v.err
  python3 - $v <<'P'
import json, sys
d = json.load(open("gpurun_out/r5b/bench_wd_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["ms_per_step"], 4), d["repeat"]["ms_per_step_median"], "loss", d["config"]["final_loss"])
P
done
unset DAM_WG_DIRECT_NO_BATCH
bash tools/timeline_now.sh wd > /dev/null && tail -1 gpurun_out/r5/wd_step_timeline.txt && grep -n "wgrad_direct\|reduce_batch" gpurun_out/r5/wd_step_timeline.txt
