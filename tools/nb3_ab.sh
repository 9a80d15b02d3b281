#!/bin/bash
# tools/nb3_ab.sh: three-block output tiles of the tile convolution (48-channel layers of the scalar models) A/B -> gpurun_out/r5b/
set -e -o pipefail
out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_conv_gpu.py tests/test_models_gpu.py tests/test_blocks_gpu.py -x -q -m gpu > $out/tests_nb3.log 2>&1 || { tail -40 $out/tests_nb3.log; exit 1; }
tail -2 $out/tests_nb3.log
for cfg in C2 C1; do
for v in on off on off; do
  if [ $v = off ]; then export DAM_CONV_NO_NB3=1; else unset DAM_CONV_NO_NB3; fi
  timeout -k 10 300 python3 bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-host-stream > $out/bench_nb3_$v.json 2> $out/bench_nb3_$v.err
  python3 - $v $cfg <<'P'
import json, sys
d = json.load(open("gpurun_out/r5b/bench_nb3_%s.json" % sys.argv[1]))
print(sys.argv[2], sys.argv[1], round(d["ms_per_step"], 4), d["repeat"]["ms_per_step_median"], "loss", d["config"]["final_loss"])
P
done
done
