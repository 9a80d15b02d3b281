#!/bin/bash
# tools/r5b_tests.sh: host-path GPU tests + the reference-API lines after the upload changes -> gpurun_out/r5b/
set -e -o pipefail
out=gpurun_out/r5b; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_host_gpu.py tests/test_ingest_gpu.py tests/test_integration_stub_gpu.py -x -q -m gpu > $out/tests_host.log 2>&1 || { tail -40 $out/tests_host.log; exit 1; }
tail -3 $out/tests_host.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-host-stream > $out/bench_base.json 2> $out/bench.err
python3 bench.py --via-trainer --steps 200 > $out/bench_line_via_trainer.json 2>> $out/bench.err
python3 bench.py --via-trainer --steps 200 --pcm-loader > $out/bench_line_via_trainer_pcm_loader.json 2>> $out/bench.err
python3 bench.py --via-trainer --dataloader-workers 6 --epoch-repeat 8 --steps 768 > $out/bench_line_via_trainer_dataloader6_long_epochs.json 2>> $out/bench.err
python3 - <<'P'
import json
for f in ("bench_base", "bench_line_via_trainer", "bench_line_via_trainer_pcm_loader", "bench_line_via_trainer_dataloader6_long_epochs"):
    d = json.load(open("gpurun_out/r5b/%s.json" % f))
    print(f, round(d["ms_per_step"], 4), d["config"].get("host_ms_per_step"))
P
