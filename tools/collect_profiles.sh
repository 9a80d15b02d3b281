#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun, from the repo root):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/  (copy the summaries into profiles/ afterwards)
# Kernel traces of bench.py for every BASELINE config (steady-state steps only, tools/trace_summary.py), then PMC passes on
# the dominant kernels in isolation: one --pmc set per run, only together with --kernel-trace (microarch guide: FETCH_SIZE
# and WRITE_SIZE in separate passes, no other trace domain next to --pmc).
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
run() { ( cd /tmp && timeout -k 10 300 rocprofv3 "$@" ) > $out/last.log 2>&1 || { tail -5 $out/last.log; exit 1; }; }   # (the guard: profiles/README.md, the counter-set hang)
for cfg in C3 C2 C1; do
  run --kernel-trace --stats --output-format csv -d $root/$out/trace_$cfg -- python3 $root/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-host-stream
  python3 tools/trace_summary.py $out/trace_$cfg 5 70 > $out/${cfg}_kernel_trace_summary.txt
  python3 tools/step_timeline.py $out/trace_$cfg > $out/${cfg}_step_timeline.txt
  cp $out/trace_$cfg/*/*_kernel_stats.csv $out/${cfg}_kernel_stats.csv
  echo "[collect] $cfg trace done"
done
run --kernel-trace --stats --output-format csv -d $root/$out/trace_C5 -- python3 $root/bench.py --config C5 --steps 3 --warmup 1 --no-roofline
cp $out/trace_C5/*/*_kernel_stats.csv $out/C5_kernel_stats.csv
echo "[collect] C5 trace done"
for probe in layer1 layer1_dgrad_epi1 layer1_dgrad_epi2 layer1_dgrad_epi3 layer1_wgrad layer3 layer6 layer6_wgrad stft; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    run --pmc $set --kernel-trace --output-format csv -d $root/$out/pmc_${probe}_$i -- python3 $root/tools/conv_probe.py $probe 8
  done
  echo "[collect] pmc $probe done"
done
python3 tools/pmc_summary.py $out > $out/pmc_summary.csv
python3 bench.py --steps 20 --warmup 3 > $out/bench_line_C3.json 2> $out/bench.err
python3 bench.py --config C5 --steps 10 > $out/bench_line_C5.json 2>> $out/bench.err
python3 bench.py --config C2 --steps 10 > $out/bench_line_C2.json 2>> $out/bench.err
python3 bench.py --config C1 --steps 10 > $out/bench_line_C1.json 2>> $out/bench.err
python3 bench.py --via-trainer --steps 200 > $out/bench_line_via_trainer.json 2>> $out/bench.err
python3 bench.py --via-trainer --steps 200 --pcm-loader > $out/bench_line_via_trainer_pcm_loader.json 2>> $out/bench.err
python3 bench.py --ingest > $out/bench_line_ingest.json 2>> $out/bench.err
python3 bench.py --via-trainer --dataloader-workers 6 --steps 96 > $out/bench_line_via_trainer_dataloader6.json 2>> $out/bench.err
python3 bench.py --via-trainer --dataloader-workers 6 --epoch-repeat 8 --steps 768 > $out/bench_line_via_trainer_dataloader6_long_epochs.json 2>> $out/bench.err
python3 bench.py --no-graph --steps 20 --no-cpu-baseline --no-roofline --no-host-stream > $out/bench_line_C3_eager_no_graph.json 2>> $out/bench.err
DAM_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 10 --warmup 2 --breakdown --no-roofline --no-host-stream 2>> $out/bench.err | grep '^{' > $out/bench_line_ddp2_gloo_rehearsal.json
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-copy-mark > $out/bench_line_C3_no_copy_mark.json 2>> $out/bench.err
tail -c 600 $out/bench_line_C3.json
