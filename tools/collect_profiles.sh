#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun, from the repo root):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*  (copy the summaries into profiles/ afterwards)
# One --pmc set per run, only together with --kernel-trace (see the microarch guide's rocprofv3 section).
set -e -o pipefail
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
run() { ( cd /tmp && rocprofv3 "$@" ) > $out/last.log 2>&1 || { tail -5 $out/last.log; exit 1; }; }
run --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline
echo "[collect] bench trace done"
for probe in layer1 layer1_wgrad; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    run --pmc $set --kernel-trace --output-format csv -d $root/$out/pmc_${probe}_$i -- python3 $root/tools/conv_probe.py $probe 8
    echo "[collect] pmc $probe set $i done"
  done
done
python3 tools/trace_summary.py $out/trace 8 60 > $out/kernel_trace_summary.txt
python3 tools/pmc_summary.py $out > $out/pmc_summary.csv
python3 bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err
tail -c 1500 $out/bench_line.json
