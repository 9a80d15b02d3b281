#!/bin/bash
# PMC passes on single kernels of the hot path (run on the GPU box through gpurun, from the repo root):
#   tools/pmc_probe.sh <tag> <probe> [<probe> ...]   -> gpurun_out/<tag>/pmc_<probe>_<set>/ + gpurun_out/<tag>/pmc_summary.csv
# One --pmc set per run, only together with --kernel-trace (the microarch guide's rocprofv3 section: FETCH_SIZE and
# WRITE_SIZE in separate passes; no other trace domain next to --pmc).
set -e -o pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
root=$(pwd)
run() { ( cd /tmp && rocprofv3 "$@" ) > $out/last.log 2>&1 || { tail -5 $out/last.log; exit 1; }; }
for probe in "$@"; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    run --pmc $set --kernel-trace --output-format csv -d $root/$out/pmc_${probe}_$i -- python3 $root/tools/conv_probe.py $probe 8
    echo "[pmc] $probe set $i done"
  done
done
python3 tools/pmc_summary.py $out > $out/pmc_summary.csv
cat $out/pmc_summary.csv
