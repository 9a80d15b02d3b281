#!/bin/bash
# tools/pmc_quick.sh <tag> <probe>: the two SQ counter sets only (instruction mix, wait breakdown) for one conv_probe.py probe
set -e -o pipefail
tag=$1; probe=$2
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/$out/pmc_${probe}_$i -- python3 $root/tools/conv_probe.py $probe 8 ) > $out/last.log 2>&1 || { tail -5 $out/last.log; }
done
python3 tools/pmc_summary.py $out | grep -v "pack_weights"
