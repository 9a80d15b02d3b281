#!/bin/bash
# (the TCP_* / TCC_* derived sets hung rocprofv3 on this pool for 7 minutes: not collected)
# PMC passes on the one-launch down-sampling kernels in isolation (tools/conv_probe.py probes), incl. HBM bytes
set -e -o pipefail
tag=${1:-r04_pmc_s2}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
for probe in ${2:-l2s2_pair l2s2_dgrad l3s2_dgrad l6s2_dgrad}; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    echo "[pmc] $probe set $i"
    ( cd /tmp && timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/$out/pmc_${probe}_$i -- python3 $root/tools/conv_probe.py $probe 8 ) > $out/last.log 2>&1 || { echo "set $i failed"; tail -3 $out/last.log; }
  done
  echo "[pmc] $probe done"
done
python3 tools/pmc_summary.py $out | grep -v "pack_weights" > $out/pmc_summary.csv
rm -rf $out/pmc_*/
wc -l $out/pmc_summary.csv
