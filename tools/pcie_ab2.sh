#!/bin/bash
# tools/pcie_ab2.sh: where the gated upload really starts, for two positions of the step mark
set -e -o pipefail
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/r5b; mkdir -p $out
for at in boundary forward; do
(cd /tmp && DAM_COPY_MARK_AT=$at timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/trace_pcie_$at -- python3 $root/bench.py --steps 10 --warmup 1 --repeat 1 --no-cpu-baseline --no-roofline) > $out/pcie_trace_$at.log 2>&1
python3 tools/pcie_trace.py $out/trace_pcie_$at 8 > $out/pcie_trace_mark_$at.txt
rm -rf $out/trace_pcie_$at
tail -25 $out/pcie_trace_mark_$at.txt
done
