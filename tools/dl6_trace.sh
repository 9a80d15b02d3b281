#!/bin/bash
# tools/dl6_trace.sh: KERNEL trace (no memory-copy domain: that one hangs the tool with forked workers, profiles/README.md) of
# ModelTrainer.fit over DataLoader(num_workers=6, pin_memory=True): per-step period, kernel time, largest gaps -> gpurun_out/r5b/dl6_*
set -e
out=gpurun_out/r5b; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
( cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace_dl6 -- python3 $root/bench.py --via-trainer --dataloader-workers 6 --epoch-repeat 4 --steps 192 ) > $out/dl6_last.log 2>&1 || { tail -8 $out/dl6_last.log; exit 1; }
python3 tools/step_gaps.py $out/trace_dl6 > $out/dl6_step_gaps.txt
python3 - $out/trace_dl6 >> $out/dl6_step_gaps.txt <<'P'
import csv, glob, sys
kt = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
st = [i for i, r in enumerate(rows) if 'stft2048' in r['Kernel_Name']]
a, b = st[-4], st[-3]
t0 = int(rows[a]['Start_Timestamp'])
print('--- one steady step: every kernel that is not part of the captured step chain, and the step\'s first / last kernels')
for i in range(max(0, a - 6), b + 3):
    r = rows[i]
    n = r['Kernel_Name']
    if i < a + 3 or i > b - 6 or 'copyBuffer' in n or 'Fill' in n or 'elementwise' in n:
        print('  %9.1f .. %9.1f  (%7.1f us)  q%s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
              (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), n[:70]))
P
rm -rf $out/trace_dl6
cat $out/dl6_step_gaps.txt
