#!/bin/bash
# tools/dl6_trace.sh: kernel + memory-copy trace of ModelTrainer.fit over DataLoader(num_workers=6, pin_memory=True) -- where do the
# 1.3 ms per step go that the device idles?  -> gpurun_out/r5b/dl6_*
set -e
out=gpurun_out/r5b; mkdir -p $out; export TMPDIR=/tmp; root=$(pwd)
( cd /tmp && timeout -k 10 280 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/trace_dl6 -- python3 $root/bench.py --via-trainer --dataloader-workers 6 --epoch-repeat 4 --steps 384 ) > $out/dl6_last.log 2>&1 || { tail -8 $out/dl6_last.log; exit 1; }
python3 tools/step_gaps.py $out/trace_dl6 > $out/dl6_step_gaps.txt
python3 - $out/trace_dl6 >> $out/dl6_step_gaps.txt <<'P'
import csv, glob, sys
d = sys.argv[1]
mc = glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)
cp = sorted(csv.DictReader(open(mc[0])), key=lambda r: int(r['Start_Timestamp']))
big = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction']) for r in cp if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 100000]
print('copies longer than 0.1 ms: %d' % len(big))
for s, e, dr in big[-12:]:
    print('  %s %.1f us' % (dr, (e - s) / 1e3))
kt = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
st = [int(r['Start_Timestamp']) for r in rows if 'stft2048' in r['Kernel_Name']]
# for the last steps: when did the step's upload end relative to the step's start, and when did the previous step end
ends = {}
prev_end = None
out = []
for i in range(len(st) - 10, len(st) - 1):
    s0 = st[i]
    last_copy = max((e for s, e, dr in big if e <= s0 + 50000), default=None)
    k_before = max((int(r['End_Timestamp']) for r in rows if int(r['End_Timestamp']) <= s0), default=None)
    out.append('  step at %d us: last big copy ended %.1f us before it, previous kernel ended %.1f us before it'
               % ((s0 - st[0]) // 1000, (s0 - last_copy) / 1e3 if last_copy else -1, (s0 - k_before) / 1e3 if k_before else -1))
print('\n'.join(out))
P
rm -rf $out/trace_dl6
cat $out/dl6_step_gaps.txt
