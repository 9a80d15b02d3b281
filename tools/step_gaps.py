#!/usr/bin/env python3
"""tools/step_gaps.py <rocprofv3 out dir> : per training step (from one stft2048 launch to the next) the wall period, the sum of kernel
durations and the largest idle gaps between consecutive kernels, plus the memory copies that overlap -- where a loop that feeds the
captured step loses time against the step itself."""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
starts = [i for i, k in enumerate(ks) if 'stft2048' in k[2] or 'stft_generic' in k[2]]
per, busy, gaps = [], [], []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ks[a:b]
    per.append((ks[b][0] - seg[0][0]) / 1e3)
    busy.append(sum(e - s for s, e, _ in seg) / 1e3)
    g = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)] + [(ks[b][0] - seg[-1][1]) / 1e3]
    big = sorted(((x, seg[i][2][:40]) for i, x in enumerate(g)), reverse=True)[:3]
    gaps.append(big)
n = len(per)
mid = sorted(per)[n // 2]
print('steps %d: period median %.1f us, min %.1f, max %.1f; kernel time median %.1f us' % (n, mid, min(per), max(per), sorted(busy)[n // 2]))
for i in range(max(0, n - 6), n):
    print('  step %d: period %.1f busy %.1f  largest gaps: %s' % (i, per[i], busy[i], ', '.join('%.0f us after %s' % g for g in gaps[i])))
mc = glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)
if mc:
    cp = list(csv.DictReader(open(mc[0])))
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in cp]
    print('memory copies: %d, total %.1f ms, longest %.1f us' % (len(cp), sum(dur) / 1e3, max(dur) if dur else 0))
