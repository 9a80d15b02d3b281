#!/usr/bin/env python3
"""tools/step_timeline.py <rocprofv3 csv dir> : the kernels of the LAST complete training step in launch order -- index, start offset,
duration, gap to the previous kernel, workgroups, name -- for reading the serial chain of a step (which launches are short, where the
deep stages sit)."""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'stft2048_kernel' in r['Kernel_Name'] or 'stft_generic_kernel' in r['Kernel_Name']]
lo, hi = starts[-2], starts[-1]
def short(n):
    n = n.replace('dam::(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]
t0, prev, acc = int(rows[lo]['Start_Timestamp']), None, 0.0
for k, r in enumerate(rows[lo:hi]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    prev = e
    acc += (e - s) / 1e3
    print('%3d  t %7.1f  dur %6.1f  gap %5.1f  wg %5d y %-3s z %-2s %s' % (k, (s - t0) / 1e3, (e - s) / 1e3, gap, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
                                                                    r['Grid_Size_Y'], r['Grid_Size_Z'], short(r['Kernel_Name'])))
print('kernels %d, kernel time %.1f us, span %.1f us' % (hi - lo, acc, (prev - t0) / 1e3))
