#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV: per kernel name and per (kernel, grid) instance, per step."""
import collections, csv, glob, re, sys
d, nsteps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
agg, byname, tot = collections.OrderedDict(), collections.Counter(), 0.0
def short(n):
    n = n.replace('dam::(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]
for r in csv.DictReader(open(f)):
    key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    dur = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dur; tot += dur; byname[key[0]] += dur
print('--- by kernel (us per step)')
for n, dur in byname.most_common(22):
    print('%-46s %8.1f %5.1f%%' % (n[:46], dur / nsteps / 1e3, 100 * dur / tot))
if len(sys.argv) > 3:
    print('--- by instance')
    for k, (c, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3])]:
        print('%-40s wg %-7s y %-4s z %-3s calls/step %5.1f avg %7.1f us per-step %7.1f' % (k[0][:40], k[1], k[2], k[3], c / nsteps, dur / c / 1e3, dur / nsteps / 1e3))
print('total per step ms %.3f' % (tot / nsteps / 1e6))
