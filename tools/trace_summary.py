#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of bench.py: per kernel name and per (kernel, grid) instance, per step.

  trace_summary.py <dir> <n_steps> [<n_instances>]

Only the LAST n_steps training steps are counted (a step starts at its STFT front-end launch), so one-off work --
parameter flattening, synthetic-data generation, eager warm-up -- does not leak into the per-step figures.  Also prints
the wall span of those steps (first kernel start to last kernel end) next to the sum of kernel durations: the
difference is launch gaps / dependency bubbles."""
import collections, csv, glob, re, sys
d, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
def short(n):
    n = n.replace('dam::(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'stft2048_kernel' in r['Kernel_Name'] or 'stft_generic_kernel' in r['Kernel_Name']]
if len(starts) > nsteps:
    # steps = [starts[-nsteps-1], starts[-1]): the last launch opens a step that is cut off by the end of the timed region
    lo, hi = starts[-nsteps - 1], starts[-1]
    rows = rows[lo:hi]
else:
    nsteps = max(1, len(starts))
agg, byname, tot = collections.OrderedDict(), collections.Counter(), 0.0
for r in rows:
    key = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    dur = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dur; tot += dur; byname[key[0]] += dur
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / nsteps / 1e6
print('--- by kernel (us per step, last %d steps)' % nsteps)
for n, dur in byname.most_common(24):
    print('%-46s %8.1f %5.1f%%' % (n[:46], dur / nsteps / 1e3, 100 * dur / tot))
if len(sys.argv) > 3:
    print('--- by instance')
    for k, (c, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3])]:
        print('%-40s wg %-7s y %-4s z %-3s calls/step %5.1f avg %7.1f us per-step %7.1f' % (k[0][:40], k[1], k[2], k[3], c / nsteps, dur / c / 1e3, dur / nsteps / 1e3))
print('kernels per step %.1f' % (len(rows) / nsteps))
print('sum of kernel durations per step ms %.3f' % (tot / nsteps / 1e6))
print('wall span per step ms %.3f  (gaps %.3f ms = %.1f us per kernel boundary)' % (span, span - tot / nsteps / 1e6, 1e3 * (span - tot / nsteps / 1e6) / (len(rows) / nsteps)))
