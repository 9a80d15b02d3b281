#!/usr/bin/env python3
"""tools/pcie_trace.py <rocprofv3 csv dir> [steps]: where the host-streamed (`pcie_inclusive`) region of bench.py loses against the
HBM-resident region.  Reads the kernel trace and the memory-copy trace of ONE bench.py run (resident regions first, the streamed region
last) and prints, for the last `steps` steps of each kind: the step period (STFT start to STFT start), the kernel time inside a step,
the kernels that got slower, what copy kernels (blit) ran on the device, and every host-to-device copy's interval against the steps."""
import csv, glob, re, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def short(n):
    n = n.replace('dam::(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([\w:]+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]


kf = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(kf)), key=lambda r: int(r['Start_Timestamp']))
cf = glob.glob(d + '/**/*_memory_copy_trace.csv', recursive=True)
copies = sorted(csv.DictReader(open(cf[0])), key=lambda r: int(r['Start_Timestamp'])) if cf else []
big = [c for c in copies if 'HOST_TO_DEVICE' in c.get('Direction', '') and
       int(c['End_Timestamp']) - int(c['Start_Timestamp']) > 200000]
print('copies in trace: %d, host-to-device longer than 0.2 ms: %d' % (len(copies), len(big)))
starts = [i for i, r in enumerate(rows) if 'stft2048_kernel' in r['Kernel_Name']]
if not big:
    print('no long host-to-device copy found: was the streamed region on?')
t_first_big = int(big[0]['Start_Timestamp']) if big else 1 << 62
res = [i for i in starts if int(rows[i]['Start_Timestamp']) < t_first_big]
stre = [i for i in starts if int(rows[i]['Start_Timestamp']) >= t_first_big]


def region(idx, label):
    idx = idx[-(steps + 1):]
    per, ktime, by = [], [], {}
    for a, b in zip(idx[:-1], idx[1:]):
        per.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
        kt = 0.0
        for r in rows[a:b]:
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            kt += dur
            k = short(r['Kernel_Name'])
            by[k] = by.get(k, 0.0) + dur / (len(idx) - 1)
        ktime.append(kt)
    print('%s: %d steps, period median %.1f us (min %.1f max %.1f), kernel time median %.1f us, kernels per step %d'
          % (label, len(per), sorted(per)[len(per) // 2], min(per), max(per), sorted(ktime)[len(ktime) // 2], idx[1] - idx[0]))
    return by, idx


by_r, _ = region(res, 'resident')
by_s, idx_s = region(stre, 'streamed')
print('--- kernels by per-step time, streamed - resident (us), |delta| > 1')
for k in sorted(set(by_r) | set(by_s), key=lambda k: -(by_s.get(k, 0) - by_r.get(k, 0))):
    dlt = by_s.get(k, 0) - by_r.get(k, 0)
    if abs(dlt) > 1:
        print('  %-60s %8.1f -> %8.1f  %+7.1f' % (k, by_r.get(k, 0), by_s.get(k, 0), dlt))
print('--- host-to-device copies against the streamed steps (us from the first listed step)')
if len(idx_s) > 1:
    t0 = int(rows[idx_s[0]]['Start_Timestamp'])
    for i in idx_s:
        print('  step start %9.1f' % ((int(rows[i]['Start_Timestamp']) - t0) / 1e3))
    for c in big:
        s, e = int(c['Start_Timestamp']), int(c['End_Timestamp'])
        if s >= t0:
            print('  copy %9.1f .. %9.1f  (%.1f us)' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))

# where in the step does each copy start?  (kernel running at that time, index within its step)
if len(idx_s) > 1 and big:
    print('--- kernel running when each copy starts / ends')
    for c in big:
        s0, e0 = int(c['Start_Timestamp']), int(c['End_Timestamp'])
        for a, b in zip(idx_s[:-1], idx_s[1:]):
            if int(rows[a]['Start_Timestamp']) <= s0 < int(rows[b]['Start_Timestamp']):
                for j in range(a, b):
                    if int(rows[j]['Start_Timestamp']) <= s0 and (j + 1 == b or int(rows[j + 1]['Start_Timestamp']) > s0):
                        print('  copy starts in kernel %3d of its step (%s, started %.1f us earlier)'
                              % (j - a, short(rows[j]['Kernel_Name']), (s0 - int(rows[j]['Start_Timestamp'])) / 1e3))
    a, b = idx_s[-2], idx_s[-1]
    print('--- gaps > 3 us between consecutive kernels of the last streamed step')
    for j in range(a + 1, b):
        gap = (int(rows[j]['Start_Timestamp']) - int(rows[j - 1]['End_Timestamp'])) / 1e3
        if gap > 3:
            print('  before kernel %3d (%s): %.1f us' % (j - a, short(rows[j]['Kernel_Name']), gap))
