"""Anonymous private mappings (>= 8 KB, without VM_DONTCOPY) of a GPU-initialised torch process: run once as is and once
under HSA_USERPTR_FOR_PAGED_MEM=0 and diff the size histograms -- what disappears is user-pointer GPU memory.  (diagnostic)"""
import re
import sys
from collections import Counter

import torch

torch.cuda.init()
x = torch.zeros(1 << 20, device='cuda')
s = [torch.cuda.Stream() for _ in range(2)]
with torch.cuda.stream(s[0]):
    x.add_(1)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == 'guard':
    sys.path.insert(0, '.')
    import deep_audio_mixer_amd  # noqa: F401
    from deep_audio_mixer_amd import staging
    print('guard', staging.dontfork_pinned_host_memory())
rows, cur = [], None
for line in open('/proc/self/smaps'):
    m = re.match(r'^([0-9a-f]+)-([0-9a-f]+) (\S+) \S+ \S+ (\d+)\s*(.*)$', line)
    if m:
        cur = dict(lo=int(m.group(1), 16), hi=int(m.group(2), 16), perms=m.group(3), ino=int(m.group(4)), name=m.group(5), flags='', rss=0)
        rows.append(cur)
    elif line.startswith('VmFlags:'):
        cur['flags'] = line.split(':')[1].strip()
    elif line.startswith('Rss:'):
        cur['rss'] = int(line.split()[1])
c = Counter()
for r in rows:
    if r['ino'] == 0 and r['name'] == '' and r['perms'][3] == 'p' and r['perms'] != '---p' and (r['hi'] - r['lo']) >= 8192:
        c[((r['hi'] - r['lo']) >> 10, r['perms'], 'dc' if ' dc' in ' ' + r['flags'] else '--', 'rss' if r['rss'] else 'norss')] += 1
for k in sorted(c):
    print('%10d KB %s %s %s x%d' % (k + (c[k],)))
print('---- small anonymous non-dc mappings with neighbours')
for i, r in enumerate(rows):
    if r['ino'] == 0 and r['name'] == '' and r['perms'][3] == 'p' and r['perms'] != '---p' and 8192 <= (r['hi'] - r['lo']) <= (64 << 10) and ' dc' not in ' ' + r['flags']:
        for q in rows[max(0, i - 2):i + 3]:
            print('%s %x-%x %8d KB %s %-40s [%s]' % ('>>' if q is r else '  ', q['lo'], q['hi'], (q['hi'] - q['lo']) >> 10, q['perms'], q['name'][:40], q['flags']))
        print()
big = [q for q in rows if q['perms'] == '---p' and (q['hi'] - q['lo']) >= (1 << 30)]
print('PROT_NONE reservations >= 1 GB:', [(hex(q['lo']), (q['hi'] - q['lo']) >> 30) for q in big])
