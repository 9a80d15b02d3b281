"""Which mappings of a GPU-initialised torch process are ROCr host allocations that a fork would still copy (no VM_DONTCOPY)?
(diagnostic for staging.dontfork_pinned_host_memory / dam_host_dontfork_pinned)"""
import ctypes
import re
import sys

import torch

sys.path.insert(0, '.')
import deep_audio_mixer_amd  # noqa: F401,E402
from deep_audio_mixer_amd import staging  # noqa: E402

torch.cuda.init()
x = torch.zeros(1 << 20, device='cuda')
p = [torch.empty(48 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
y = torch.ones(1000).cuda()          # a pageable H2D copy
s = torch.cuda.Stream()
torch.cuda.synchronize()
print('guard marked', staging.dontfork_pinned_host_memory())
path = [ln.split()[-1] for ln in open('/proc/self/maps') if 'libhsa-runtime64' in ln][0]
hsa = ctypes.CDLL(path)


class Info(ctypes.Structure):
    _fields_ = [('size', ctypes.c_uint32), ('type', ctypes.c_int32), ('agentBase', ctypes.c_void_p), ('hostBase', ctypes.c_void_p),
                ('bytes', ctypes.c_size_t), ('userData', ctypes.c_void_p), ('owner', ctypes.c_uint64), ('flags', ctypes.c_uint8),
                ('registered', ctypes.c_uint8), ('pad', ctypes.c_uint8 * 46)]


hsa.hsa_amd_pointer_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(Info), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def info(addr):
    i = Info()
    i.size = ctypes.sizeof(Info)
    if hsa.hsa_amd_pointer_info(addr, ctypes.byref(i), None, None, None) != 0:
        return None
    return i


rows, cur = [], None
for line in open('/proc/self/smaps'):
    m = re.match(r'^([0-9a-f]+)-([0-9a-f]+) (\S+) \S+ \S+ (\d+)\s*(.*)$', line)
    if m:
        cur = dict(lo=int(m.group(1), 16), hi=int(m.group(2), 16), perms=m.group(3), ino=int(m.group(4)), name=m.group(5), flags='')
        rows.append(cur)
    elif line.startswith('VmFlags:'):
        cur['flags'] = line.split(':')[1].strip()
left = 0
for r in rows:
    if ' dc' in ' ' + r['flags'] or r['perms'] == '---p':
        continue
    step = max(4096, (r['hi'] - r['lo']) // 64 // 4096 * 4096)
    hits = []
    a = r['lo']
    while a < r['hi']:
        i = info(a)
        if i is not None and i.type != 0:
            hits.append((a, i.type, i.hostBase, i.bytes))
        a += step
    if hits:
        left += 1
        print('%x-%x %s ino %d name %r flags [%s]: %d probes known to ROCr, e.g. type %d base %s bytes %d' % (
            r['lo'], r['hi'], r['perms'], r['ino'], r['name'], r['flags'], len(hits), hits[0][1],
            hex(hits[0][2]) if hits[0][2] else None, hits[0][3]))
print('mappings ROCr knows that a fork would still copy:', left)
