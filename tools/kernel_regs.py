#!/usr/bin/env python3
"""tools/kernel_regs.py <file.s> [filter]: VGPR / SGPR / scratch / LDS of every kernel in a hipcc -save-temps assembly file."""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    g = lambda k: re.search(k + r' (\d+)', body).group(1)
    print('%-70s vgpr %s sgpr %s scratch %s' % (dem[dem.find('dam::'):dem.find('>') + 1][:70], g('next_free_vgpr'), g('next_free_sgpr'), g('private_segment_fixed_size')))
