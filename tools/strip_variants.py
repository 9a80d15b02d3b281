#!/usr/bin/env python3
"""Times the C3-step launches of the strip convolution under every tools/libdam_<name>.so given on the command line
(diagnostic variants built by tools/build_variant.sh; results of such builds are wrong by construction)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import strip_ladder as L
rows = {'product': L.run_child({})}
for name in sys.argv[1:]:
    lib = os.path.join(L.ROOT, 'tools', 'libdam_%s.so' % name)
    rows[name] = L.run_child({'DAM_LIB_PATH': lib})
rows = {k: v for k, v in rows.items() if not isinstance(v, str) or print(k, 'FAILED', v)}
L.table(rows)
