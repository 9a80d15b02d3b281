#!/usr/bin/env python3
"""A/B of the strip convolution's two forms on the launches of a C3 step: self-overlapped (default) against ping-pong
(DAM_STRIP_PINGPONG=1), each in its own process; us per launch and clocks per MFMA per SIMD at 2.4 GHz."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import strip_ladder as L   # noqa: E402  (prints nothing on import when no variant library exists)
rows = {}
for name, env in (('self-overlapped', {}), ('ping-pong', {'DAM_STRIP_PINGPONG': '1'})):
    r = L.run_child(env)
    if isinstance(r, str):
        print(name, 'FAILED', r)
    else:
        rows[name] = r
L.table(rows)
