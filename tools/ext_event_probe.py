#!/usr/bin/env python3
"""tools/ext_event_probe.py: does an EXTERNAL event recorded inside a captured hipGraph (hipEventRecordExternal through dam_step_mark_*; torch.cuda.Event(
external=True) is refused by the ROCm build of torch: "External events are disallowed in rocm") order work on ANOTHER stream against the middle of a graph replay?  The streamed training leg wants its host-to-device
copy of batch k+1 to start in the middle of step k (behind the forward pass's latency-bound launches), without cutting the graph.

Graph: A = long kernel (writes flag 1), [external record], B = long kernel (writes flag 2).  Side stream: wait(event) then reads the
flags into `seen`.  Expected per replay: seen == 1 ... (A done, B not yet) if the wait tracks the in-graph record; 2 would mean the wait
only resolved at the end of the graph (or the event completed late), 0 that it did not wait at all."""
import sys
import time
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import deep_audio_mixer_amd  # noqa: E402,F401
from deep_audio_mixer_amd import staging  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
n = 1 << 28
a = torch.zeros(n, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
seen = torch.zeros(16, dtype=torch.int32, device=dev)
ev = staging.StepMark()


def body():
    a.add_(1.0)
    a.mul_(1.0)
    flag.fill_(1)
    ev.record()
    for _ in range(6):
        a.add_(1.0)
    flag.fill_(2)


s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
t0 = time.time()
for k in range(8):
    flag.zero_()
    g.replay()
    with torch.cuda.stream(side):
        ev.wait(side)
        seen[k:k + 1].copy_(flag)
    torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print('seen per replay (1 = the side stream ran between A and B):', seen[:8].tolist(), 'wall %.3f s' % (time.time() - t0))
# the FIRST replay of a fresh graph may hold the waiter until the whole graph is done (2: late, the safe side; seen on this stack);
# 0 anywhere = the wait did not hold, and from the second replay on the waiter must run between A and B
vals = seen[:8].tolist()
sys.exit(0 if vals[0] in (1, 2) and all(v == 1 for v in vals[1:]) else 1)
