#!/usr/bin/env python3
"""Which torch ops (not css_amd kernels) launch work inside one training step: counts of aten ops and of device memcpy / memset /
fill kernels, from torch.profiler.  Usage on the GPU box:  python3 scripts/prof_ops.py [workload] > gpurun_out/prof_ops.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
tr, batch, meta = bench.build(wl, dev, 0)
for _ in range(2):
    tr.step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(*batch)
    torch.cuda.synchronize()
ka = prof.key_averages()
print("== aten ops by count")
for e in sorted([e for e in ka if e.key.startswith("aten::")], key=lambda e: -e.count)[:40]:
    print(f"{e.key:50s} n={e.count:6d} cpu_total_ms={e.cpu_time_total / 1e3:8.2f} dev_total_ms={e.device_time_total / 1e3:8.2f}")
print("== device activities that are not css kernels")
for e in sorted([e for e in ka if e.device_time_total > 0 and not e.key.startswith("aten::")], key=lambda e: -e.count)[:60]:
    print(f"{e.key[:110]:110s} n={e.count:6d} dev_total_ms={e.device_time_total / 1e3:8.2f}")
