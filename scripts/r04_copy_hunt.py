#!/usr/bin/env python3
"""Who issues the ~460 small device copies of a step (profiles/r03_bench_kernel_stats.csv: __amd_rocclr_copyBuffer, 463 calls, 1.7 ms)?
One profiled c2 step under torch.profiler with Python stacks; prints the aten::copy_ / aten::to / aten::contiguous / aten::clone call
sites (innermost css_amd frame) with their counts."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
size = int(os.environ.get("HUNT_SIZE", "513"))
tr, batch, meta = bench.build("c2", dev, 0, size=size, batch=int(os.environ.get("HUNT_BATCH", "16")))
for _ in range(3):
    tr.step(*batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
# which C-ABI entry points does a step call, how often - and for the per-layer weight layout calls, for which weights
from css_amd import _lib, ops
import css_amd.loss.loss as L
import css_amd.functional as F_
import css_amd.train_step as TS
calls = collections.Counter()
wl = collections.Counter()
real_call = _lib.call
def spy(name, *a):
    calls[name] += 1
    if name == "css_weight_layout":
        wl[(tuple(a[0].shape), tuple(a[1].shape), str(a[1].dtype), a[6])] += 1
    return real_call(name, *a)
for m in (ops, L, F_, TS, _lib):
    if hasattr(m, "call"):
        m.call = spy
tr.step(*batch)
torch.cuda.synchronize()
for m in (ops, L, F_, TS, _lib):
    if hasattr(m, "call"):
        m.call = real_call
print("--- C-ABI calls of one step")
for k, v in calls.most_common():
    print(f"{v:5d}  {k}")
print("--- css_weight_layout per-layer calls (w shape, out shape, dtype, dgrad)")
for k, v in wl.most_common(40):
    print(f"{v:3d}  {k}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    tr.step(*batch)
    torch.cuda.synchronize()
sites = collections.Counter()
shapes = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::_foreach_add_"):
        frame = next((s for s in (ev.stack or []) if "css_amd" in s or "bench.py" in s), "(no css_amd frame)")
        sites[(ev.name, frame)] += 1
        shapes[(ev.name, frame)][""] += 1
for (name, frame), n in sites.most_common(40):
    print(f"{n:5d}  {name:18s} {frame}   {dict(shapes[(name, frame)].most_common(2))}")
print("--- device-side summary (top 25 by count)")
ka = prof.key_averages()
for e in sorted(ka, key=lambda e: -e.count)[:25]:
    print(f"{e.count:6d}  {e.key[:100]}  cuda_total {e.device_time_total / 1e3:.3f} ms")
