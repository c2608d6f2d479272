"""Backward on a partitioned chip.

The student's backward (/root/reference/mix_label.py:193: ``total_loss.backward()``) is two chains that share no data until the optimizer: the
critical chain - batch-norm backward (HBM-bound: two passes over the layer tensor, no MFMA) and the data gradients (MFMA-bound) - and the weight
gradients (MFMA-bound, ~2-3 TB/s of HBM traffic).  On ONE stream they take turns on the whole chip; on two plain streams they still take turns
(the persistent kernels hold every CU: profiles/r04_wgrad_stream_ab.txt).  With two CU-MASKED streams - ``main_cus`` CUs for batch norm + data
gradients, the rest for the weight gradients - the memory-bound and the matrix-bound kernels run at the same time: one layer-3 Bottleneck backward
1209 -> 1106 us at 192 + 64 CUs (scripts/partition_bench.hip, profiles/r05_partition_bench.txt).

How it is wired without touching the autograd engine's stream rules (every backward node runs on the stream its forward ran on):
``ops.on_backward_stream`` wraps every custom ``backward``: inside a partition window the body runs with the main stream current (torch
allocations, fills and the css_* launches all follow ``torch.cuda.current_stream``).  The loss nodes first make the main stream wait for the engine's
stream (their incoming gradient comes from torch's scalar arithmetic); a node that hands a gradient to a torch-native consumer makes the engine's stream
wait for the main stream; between two nodes of this package nothing is needed.  The first window walks the graph: a torch-native kernel node between
two of ours switches to the strict mode (both waits around EVERY node: +40 ms of host time per step when it was the only mode - the first A/B of
profiles/r05_partition_ab.txt).  ``_Conv2d.backward`` sends its weight gradient to the side stream behind an event; the operands stay referenced until the side
stream has passed them (no ``record_stream``: the allocator never sees a cross-stream free).  The window ends with the engine's stream waiting for both.

Results are independent of the partition: the data-gradient kernels compute the same tiles on a smaller grid, the weight-gradient slice plan is made
for the whole device (css_wgrad_plan_), every reduction keeps its order - ``tests/test_partition_gpu.py`` compares a partitioned step with an
unpartitioned one bit for bit.  Off when collectives are on (world > 1): the gradient buckets are launched from the backward chain and would have
to wait for the side stream; a multi-GPU run is what would say whether that pays.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes
import os

import torch

from ._lib import call, query

_parts = {}
_active = None                      # the partition whose window is open on this thread of control (backward of one trainer step)


def requested() -> int:
    """CUs of the main partition (CSS_BWD_PARTITION; 0 = off)."""
    return int(os.environ.get("CSS_BWD_PARTITION", "0"))


class BwdPartition:
    def __init__(self, device, main_cus):
        dev = torch.device(device)
        d = dev.index if dev.index is not None else torch.cuda.current_device()
        total = query("css_device_cu_count", d)
        if not (0 < main_cus < total) or main_cus % 8 or total % 8:
            raise ValueError(f"main partition of {main_cus} CUs on a device with {total}")
        ptrs = []
        for first, n in ((0, main_cus), (main_cus, total - main_cus)):
            p = ctypes.c_void_p()
            call("css_stream_create_masked", d, first, n, ctypes.byref(p))
            ptrs.append(p.value)
        self.device, self.main_cus, self.side_cus = dev, main_cus, total - main_cus
        self.main = torch.cuda.ExternalStream(ptrs[0], device=dev)
        self.side = torch.cuda.ExternalStream(ptrs[1], device=dev)
        self._held = []                               # tensors the side stream's kernels read: referenced until the window closes
        self._events, self._ev_next = [], 0           # reusable events (created once: an Event() per convolution showed up in the host's time)
        # strict: both-way synchronisation around EVERY node.  None = not decided yet: the first window walks the graph (ops.graph_allows_light_partition)
        self.strict = True if os.environ.get("CSS_BWD_PARTITION_STRICT") == "1" else None

    # ---- the window: one backward pass ----
    @contextlib.contextmanager
    def window(self, root=None):
        global _active
        if _active is not None:
            raise RuntimeError("nested backward partition windows")
        if self.strict is None:
            from . import ops
            self.strict = not (root is not None and ops.graph_allows_light_partition(root))
        cur = torch.cuda.current_stream(self.device)
        self.main.wait_stream(cur)
        self._ev_next = 0
        _active = self
        try:
            yield self
        finally:
            _active = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.main)
            cur.wait_stream(self.side)
            self._held.clear()                        # every later use of those blocks is ordered behind the two waits above

    def event(self):
        """A reusable event of this window (re-recording an event whose earlier waits were already queued is fine: a wait binds to the record
        that precedes it in host order)."""
        if self._ev_next == len(self._events):
            self._events.append(torch.cuda.Event())
        ev = self._events[self._ev_next]
        self._ev_next += 1
        return ev

    def hold(self, *tensors):
        """Keep ``tensors`` alive until the window closes (the side stream reads them at a time of its own; no ``record_stream``, no allocator
        events: at the bench's sizes this keeps ~17 GB of incoming gradients alive for the length of a backward pass - of 288)."""
        self._held.append(tensors)


def active():
    return _active


def get(device):
    """The process-wide partition of ``device`` (streams are created once and live as long as the process), or None when switched off."""
    n = requested()
    if n <= 0:
        return None
    key = (torch.device(device).index, n)
    if key not in _parts:
        _parts[key] = BwdPartition(device, n)
    return _parts[key]
