"""SyncBN statistics through peer-mapped device memory (kernel: css_amd/csrc/peer.hip).

``nn.SyncBatchNorm`` (what /root/reference/mix_label.py:76 converts every batch norm into) exchanges statistics in the forward and two sums in the
backward of EVERY layer: ~350 collectives per training step, each a host-issued RCCL call with a stream hand-over (measured: +4.9 ms per step
at one rank before any wire time, profiles/r03_force_coll_line.json).  Here every rank owns ONE exchange buffer - fine-grained device memory of the
library's own (css_peer_alloc) - that all ranks of the node map (hipIpc handles; on this image dmabuf IPC, HSA_ENABLE_IPC_MODE_LEGACY=0); an exchange is one
single-workgroup kernel per rank that publishes the local statistics, waits for the peers' flags with a bounded spin, adds the ranks' numbers in
rank order and - in forward - finalises mean / invstd / scale / shift in the same launch.  No RCCL call, no host round trip, bit-identical sums
on all ranks.

Opt-in and EXPERIMENTAL: ``CSS_SYNCBN=peer`` (default ``rccl``: the all-reduce path, the only one that has run on more than one process).  A group
that spans hosts, has more than 64 ranks, runs without dmabuf IPC or fails to map a buffer falls back to RCCL on every rank (one agreed verdict).  The single-process
parts - buffer layout, registration, the kernel in loop-back (world 1) and several "ranks" played in one process - are covered by
tests/test_dist_gpu.py; a node with >= 2 GPUs needs a RUN of tests/test_dist_gpu.py::test_peer_syncbn_two_processes, not a design.
"""
from __future__ import annotations

import ctypes
import os
import socket

import torch
import torch.distributed as dist

from ._lib import CssHipError, call, dev_stream, query

MAX_CHANNELS = 2048          # widest batch norm of the network (ResNet-101 layer4)
MAX_GROUPS = 2               # forward passes batched into one tensor (labeled + unlabeled)
MAX_WORLD = 64               # one polling lane per peer in the exchange kernel's first wave (peer.hip)
# Bound of the in-kernel wait for a peer's flag.  The order of RCCL's own watchdog (minutes), NOT of a step (5 s tripped on a
# healthy rank that was merely late - rank 0 writing a checkpoint, a data stall, a first-step module load); tests pass their own short value.
TIMEOUT_TICKS = int(float(os.environ.get("CSS_PEER_TIMEOUT_S", "600")) * 100e6)      # wall_clock64 runs at 100 MHz


class PeerUnavailable(RuntimeError):
    """The peer exchange cannot serve this process group (ranks on several hosts, too many ranks, legacy IPC mode, a failed mapping): the caller
    falls back to the RCCL all-reduce path - on EVERY rank, the verdict is agreed before anybody uses the exchange."""


class PeerExchange:
    """The exchange buffers of all ranks as seen from this rank, and the sequence number of the next exchange.

    The buffer is FINE-GRAINED device memory of this library's own (css_peer_alloc: hipExtMallocWithFlags), not a block of torch's caching
    allocator: peers poll its flags and read its payload while the kernel that wrote them is still running, and HIP promises that kind of
    cross-device visibility for fine-grained / uncached memory only.  Peers map it through hipIpc handles exchanged with
    ``all_gather_object`` - node-local by construction: the group must live on ONE host."""

    def __init__(self, device, group=None, slot_doubles=MAX_GROUPS * (2 * MAX_CHANNELS + 1), timeout_ticks=None):
        self.device = torch.device(device)
        self.slot_doubles = int(slot_doubles)
        self.timeout_ticks = TIMEOUT_TICKS if timeout_ticks is None else int(timeout_ticks)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.nbytes = query("css_peer_buffer_bytes", self.slot_doubles)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._own, self._mapped = None, []
        d = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dev = d
        why = None
        if self.world > MAX_WORLD:
            why = f"{self.world} ranks (the exchange kernel polls at most {MAX_WORLD} peers)"
        elif self.world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            why = "HSA_ENABLE_IPC_MODE_LEGACY=0 is not set (this driver only supports dmabuf IPC)"
        own, kind = ctypes.c_void_p(), ctypes.c_int(0)
        if why is None:
            try:
                call("css_peer_alloc", self.nbytes, d, ctypes.byref(own), ctypes.byref(kind))     # zero-filled: flags = "nothing published"
                self._own, self.mem_kind = own.value, {1: "fine-grained", 2: "uncached"}[kind.value]
            except CssHipError as e:
                why = f"css_peer_alloc failed ({e})"
        ptrs = [self._own]
        if self.world > 1:
            handle = (ctypes.c_ubyte * 64)()
            if why is None:
                try:
                    call("css_peer_ipc_export", self._own, d, handle)
                except CssHipError as e:
                    why = f"hipIpcGetMemHandle failed ({e})"
            infos = [None] * self.world
            dist.all_gather_object(infos, (socket.gethostname(), bytes(handle), why), group=group)
            if why is None:
                bad = [f"rank {r}: {w}" for r, (_, _, w) in enumerate(infos) if w]
                hosts = sorted({h for h, _, _ in infos})
                if bad:
                    why = "; ".join(bad)
                elif len(hosts) > 1:
                    why = f"the group spans {len(hosts)} hosts ({', '.join(hosts[:4])}): hipIpc mappings are node-local"
            ptrs = []
            if why is None:
                for r, (_, h, _) in enumerate(infos):
                    if r == self.rank:
                        ptrs.append(self._own)
                        continue
                    p = ctypes.c_void_p()
                    try:
                        call("css_peer_ipc_open", (ctypes.c_ubyte * 64).from_buffer_copy(h), d, ctypes.byref(p))
                    except CssHipError as e:
                        why = f"rank {self.rank}: hipIpcOpenMemHandle of rank {r}'s buffer failed ({e})"
                        break
                    self._mapped.append(p.value)
                    ptrs.append(p.value)
            # every rank has mapped every buffer before the first exchange - or NOBODY uses the exchange (one agreed verdict)
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, why, group=group)
            why = why or next((v for v in verdicts if v), None)
        if why is not None:
            self.close()
            raise PeerUnavailable(why)
        self.bases = torch.tensor(ptrs, dtype=torch.int64, device=self.device)
        self.seq = 0

    def close(self):
        for p in self._mapped:
            query("css_peer_ipc_close", p, self._dev)
        self._mapped = []
        if self._own:
            query("css_peer_free", self._own, self._dev)
            self._own = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def finalize(self, local_stats, G, C, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, count_out):
        """Forward exchange: local [G][2][C] sums + [G] local counts -> global statistics, finalised (css_bn_finalize's outputs)."""
        self.seq += 1
        dev, st = dev_stream(local_stats)
        call("css_bn_peer_finalize", self.bases, self.world, self.rank, self.seq, self.slot_doubles, local_stats, G, C, gamma, beta, running_mean,
             running_var, float(momentum), float(eps), mean, invstd, scale, shift, count_out, self.status, self.timeout_ticks, 0, dev, st)

    def gather(self, sums):
        """Backward exchange: ``sums`` (fp64, any length <= slot_doubles) becomes the sum over the ranks, in place."""
        self.seq += 1
        dev, st = dev_stream(sums)
        call("css_bn_peer_gather", self.bases, self.world, self.rank, self.seq, self.slot_doubles, sums, sums.numel(), sums, self.status,
             self.timeout_ticks, 0, dev, st)

    def check(self):
        """Host side, end of a step: raise if an exchange of this rank gave up waiting for a peer (its statistics were then local-only)."""
        s = int(self.status.item())
        if s:
            self.status.zero_()
            raise RuntimeError(f"SyncBN peer exchange {s} timed out waiting for a peer (CSS_PEER_TIMEOUT_S={self.timeout_ticks / 100e6:g}): "
                               "the statistics of that layer were incomplete - the step is invalid")


_exchange = None
_fallback = None              # why the peer path was refused for this process (then: RCCL for the rest of the run)


def enabled() -> bool:
    """CSS_SYNCBN=peer was asked for AND the exchange could be set up (or has not been tried yet).  EXPERIMENTAL: the two-process run
    (tests/test_dist_gpu.py::test_peer_syncbn_two_processes) has never executed on this pool; the default stays ``rccl``."""
    return os.environ.get("CSS_SYNCBN", "rccl") == "peer" and _fallback is None


def exchange(device):
    """The process-wide exchange (created on first use - by every rank at the same point: the first synchronised batch norm), or None when
    the group cannot use it (PeerUnavailable: agreed on every rank; the callers then take the RCCL path)."""
    global _exchange, _fallback
    if _fallback is not None:
        return None
    if _exchange is None or _exchange.device != torch.device(device):
        try:
            _exchange = PeerExchange(device)
        except PeerUnavailable as e:
            _fallback = str(e)
            import warnings
            warnings.warn(f"CSS_SYNCBN=peer is not available for this process group ({e}): SyncBN statistics go through RCCL")
            return None
    return _exchange


def reset():
    global _exchange, _fallback
    if _exchange is not None:
        _exchange.close()
    _exchange, _fallback = None, None
