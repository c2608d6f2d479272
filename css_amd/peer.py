"""SyncBN statistics through peer-mapped device memory (kernel: css_amd/csrc/peer.hip).

``nn.SyncBatchNorm`` (what /root/reference/mix_label.py:76 converts every batch norm into) exchanges statistics in the forward and two sums in the
backward of EVERY layer: ~350 collectives per training step, each a host-issued RCCL call with a stream hand-over (measured: +4.9 ms per step
at one rank before any wire time, profiles/r03_force_coll_line.json).  Here every rank owns ONE exchange buffer that all ranks of the node map
(``hipIpc`` through torch's CUDA-IPC storage sharing - on this image dmabuf IPC, HSA_ENABLE_IPC_MODE_LEGACY=0); an exchange is one
single-workgroup kernel per rank that publishes the local statistics, waits for the peers' flags with a bounded spin, adds the ranks' numbers in
rank order and - in forward - finalises mean / invstd / scale / shift in the same launch.  No RCCL call, no host round trip, bit-identical sums
on all ranks.

Opt-in: ``CSS_SYNCBN=peer`` (default ``rccl``: the all-reduce path, the only one that has run on more than one process).  The single-process
parts - buffer layout, registration, the kernel in loop-back (world 1) and several "ranks" played in one process - are covered by
tests/test_dist_gpu.py; a node with >= 2 GPUs needs a RUN of tests/test_dist_gpu.py::test_peer_syncbn_two_processes, not a design.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from ._lib import call, dev_stream, query

MAX_CHANNELS = 2048          # widest batch norm of the network (ResNet-101 layer4)
MAX_GROUPS = 2               # forward passes batched into one tensor (labeled + unlabeled)
TIMEOUT_TICKS = int(float(os.environ.get("CSS_PEER_TIMEOUT_S", "5")) * 100e6)      # wall_clock64 runs at 100 MHz


class PeerExchange:
    """The exchange buffers of all ranks as seen from this rank, and the sequence number of the next exchange."""

    def __init__(self, device, group=None, slot_doubles=MAX_GROUPS * (2 * MAX_CHANNELS + 1)):
        self.device = torch.device(device)
        self.slot_doubles = int(slot_doubles)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        nbytes = query("css_peer_buffer_bytes", self.slot_doubles)
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)          # (flags start at 0 = "nothing published")
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._peers = [self.buf]                     # keeps the mapped storages alive
        if self.world > 1:
            torch.cuda.synchronize(self.device)      # the zero-fill is done before anybody maps the buffer
            handle = self.buf.untyped_storage()._share_cuda_()
            handles = [None] * self.world
            dist.all_gather_object(handles, handle, group=group)
            self._peers = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    self._peers.append(self.buf)
                else:
                    st = torch.UntypedStorage._new_shared_cuda(*h)
                    self._peers.append(torch.empty(0, dtype=torch.uint8, device=self.device).set_(st, 0, (nbytes,)))
            dist.barrier(group=group)                # every rank has mapped every buffer before the first exchange
        self.bases = torch.tensor([t.data_ptr() for t in self._peers], dtype=torch.int64, device=self.device)
        self.seq = 0

    def finalize(self, local_stats, G, C, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, count_out):
        """Forward exchange: local [G][2][C] sums + [G] local counts -> global statistics, finalised (css_bn_finalize's outputs)."""
        self.seq += 1
        dev, st = dev_stream(local_stats)
        call("css_bn_peer_finalize", self.bases, self.world, self.rank, self.seq, self.slot_doubles, local_stats, G, C, gamma, beta, running_mean,
             running_var, float(momentum), float(eps), mean, invstd, scale, shift, count_out, self.status, TIMEOUT_TICKS, 0, dev, st)

    def gather(self, sums):
        """Backward exchange: ``sums`` (fp64, any length <= slot_doubles) becomes the sum over the ranks, in place."""
        self.seq += 1
        dev, st = dev_stream(sums)
        call("css_bn_peer_gather", self.bases, self.world, self.rank, self.seq, self.slot_doubles, sums, sums.numel(), sums, self.status,
             TIMEOUT_TICKS, 0, dev, st)

    def check(self):
        """Host side, end of a step: raise if an exchange of this rank gave up waiting for a peer (its statistics were then local-only)."""
        s = int(self.status.item())
        if s:
            self.status.zero_()
            raise RuntimeError(f"SyncBN peer exchange {s} timed out waiting for a peer (CSS_PEER_TIMEOUT_S={TIMEOUT_TICKS / 100e6:g}): "
                               "the statistics of that layer were incomplete - the step is invalid")


_exchange = None


def enabled() -> bool:
    return os.environ.get("CSS_SYNCBN", "rccl") == "peer"


def exchange(device) -> PeerExchange:
    """The process-wide exchange (created on first use - by every rank at the same point: the first synchronised batch norm)."""
    global _exchange
    if _exchange is None or _exchange.device != torch.device(device):
        _exchange = PeerExchange(device)
    return _exchange


def reset():
    global _exchange
    _exchange = None
