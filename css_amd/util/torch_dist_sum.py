"""torch_dist_sum (generalframeworks/util/torch_dist_sum.py:6-20): SUM all-reduce of each argument, asynchronously, then wait.
With no process group (single GPU) the clones are returned unchanged."""
import torch
import torch.distributed as dist


def torch_dist_sum(gpu, *args):
    outs, pending = [], []
    live = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    for arg in args:
        t = arg.clone().detach()
        outs.append(t)
        if live:
            pending.append(dist.all_reduce(t, async_op=True))
    for p in pending:
        p.wait()
    return outs
