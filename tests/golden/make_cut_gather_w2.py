#!/usr/bin/env python3
"""Golden vectors for the in-step mixing under TWO ranks, captured from the reference itself: generate_cut_gather_2 / _3
(/root/reference/generalframeworks/dataset_helpers/VOC.py:393-477) run by two gloo processes on CPU tensors, numpy seeded per rank.

Runs only in the build container (needs /root/reference); writes cut_gather_w2.npz (inputs + the slice each rank got back).  Shims as
in make_golden.py: stub ``torchvision`` module names (VOC.py imports them at the top; the mixing functions do not use them).

    python tests/golden/make_cut_gather_w2.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
B, H, W = 3, 20, 24


def inputs(rank):
    g = torch.Generator().manual_seed(900 + rank)
    return (torch.randn(B, 3, H, W, generator=g), torch.randint(0, 5, (B, H, W), generator=g), torch.rand(B, H, W, generator=g),
            torch.rand(B, H, W, generator=g))


def worker(rank, world, port, q):
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.path.insert(0, REF)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from generalframeworks.dataset_helpers import VOC as ref
    out = {}
    for mode in ("none", "cutmix", "cutout"):
        img, lab, l1, l2 = inputs(rank)
        np.random.seed(50 + rank)
        out[mode] = [t.numpy() for t in ref.generate_cut_gather_2(img, lab.clone(), l1, l2, mode=mode)]
        if mode != "cutmix":
            continue                 # (generate_cut_gather_3 has no 'none' mode - VOC.py:468 raises - and its cutout branch never
                                     #  fills new_label2: torch.cat of an empty list, VOC.py:455-462,475)
        img, lab, l1, l2 = inputs(rank)
        np.random.seed(50 + rank)
        out[mode + "3"] = [t.numpy() for t in ref.generate_cut_gather_3(img, lab.clone(), lab + 1, l1, l2, mode=mode)]
    q.put((rank, out))
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29623, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join()
    arrs = {}
    for r in range(2):
        for k, t in enumerate(inputs(r)):
            arrs[f"in_r{r}_{k}"] = t.numpy()
        for mode, outs in res[r].items():
            for k, o in enumerate(outs):
                arrs[f"out_{mode}_r{r}_{k}"] = o
    np.savez_compressed(os.path.join(HERE, "cut_gather_w2.npz"), **arrs)
    print("wrote cut_gather_w2.npz:", len(arrs), "arrays")
