#!/usr/bin/env python3
"""Throughput of the CSS mix_label training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one iteration of mix_label.train (teacher fwd on labeled+unlabeled, student fwd+bwd on labeled+augmented
unlabeled, CE + confidence-weighted CE + prototype contrastive loss, fused SGD+EMA) over B labeled + B unlabeled
synthetic 513x513 crops per GPU (BASELINE.json configs[1]: VOC-shaped, torchvision-shaped ResNet-101 DeepLabv3+, B=16, bf16).
Weak scaling: every rank owns its own B+B crops; SyncBN statistics, prototype sums and the flat gradient are the only
exchanges (RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_FLOP_513_TV = 541.7e9          # forward FLOPs / image, tv-R101, 513^2, K=21
PEAK_BF16 = 2.5e15                 # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def synth_batch(B, S, K, seed, dev):
    """images N(0,1); labels piecewise-constant 32x32 blocks uniform over K classes, 5 % of blocks = -1."""
    g = torch.Generator().manual_seed(seed)
    l_img = torch.randn(B, 3, S, S, generator=g)
    u_img = torch.randn(B, 3, S, S, generator=g)
    nb = (S + 31) // 32
    blk = torch.randint(0, K, (B, nb, nb), generator=g)
    blk[torch.rand(B, nb, nb, generator=g) < 0.05] = -1
    l_lab = blk.repeat_interleave(32, 1).repeat_interleave(32, 2)[:, :S, :S].contiguous()
    return l_img.to(dev), l_lab.to(dev), u_img.to(dev)


def cpu_baseline(threads=None):
    """The oracle (CPU restatement of the reference path, kind 'port') on BASELINE config 1 (321x321, B=2+2, fp32,
    tv-R101, Q=256, N=512): one un-timed teacher-only warm-up, then ONE timed training step (bounded sample)."""
    import numpy as np
    from oracle import css_oracle as O
    if threads:
        torch.set_num_threads(threads)
    K, S, B = 21, 321, 2
    st = O.MixState("tv", K, 256, 3407)
    g = torch.Generator().manual_seed(3407)
    l_img, u_img = torch.randn(B, 3, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
    blk = torch.randint(0, K, (B, 11, 11), generator=g)
    l_lab = blk.repeat_interleave(32, 1).repeat_interleave(32, 2)[:, :S, :S].contiguous()
    torch.manual_seed(0)
    np.random.seed(0)
    with torch.no_grad():
        O.deeplab_forward(st.teacher, l_img[:1], "tv", False, K, 256)     # warm the allocator / oneDNN primitives
    t0 = time.time()
    O.train_step_mix(st, l_img, l_lab, u_img, lr=6.4e-3, temp_model=0.5, strong_threshold=0.8, weak_threshold=0.0, un_threshold=0.97,
                     num_queries=256, num_negatives=512)
    dt = time.time() - t0
    return {"value": round(2 * B / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 step of mix_label.train semantics at BASELINE configs[0] (321x321, B=2+2, fp32, tv-R101, Q=256, N=512), {dt:.1f} s"}


def pmc_traffic():
    """HBM bytes per conv_igemm_dma256_kernel launch from the committed rocprofv3 PMC passes (profiles/*pmc_hbm_traffic.csv: separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE doubled per the gfx950 correction of the guide)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.csv")))
    if not files:
        return None
    rd = wr = n = 0.0
    for r in csv.DictReader(open(files[-1])):
        if "conv_igemm_dma256" in r["kernel"]:
            k = float(r["launches"])
            rd += float(r["read_MB_per_launch_corrected_x2"]) * k
            wr += float(r["write_MB_per_launch"]) * k
            n += k
    return round((rd + wr) / n * 1e6) if n else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=513)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mix", default="cutmix")
    ap.add_argument("--aug", default="identity", choices=["identity", "pil"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2: VOC-shaped 513^2 tv-R101 B=16 (default, the headline metric); c4: Cityscapes-shaped 769^2 deep-stem R101 "
                         "K=19 OHEM B=8; c5: c4 with Q=1024, N=2048")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # CSS_FORCE_COLLECTIVES=1 on one GPU: a 1-rank RCCL group with every data-parallel exchange switched on (ops.collectives_on) -
    # measures what the ~450 collective calls of a step cost before any wire time
    forced = world == 1 and os.environ.get("CSS_FORCE_COLLECTIVES") == "1"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29581")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from css_amd import _lib
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from css_amd.train_step import MixTrainer

    K, S, B = 21, a.size, a.batch
    backbone, sup, Q, N = "tv", "ce", 256, 512
    if a.workload in ("c4", "c5"):
        K, S, B, backbone, sup = 19, 769, 8, "stem", "ohem"
        if a.workload == "c5":
            Q, N = 1024, 2048
    torch.manual_seed(3407)
    # --aug pil: the reference's in-step PIL pipeline on the device (random rescale 0.5-1.5, pad, crop, colour jitter, blur, flip,
    # 8-bit quantisation; css_amd/csrc/aug.hip) instead of the identity stand-in
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (0.5, 1.5) if a.aug == "pil" else (1.0, 1.0), "mix_mode": a.mix, "device_aug": a.aug}}
    bb = resnet.resnet101_tv(zero_init_residual=False) if backbone == "tv" else resnet.resnet101(zero_init_residual=False)
    with contextlib.redirect_stdout(sys.stderr):     # the constructor prints like the reference's; stdout carries the ONE JSON line only
        model = Model_mix(bb, num_classes=K, output_dim=256, config=cfg, temp=0.5)
    # seeded non-degenerate weights: Kaiming convs (constructor), BN gamma~U(.5,1.5), beta~N(0,.1)
    g = torch.Generator().manual_seed(3407)
    with torch.no_grad():
        for mod in model.model.modules():
            if mod.__class__.__name__ == "HipBatchNorm2d":
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.25)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
        model.ema_model.load_state_dict(model.model.state_dict())
    model = model.to(dev).train().set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    tr = MixTrainer(model, K, lr=6.4e-3, total_iter=80000, num_queries=Q, num_negatives=N, strong_threshold=0.8, weak_threshold=0.7,
                    un_threshold=0.97, sup=sup, ohem_min_kept=50000 * B)
    l_img, l_lab, u_img = synth_batch(B, S, K, 3407 + rank, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.step(l_img, l_lab, u_img)
    # Per-kernel HIP-event bracketing (roofline leg) costs ~2.8 us per event pair on the stream, 2.6 ms per step over the
    # ~920 conv kernel launches of a step: it is switched on for the LAST of the K timed steps only.
    _lib.lib().css_prof_reset()
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if i == a.steps - 1 and not os.environ.get("CSS_BENCH_NOPROF"):
            _lib.lib().css_prof_enable(1)
        out = tr.step(l_img, l_lab, u_img)
    sync()
    dt = time.perf_counter() - t0
    _lib.lib().css_prof_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    import ctypes
    prof = {}
    KINDS = ((0, "conv_fwd_other"), (1, "conv_dgrad_other"), (2, "conv_wgrad_other"), (3, "contrast_gather"), (4, "similarity"),
             (5, "igemm256_fwd"), (6, "igemm256_dgrad"), (7, "wgrad256"))
    for kind, name in KINDS:
        ms, n, w = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.lib().css_prof_read(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        prof[name] = (ms.value, n.value, w.value)
    if rank == 0:
        losses = {k: float(v) for k, v in out.items() if k != "pseudo"}

        def tot(*names):
            return tuple(sum(prof[n][i] for n in names) for i in range(3))
        # dominant kernel: conv_igemm_dma256_kernel (forward + dgrad launches; one event pair per kernel launch)
        ig_ms, ig_n, ig_fl = tot("igemm256_fwd", "igemm256_dgrad")
        ach = ig_fl / (ig_ms * 1e-3) / 1e12 if ig_ms > 0 else 0.0
        groups = {"conv_igemm_dma256_kernel": tot("igemm256_fwd", "igemm256_dgrad"), "conv_wgrad_dma256_kernel": prof["wgrad256"],
                  "conv_fwd_all_kernels": tot("conv_fwd_other", "igemm256_fwd"), "conv_dgrad_all_kernels": tot("conv_dgrad_other", "igemm256_dgrad"),
                  "conv_wgrad_all_kernels": tot("conv_wgrad_other", "wgrad256"), "contrast_gather": prof["contrast_gather"],
                  "similarity": prof["similarity"]}
        res = {
            "metric": "training images/sec at 513x513 R101-DeepLabv3+ (mix_label step, labeled+unlabeled crops consumed)",
            "value": round(2 * B * world * a.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: VOC-shaped mix_label step, tv-ResNet-101 DeepLabv3+, {S}x{S}, B={B}+{B} per GPU, "
                                    f"K={K}, Q={Q}, N={N}, mix_mode={a.mix}" + (", device_aug=pil" if a.aug == "pil" else "")) if a.workload == "c2" else
                                   (f"BASELINE configs[{3 if a.workload == 'c4' else 4}] shape on {world} GPU(s): Cityscapes-shaped mix_label step, deep-stem "
                                    f"ResNet-101 DeepLabv3+, {S}x{S}, B={B}+{B} per GPU, K={K}, OHEM, Q={Q}, N={N}, mix_mode={a.mix}"),
                       "global_batch": 2 * B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_dma256_kernel (forward + dgrad launches of the last timed step)",
                         "achieved": round(ach, 2), "peak": PEAK_BF16 / 1e12 if a.dtype == "bf16" else 157.3, "unit": "TFLOP/s",
                         "frac": round(ach * 1e12 / (PEAK_BF16 if a.dtype == "bf16" else 157.3e12), 4),
                         "traffic": pmc_traffic(), "launches_per_step": ig_n, "avg_launch_us": round(ig_ms * 1e3 / max(ig_n, 1), 2),
                         "alg_flops_per_launch": ig_fl / max(ig_n, 1)},
            "kernels": {k: {"ms_per_step": round(v[0], 3), "launches_per_step": v[1],
                            "alg_tflops_or_GBs": round(v[2] / max(v[0] * 1e-3, 1e-12) / (1e12 if k.startswith("conv") else 1e9), 2)}
                        for k, v in groups.items()},
            "step_alg_tflops": round(8 * B * (FWD_FLOP_513_TV * (S / 513.0) ** 2 if backbone == "tv" else 1239.1e9 * (S / 769.0) ** 2)
                                     / (dt / a.steps) / 1e12, 2),
            "losses": {k: round(v, 4) for k, v in losses.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
