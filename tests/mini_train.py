"""A miniature but COMPLETE training run of the reference's mix_label.py main loop on the HIP path (helper of
tests/test_mini_training_gpu.py and scripts/mini_train_curves.py; VERDICT r05 item 2 / the closable proxy of north_star's mIoU clause):

    VOC-shaped scratch tree -> VOC_BuildData(...).build() -> three DataLoaders            mix_label.py:36-60
    per epoch: train() over the labeled loader, the unlabeled iterator beside it          mix_label.py:149-197
               (MixTrainer.step, in-step augmentation device_aug='pil', cutmix)
               test() on the EMA model -> mIoU                                            mix_label.py:199-225, util/miou.py:3-9
               save best_model.pth when the mIoU is the best so far                       mix_label.py:131-147
    resume: load_checkpoint -> start_epoch -> the same loop                               mix_label.py:104-112

The task is synthetic but LEARNABLE: the label of a pixel is a function of the image around it (class-dependent brightness / texture /
saturation statistics under noise), there is a held-out validation split, a small labeled split and a larger unlabeled one - so the mIoU of
the EMA model rises from chance, which random-label synthetic data (tests/test_bf16_trajectory_gpu.py) cannot show.

Every random stream of an epoch (loader shuffles, worker seeds, in-step augmentation draws, sampler seed) is re-seeded from (seed, epoch)
at the epoch's start, as a DistributedSampler.set_epoch does for the reference's shuffles (mix_label.py:152-154): the state of a run at an
epoch boundary is then the checkpoint alone, which is what makes "resume == never interrupted" a testable, bit-for-bit statement.
"""
import os
import random
import time

import numpy as np
import torch

K = 6          # classes of the synthetic task (0 = background)


def _render(lab, rng):
    """uint8 RGB image whose statistics around a pixel determine its class (255 = boundary pixels take their neighbours' look)."""
    h, w = lab.shape
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3), np.float32)
    look = np.where(lab == 255, 0, lab)
    stripes_h = np.where((yy // 4) % 2 == 0, 60.0, 200.0)
    stripes_v = np.where((xx // 4) % 2 == 0, 60.0, 200.0)
    checker = np.where(((yy // 4) + (xx // 4)) % 2 == 0, 50.0, 210.0)
    for c, base in ((0, 90.0), (1, 215.0), (2, stripes_h), (3, stripes_v), (4, checker)):
        m = look == c
        img[m] = base[m][:, None] if isinstance(base, np.ndarray) else base
    m = look == 5
    img[m] = np.array([220.0, 45.0, 45.0], np.float32)           # saturated: survives the hue jitter as "not gray"
    img += rng.randn(h, w, 3).astype(np.float32) * 10.0
    return np.clip(img, 0, 255).astype(np.uint8)


def _label_map(h, w, rng):
    lab = np.zeros((h, w), np.uint8)
    for _ in range(rng.randint(3, 7)):
        c = rng.randint(1, K)
        bh, bw = rng.randint(28, 72), rng.randint(28, 72)
        y0, x0 = rng.randint(-8, h - 20), rng.randint(-8, w - 20)
        ys, xs = slice(max(y0, 0), min(y0 + bh, h)), slice(max(x0, 0), min(x0 + bw, w))
        lab[max(y0 - 1, 0):min(y0 + bh + 1, h), max(x0 - 1, 0):min(x0 + bw + 1, w)] = 255      # VOC-style ignore rim
        lab[ys, xs] = c
    return lab


def make_dataset(root, txt, S=129, n_l=16, n_u=32, n_val=16, repeat_l=8, repeat_u=4, seed=0):
    """JPEGImages/<id>.jpg + SegmentationClassAug/<id>.png + <txt>/<label_num>/<seed>/{labeled,unlabeled,valid}_filename.txt, the layout
    VOC_BuildData reads (VOC.py:29-62).  Training images are larger than the crop (the CPU transform rescales 0.5-1.5 and crops);
    validation images are exactly S x S (scale 1, nothing to crop: the validation set is the same pixels in every run).  The id lists
    repeat every id (an "epoch" of the loop = repeat x the distinct images, each repeat a fresh random crop)."""
    from PIL import Image
    os.makedirs(f"{root}/JPEGImages", exist_ok=True)
    os.makedirs(f"{root}/SegmentationClassAug", exist_ok=True)
    rng = np.random.RandomState(seed)
    ids = {"l": [], "u": [], "v": []}
    for part, n in (("l", n_l), ("u", n_u), ("v", n_val)):
        for i in range(n):
            name = f"2007_{part}{i:05d}"
            h, w = (S, S) if part == "v" else (S + rng.randint(8, 48), S + rng.randint(8, 48))
            lab = _label_map(h, w, rng)
            Image.fromarray(_render(lab, rng)).save(f"{root}/JPEGImages/{name}.jpg", quality=92)
            Image.fromarray(lab).save(f"{root}/SegmentationClassAug/{name}.png")
            ids[part].append(name)
    d = f"{txt}/{n_l}/3407"
    os.makedirs(d, exist_ok=True)
    for fname, part, rep in (("labeled_filename.txt", "l", repeat_l), ("unlabeled_filename.txt", "u", repeat_u), ("valid_filename.txt", "v", 1)):
        with open(f"{d}/{fname}", "w") as f:
            f.write("\n".join(ids[part] * rep))
    return dict(data_path=root, txt_path=txt, label_num=n_l, seed=3407, crop_size=[S, S])


def seed_all(s):
    random.seed(s)
    np.random.seed(s % (2 ** 32))
    torch.manual_seed(s)


def fingerprint(t):
    """Two order-sensitive checksums of a tensor's bits (equal fingerprints <=> bit-identical, for all practical purposes)."""
    v = t.detach().contiguous().flatten().view(torch.int32).to(torch.int64)
    idx = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521 + 1
    return int(v.sum()), int((v * idx).sum())


def run(ds, dtype, epochs, seed=1, B=4, lr=0.02, ema_alpha=0.95, perturb=None, ckpt_dir=None, snapshot_epoch=None, resume=None,
        workers=0, device=None, log=None, mp_context=None):
    """The main loop of mix_label.py:86-147 for ``epochs`` epochs.  ``perturb``: seed of a 1-ulp relative perturbation of every input batch
    (the "another correct fp32 run" of the mIoU noise floor).  ``snapshot_epoch``: also save <ckpt_dir>/snap.pth after that epoch (exact-resume
    format).  ``resume``: path of a checkpoint to continue from.  Returns the mIoU curve, the best mIoU, fingerprints of the final state."""
    from css_amd import checkpoint as ck
    from css_amd import evaluate
    from css_amd.dataset_helpers import VOC
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from css_amd.train_step import MixTrainer
    dev = device or torch.device("cuda:0")
    S = ds["crop_size"][0]
    cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [0.5, 1.5], "mix_mode": "cutmix", "device_aug": "pil"}, "Network": {"num_class": K}}
    train_l, train_u, test_set = VOC.VOC_BuildData(**ds).build()
    steps_per_epoch = len(train_l) // B
    seed_all(seed)                                                             # same initial weights in every run of a seed
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, ema_alpha=ema_alpha, config=cfg, temp=0.25).to(dev)
    m.model.train()
    m.ema_model.train()
    m.set_compute_dtype(dtype)
    tr = MixTrainer(m, num_classes=K, lr=lr, total_iter=epochs * steps_per_epoch, num_queries=256, num_negatives=512)
    start_epoch = 0
    if resume:
        start_epoch = ck.load_checkpoint(resume, tr, exact_resume=True)
    # workers = 0 by default: with FORKED loader workers alive beside a live HIP context every step of the parent slows down 4-6x on this stack
    # (170 ms instead of 31-50 per step, 5 s per 16-image evaluation, once a 20-minute stall; pinned or not: profiles/r06_mini_training_timing.txt) -
    # a property of fork() next to the GPU runtime, not of this library; INTEGRATION.md 5 says what a training script should pass instead.
    mk = lambda dset, e, tag, shuffle=True: torch.utils.data.DataLoader(
        dset, batch_size=B, drop_last=True, num_workers=workers, shuffle=shuffle, pin_memory=workers > 0,
        multiprocessing_context=mp_context if workers > 0 else None,
        generator=torch.Generator().manual_seed(seed * 7919 + 31 * e + tag) if shuffle else None)
    gp = torch.Generator(device=dev).manual_seed(perturb) if perturb is not None else None
    curve, best, losses = [], 0.0, []
    timing = dict(setup=0.0, data=0.0, h2d=0.0, steps=0.0, eval=0.0, save=0.0)      # seconds: where the wall time of a run goes
    t_mark = time.time()

    def lap(key, sync=False):
        nonlocal t_mark
        if sync:
            torch.cuda.synchronize()
        now = time.time()
        timing[key] += now - t_mark
        t_mark = now
    for epoch in range(start_epoch, epochs):
        seed_all(seed * 100003 + epoch)                                        # every stream of the epoch from (seed, epoch)
        m.model.train()
        m.ema_model.train()
        u_iter = iter(mk(train_u, epoch, 1))
        acc = None
        lap("setup")
        for l_img, l_lab in mk(train_l, epoch, 0):                            # mix_label.py:156-165
            u_img, _ = next(u_iter)
            lap("data")
            l_img, l_lab, u_img = l_img.to(dev), l_lab.to(dev), u_img.to(dev)
            lap("h2d")
            if gp is not None:
                l_img = l_img * (1 + 1e-7 * torch.randn(l_img.shape, generator=gp, device=dev))
                u_img = u_img * (1 + 1e-7 * torch.randn(u_img.shape, generator=gp, device=dev))
            r = tr.step(l_img, l_lab, u_img)
            vec = torch.stack([r[k].float().reshape(()) for k in ("sup", "unsup", "contrast")])
            acc = vec if acc is None else acc + torch.nan_to_num(vec)
            lap("steps")
        losses.append((acc / steps_per_epoch).tolist())
        lap("steps", sync=True)
        miou = float(evaluate.test(mk(test_set, epoch, 2, shuffle=False), m.ema_model, cfg))    # mix_label.py:131 (every epoch here)
        curve.append(miou)
        best = max(best, miou)
        lap("eval", sync=True)
        if log:
            log(f"epoch {epoch} it {tr.it} lr {tr.lr:.5f} losses {[round(x, 4) for x in losses[-1]]} mIoU {miou:.4f} best {best:.4f}")
        if ckpt_dir and miou == best:                                          # mix_label.py:137-147
            ck.save_checkpoint(os.path.join(ckpt_dir, "best_model.pth"), tr, epoch)
        if ckpt_dir and snapshot_epoch == epoch:
            ck.save_checkpoint(os.path.join(ckpt_dir, "snap.pth"), tr, epoch, exact_resume=True)
        lap("save", sync=True)
    tr.finish()
    torch.cuda.synchronize()
    out = dict(curve=curve, best=best, losses=losses, it=tr.it, fp_student=fingerprint(tr.flat_p), fp_teacher=fingerprint(tr.flat_ema),
               fp_momentum=fingerprint(tr.flat_m), fp_proto=fingerprint(tr.prototypes),
               fp_bn=fingerprint(torch.cat([b.float().flatten() for b in m.ema_model.buffers()])))
    out["timing"] = {k: round(v, 2) for k, v in timing.items()}
    out["trainer"] = tr
    return out
