"""GPU-resident in-step augmentation of the reference
(``batch_transform_2/3`` + ``generate_cut_gather_2/3``, generalframeworks/dataset_helpers/VOC.py:325-352,393-477).

The reference round-trips every unlabeled image GPU -> CPU -> PIL -> GPU in the middle of ``Model_*.forward`` (SURVEY.md
section 8f-1).  Two modes, chosen by ``config['Dataset']['device_aug']`` (``aug_mode``); the step wrappers use ``"pil"`` when the key
is absent (the reference's YAML files do not have it, and the drop-in path must augment like the reference does):
  * ``"identity"`` (what the golden step traces were captured with; benchmarks and parity tests ask for it explicitly): geometry
    and colours untouched; only the label convention 255 ("disagree") -> -1, int64 (VOC.py:184-185);
  * ``"pil"`` (the default of ``Model_*``): the reference's whole PIL pipeline restated on 8-bit planes by HIP kernels (csrc/aug.hip), BIT-EXACT to PIL on
    identical random draws (tests/test_aug_gpu.py vs oracle/aug_oracle.py): tensor_to_pil_2's denormalise + 8-bit quantisation
    of the image and both confidence maps, random rescale (PIL BILINEAR two-pass fixed point / NEAREST), pad (reflect / 255 / 0),
    random crop, ColorJitter (PIL blends + 8-bit HSV hue shift, random order), GaussianBlur (PIL's three box passes per axis),
    horizontal flip, to_tensor, ImageNet normalisation.  The random draws are made on the host with the reference's laws
    (``draw_params``) - a few numbers per image.
Mixing (both modes): ``none`` | ``cutmix`` | ``cutout`` with the reference's box law (VOC.py:518-534), boxes drawn on the host
with numpy like the reference and applied by torch indexing on the device; classmix (VOC.py:505-510,430-437) with torch ops on the
device.  The partner of image i is ``gathered[(i+1) % B]`` (VOC.py:396-399,428): the reference all-gathers the four tensors and then
indexes the gathered batch with the LOCAL batch size, so on every rank the partner comes from RANK 0's batch, and every rank draws
one mask per gathered image and uses the block of its own rank.  ``_mix`` reproduces exactly that with one broadcast of rank 0's
tensors (and, for classmix only, an all-gather of the label maps the masks are drawn from) instead of four all-gathers;
``CSS_CUTMIX_LOCAL=1`` mixes inside the local batch instead (no collective, not the reference's law for world size > 1).
"""
from __future__ import annotations

import os

import numpy as np
import torch


def labels_to_int(labels: torch.Tensor) -> torch.Tensor:
    lab = labels.long()
    return torch.where(lab == 255, torch.full_like(lab, -1), lab)


_MODE = "identity"


class aug_mode:
    """``with aug_mode("pil")``: batch_transform* run the reference's PIL pipeline on the device (csrc/aug.hip, bit-exact to PIL:
    random rescale, pad, crop, 8-bit quantisation, colour jitter, blur, flip); ``"identity"``: geometry and colours
    untouched, label convention only (also the state outside any ``aug_mode`` context, for direct callers of batch_transform*).
    Model_* pick the mode from ``config['Dataset'].get('device_aug', 'pil')`` (ddp_model._StudentTeacher._device_aug)."""

    def __init__(self, mode):
        if mode not in ("identity", "pil"):
            raise ValueError("device_aug must be 'identity' or 'pil'")
        self.mode = mode

    def __enter__(self):
        global _MODE
        self.prev, _MODE = _MODE, self.mode

    def __exit__(self, *a):
        global _MODE
        _MODE = self.prev


def batch_transform_2(images, labels, logits_1=None, logits_2=None, crop_size=(512, 512), scale_size=(0.8, 1.0), augmentation=True):
    if _MODE == "pil":
        return device_batch_transform_2(images, labels, logits_1, logits_2, tuple(crop_size), scale_size, augmentation)
    return images, labels_to_int(labels), logits_1, logits_2


def batch_transform_3(images, labels1, labels2, logits_1=None, logits_2=None, crop_size=(512, 512), scale_size=(0.8, 1.0),
                      augmentation=True):
    if _MODE == "pil":
        b, _, h, w = images.shape
        cs = (h, w) if crop_size == -1 else tuple(crop_size)
        ps = [draw_params(h, w, cs, scale_size, augmentation) for _ in range(b)]
        img, l1, g1, g2 = device_batch_transform_2(images, labels1, logits_1, logits_2, cs, scale_size, augmentation, params=ps)
        _, l2, _, _ = device_batch_transform_2(images, labels2, logits_1, logits_2, cs, scale_size, augmentation, params=ps)
        return img, l1, l2, g1, g2
    return images, labels_to_int(labels1), labels_to_int(labels2), logits_1, logits_2


def batch_transform(images, labels, logits=None, crop_size=(512, 512), scale_size=(0.8, 1.0), augmentation=True):
    if _MODE == "pil":
        img, lab, g, _ = device_batch_transform_2(images, labels, logits, logits, tuple(crop_size) if crop_size != -1 else -1, scale_size,
                                                  augmentation)
        return img, lab, g
    return images, labels_to_int(labels), logits


def cutout_box(h, w, ratio=2, rng=np.random):
    """generate_cutout_mask (VOC.py:518-534): returns (y0, y1, x0, x1) of the zero region."""
    area = h * w / ratio
    bw = rng.randint(w / ratio + 1, w)
    bh = np.round(area / bw)
    x0 = rng.randint(0, w - bw + 1)
    y0 = rng.randint(0, h - bh + 1)
    return int(y0), int(y0 + bh), int(x0), int(x0 + bw)


def class_mask(pseudo_labels: torch.Tensor, generator=None) -> torch.Tensor:
    """generate_class_mask (VOC.py:505-510): 1 where the pixel's label is one of a random half of the image's labels."""
    labels = torch.unique(pseudo_labels)
    perm = torch.randperm(len(labels), generator=generator).to(labels.device)
    select = labels[perm][: len(labels) // 2]
    return (pseudo_labels.unsqueeze(-1) == select).any(dim=-1)


def _ranks():
    """(rank, world size) the mixing has to reproduce (VOC.py:396-402), (0, 1) without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and os.environ.get("CSS_CUTMIX_LOCAL") != "1":
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# ---- the partner batch under data parallelism -------------------------------------------------------------------------------------
# The reference mixes image i with gathered[(i + 1) % B] of the all-gathered batch (VOC.py:396-402): with the LOCAL batch size B that
# is always an image of RANK 0, so what every rank needs is rank 0's batch - a broadcast, not an all-gather.  Round 4:
#   * ONE packed buffer instead of one broadcast per tensor, class-id maps (int64, values -1 .. 254) travel as one byte per pixel:
#     117 MB -> 88 MB per step at c2 (image 50.5 + label 4.2 + two confidence maps 33.7), 4 collectives -> 1;
#   * on a process group of its own (its own RCCL communicator and stream): the transfer does not queue behind, or hold up, the
#     latency-critical SyncBN collectives of the default group;
#   * the image part does not depend on the teacher: when the in-step augmentation is the identity (benchmarks, parity traces) the model
#     starts its broadcast BEFORE the teacher pass (prefetch_partner_image) and only labels + confidence maps (38 MB) are sent behind it.
_mix_pg = None


def _mix_group():
    """Process group of the partner broadcasts (created on first use, by every rank at the same point of its first mixed step)."""
    global _mix_pg
    import torch.distributed as dist
    if _mix_pg is None:
        _mix_pg = dist.new_group()
    return _mix_pg


def _pack_layout(tensors):
    """[(offset, nbytes, as_u8)] of every tensor in the packed buffer (16-byte aligned parts) and the total size."""
    lay, off = [], 0
    for t in tensors:
        as_u8 = t.dtype == torch.int64
        nb = t.numel() * (1 if as_u8 else t.element_size())
        lay.append((off, nb, as_u8))
        off += (nb + 15) // 16 * 16
    return lay, off


def _broadcast_packed(tensors, group=None, async_op=False):
    """Rank 0's ``tensors`` on every rank, through one uint8 buffer.  -> (unpack, work): ``unpack()`` returns the tensors (contiguous,
    original dtypes and shapes); with ``async_op`` call ``work.wait()`` first."""
    import torch.distributed as dist
    lay, total = _pack_layout(tensors)
    buf = torch.empty(total, dtype=torch.uint8, device=tensors[0].device)
    if dist.get_rank() == 0:
        for t, (off, nb, as_u8) in zip(tensors, lay):
            src = t.contiguous(memory_format=torch.contiguous_format)
            src = src.to(torch.uint8) if as_u8 else src          # (-1 wraps to 255)
            buf[off:off + nb].copy_(src.reshape(-1).view(torch.uint8))
    work = dist.broadcast(buf, src=0, group=group, async_op=async_op)

    def unpack():
        outs = []
        for t, (off, nb, as_u8) in zip(tensors, lay):
            part = buf[off:off + nb]
            if as_u8:
                v = part.to(torch.int64)
                v = torch.where(v == 255, torch.full_like(v, -1), v)
            else:
                v = part.view(t.dtype)
            outs.append(v.view(t.shape))
        return outs

    return unpack, work


def prefetch_partner_image(image, mode):
    """Start the broadcast of rank 0's images now (they do not depend on the teacher pass); hand the result to generate_cut_gather_*
    as ``prefetched``.  None when nothing is exchanged (one rank, or a mode that mixes nothing across images)."""
    rank, world = _ranks()
    if world == 1 or mode not in ("cutmix", "classmix"):
        return None
    unpack, work = _broadcast_packed([image], group=_mix_group(), async_op=True)
    return unpack, work, image


def _mix(tensors, mode, rng, prefetched=None):
    image = tensors[0]
    b, _, h, w = image.shape
    if mode == "none":
        return tensors
    if mode not in ("cutmix", "cutout", "classmix"):
        raise ValueError("mode must be in none, cutout, cutmix, or classmix")
    rank, world = _ranks()
    partners = tensors                       # gathered[(i + 1) % B] = rank 0's batch
    labels_all = None
    if world > 1:
        import torch.distributed as dist
        if mode != "cutout":
            if prefetched is not None and prefetched[2] is image:
                rest, work = _broadcast_packed(tensors[1:], group=_mix_group())
                if prefetched[1] is not None:
                    prefetched[1].wait()     # (the compute stream waits; the host does not block)
                partners = prefetched[0]() + rest()
            else:
                unpack, work = _broadcast_packed(tensors, group=_mix_group())
                partners = unpack()
        if mode == "classmix":               # one mask per gathered image, drawn from that image's label map (VOC.py:412,424)
            labels_all = [torch.empty_like(tensors[1]) for _ in range(world)]
            dist.all_gather(labels_all, tensors[1].contiguous())
    if mode != "classmix" and all(t.is_cuda and t.is_contiguous() and t.element_size() in (4, 8) for t in list(tensors) + list(partners)):
        # device tensors: the boxes of the whole batch in ONE launch per tensor (css_mix_boxes) instead of a clone + one strided copy per
        # image and tensor; every rank still draws for ALL gathered images in order and keeps the block of its own rank (VOC.py:411-437)
        from .._lib import call, dev_stream
        boxes = torch.zeros((b, 4), dtype=torch.int32)
        for gi in range(world * b):
            y0, y1, x0, x1 = cutout_box(h, w, 2, rng)
            if rank * b <= gi < (rank + 1) * b:
                boxes[gi - rank * b] = torch.tensor([y0, y1, x0, x1], dtype=torch.int32)
        boxes = boxes.to(image.device)
        pj = ((torch.arange(b, dtype=torch.int32) + rank * b + 1) % b).to(image.device)      # partner of gathered image gi: (gi + 1) % B
        dev_i, st = dev_stream(image)
        outs = []
        for t, pt in zip(tensors, partners):
            o = torch.empty_like(t)
            planes = t.shape[1] if t.dim() == 4 else 1
            fill = -1 if t.dtype == torch.int64 else 0
            call("css_mix_boxes", t, pt, o, boxes, pj, b, planes, h, w, t.element_size(), 0 if mode == "cutmix" else 1, fill, dev_i, st)
            outs.append(o)
        return outs
    outs = [t.clone() for t in tensors]
    # every rank draws for ALL gathered images in order and keeps the block of its own rank (VOC.py:411-437)
    for gi in range(world * b):
        mine = rank * b <= gi < (rank + 1) * b
        i, j = gi - rank * b, (gi + 1) % b
        if mode == "classmix":
            # image i where the mask is 1, partner elsewhere (VOC.py:430-437); tensors[1] is the (first) label map
            keep = class_mask(tensors[1][i] if world == 1 else labels_all[gi // b][gi % b])
            if mine:
                for t, o, pt in zip(tensors, outs, partners):
                    o[i] = torch.where(keep if t.dim() == 3 else keep.unsqueeze(0), t[i], pt[j])
            continue
        y0, y1, x0, x1 = cutout_box(h, w, 2, rng)
        if not mine:
            continue
        for t, o, pt in zip(tensors, outs, partners):
            if mode == "cutmix":
                o[i, ..., y0:y1, x0:x1] = pt[j, ..., y0:y1, x0:x1]
            else:
                is_label = t.dtype == torch.int64
                o[i, ..., y0:y1, x0:x1] = -1 if is_label else 0
    return outs


def generate_cut_gather_2(image, label, logits1, logits2, mode="cutout", rng=np.random, prefetched=None):
    return tuple(_mix([image, label, logits1, logits2], mode, rng, prefetched))


def generate_cut_gather_3(image, label1, label2, logits1, logits2, mode="cutout", rng=np.random, prefetched=None):
    return tuple(_mix([image, label1, label2, logits1, logits2], mode, rng, prefetched))


def generate_cut_gather(image, label, logits, mode="cutout", rng=np.random, prefetched=None):
    return tuple(_mix([image, label, logits], mode, rng, prefetched))


# --------------------------------------------------------------------------------------------------------------------
# Faithful device path: the reference's PIL pipeline restated on 8-bit planes by HIP kernels (csrc/aug.hip)
# --------------------------------------------------------------------------------------------------------------------
import random as _random
from dataclasses import dataclass, field
from typing import Sequence


@dataclass
class AugParams:
    """The random draws of one transform_2 call (VOC.py:126-196); oracle/aug_oracle.py has the same record."""
    scale: float = 1.0
    crop_i: int = 0
    crop_j: int = 0
    jitter: bool = False
    order: Sequence[int] = (0, 1, 2, 3)
    brightness: float = 1.0
    contrast: float = 1.0
    saturation: float = 1.0
    hue: float = 0.0
    blur: bool = False
    sigma: float = 0.0
    flip: bool = False


def draw_params(h, w, crop_size, scale_size, augmentation, rng=None, trng=None):
    """One image's draws, with the reference's laws: scale ~ U(scale_size) (random.uniform, VOC.py:129), crop offsets uniform
    over the valid range (RandomCrop.get_params), jitter with p = 0.8 and ColorJitter((.75,1.25),(.75,1.25),(.75,1.25),(-.25,.25))
    in a random order, blur with p = 0.5 and sigma ~ U(0.15, 1.15), flip with p = 0.5 (VOC.py:162-181)."""
    rng = rng or _random
    u = (lambda: float(torch.rand(1, generator=trng))) if trng is not None else (lambda: float(torch.rand(1)))
    p = AugParams()
    p.scale = rng.uniform(scale_size[0], scale_size[1])
    rh, rw = int(h * p.scale), int(w * p.scale)
    ph, pw = max(rh, crop_size[0]), max(rw, crop_size[1])
    p.crop_i = int(torch.randint(0, ph - crop_size[0] + 1, (1,), generator=trng)) if ph > crop_size[0] else 0
    p.crop_j = int(torch.randint(0, pw - crop_size[1] + 1, (1,), generator=trng)) if pw > crop_size[1] else 0
    if augmentation:
        if u() > 0.2:
            p.jitter = True
            p.order = tuple(int(i) for i in torch.randperm(4, generator=trng))
            p.brightness = float(torch.empty(1).uniform_(0.75, 1.25, generator=trng))
            p.contrast = float(torch.empty(1).uniform_(0.75, 1.25, generator=trng))
            p.saturation = float(torch.empty(1).uniform_(0.75, 1.25, generator=trng))
            p.hue = float(torch.empty(1).uniform_(-0.25, 0.25, generator=trng))
        if u() > 0.5:
            p.blur = True
            p.sigma = rng.uniform(0.15, 1.15)
        if u() > 0.5:
            p.flip = True
    return p


def device_batch_transform_2(images, labels, logits_1, logits_2, crop_size=(512, 512), scale_size=(0.8, 1.0), augmentation=True,
                             params=None):
    """batch_transform_2 (VOC.py:339-352) without leaving the device.  images [B,3,H,W] fp32 (normalised), labels [B,H,W]
    (class ids, 255 or -1; any real dtype), logits [B,H,W] fp32 in [0,1] -> (image fp32 [B,3,Hc,Wc], label int64 with -1,
    two fp32 maps), all through the reference's 8-bit quantisation.  ``params``: one AugParams per image (drawn here if None)."""
    from .._lib import call, dev_stream
    b, _, h, w = images.shape
    if crop_size == -1:
        crop_size = (h, w)
    hc, wc = int(crop_size[0]), int(crop_size[1])
    if params is None:
        params = [draw_params(h, w, (hc, wc), scale_size, augmentation) for _ in range(b)]
    if any(p.scale < 0.5 for p in params):
        raise ValueError("the device resize restates PIL's filter for scales >= 0.5 (the reference's configs use 0.5 .. 2.0)")
    dev, st = dev_stream(images)
    geo = torch.tensor([[int(h * p.scale), int(w * p.scale), p.crop_i, p.crop_j] for p in params], dtype=torch.int32).to(images.device)
    maxlen = max(max(int(h * p.scale), int(w * p.scale)) for p in params)
    maxlen = max(maxlen, h, w)
    table = torch.empty((2 * b, maxlen), dtype=torch.int32, device=images.device)
    u8 = dict(dtype=torch.uint8, device=images.device)
    img_q, lab_q = torch.empty((b, 3, hc, wc), **u8), torch.empty((b, hc, wc), **u8)
    l1_q, l2_q = torch.empty((b, hc, wc), **u8), torch.empty((b, hc, wc), **u8)
    call("css_aug_geom", images.float().contiguous(), labels.float().contiguous(), logits_1.float().contiguous(),
         logits_2.float().contiguous(), geo, table, maxlen, b, h, w, hc, wc, img_q, lab_q, l1_q, l2_q, dev, st)
    if augmentation:
        img_q = _device_color_ops(img_q, params)
    flags = torch.tensor([1 if (augmentation and p.flip) else 0 for p in params], dtype=torch.int32).to(images.device)
    out_img = torch.empty((b, 3, hc, wc), dtype=torch.float32, device=images.device)
    out_lab = torch.empty((b, hc, wc), dtype=torch.int64, device=images.device)
    out_l1 = torch.empty((b, hc, wc), dtype=torch.float32, device=images.device)
    out_l2 = torch.empty((b, hc, wc), dtype=torch.float32, device=images.device)
    call("css_aug_finish", img_q, lab_q, l1_q, l2_q, flags, b, hc, wc, out_img, out_lab, out_l1, out_l2, dev, st)
    return out_img, out_lab, out_l1, out_l2


def _box_blur_weights(sigma):
    """ImageFilter.GaussianBlur(radius=sigma) -> BoxBlur.c: the fractional box radius of three passes (_gaussian_blur_radius,
    float arithmetic as in C) and the 24-bit fixed-point weights of ImagingHorizontalBoxBlur.  Returns (ww, fw) for an integer
    radius of 0, which is what sigma <= 1.15 gives (the reference draws sigma from [0.15, 1.15], VOC.py:170)."""
    f32 = np.float32
    radius = f32(sigma)
    sigma2 = f32(f32(radius * radius) / f32(3))
    big_l = f32(np.sqrt(12.0 * float(sigma2) + 1.0))
    small_l = f32(np.floor((float(big_l) - 1.0) / 2.0))
    a = f32(f32(f32(2) * small_l + f32(1)) * f32(f32(small_l * f32(small_l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(small_l + f32(1)) * f32(small_l + f32(1))))))
    fr = f32(small_l + a)
    if int(fr) != 0:
        raise ValueError("GaussianBlur sigma > 1.15 needs an integer box radius >= 1: outside the reference's range")
    ww = int(f32(16777216.0) / f32(f32(fr * f32(2)) + f32(1)))
    fw = ((1 << 24) - ww) // 2
    return ww, fw


def _pack_color_params(params):
    rows = []
    for p in params:
        row = [0] * 16
        if p.jitter:
            row[0] = 1
            row[1:5] = [int(o) for o in p.order]
            row[5:8] = [int(np.float32(v).view(np.int32)) for v in (p.brightness, p.contrast, p.saturation)]
            row[8] = int(p.hue * 255) & 0xFF                  # np.uint8(hue_factor * 255): truncation, wrap-around
        if p.blur:
            row[9] = 1
            row[10], row[11] = _box_blur_weights(p.sigma)
        rows.append(row)
    return torch.tensor(rows, dtype=torch.int32)


def _device_color_ops(img_q, params):
    """Colour jitter and Gaussian blur on the uint8 image planes (csrc/aug.hip), in the reference's order: jitter, then blur."""
    from .._lib import call, dev_stream
    any_j, any_b = any(p.jitter for p in params), any(p.blur for p in params)
    if not (any_j or any_b):
        return img_q
    b, _, h, w = img_q.shape
    jp = _pack_color_params(params).to(img_q.device)
    tmp = torch.empty_like(img_q)
    sums = torch.empty(b, dtype=torch.int64, device=img_q.device)
    dev, st = dev_stream(img_q)
    call("css_aug_color", img_q, tmp, jp, sums, b, h, w, int(any_j), int(any_b), dev, st)
    return img_q
