"""GPU-resident stand-in for the in-step augmentation of the reference
(``batch_transform_2/3`` + ``generate_cut_gather_2/3``, generalframeworks/dataset_helpers/VOC.py:325-352,393-477).

The reference round-trips every unlabeled image GPU -> CPU -> PIL -> GPU in the middle of ``Model_*.forward``; that path
is data augmentation ("next" row) and is out of scope for bit parity.  What the hot path needs from
it is kept, on the device:
  * geometry: identity (scale 1.0, crop = input size), i.e. what ``scale_size=(1.0,1.0)`` and ``crop_size == image size``
    give in the reference;
  * label convention: 255 ("disagree") -> -1, int64  (VOC.py:184-185);
  * mixing: ``none`` | ``cutmix`` | ``cutout`` with the reference's box law (VOC.py:518-534) and partner ``(i+1) % B``
    (VOC.py:428), boxes drawn on the host with numpy like the reference and applied by torch indexing on the device.
Colour jitter / blur / flip / random rescale are NOT applied (statistical, not part of the parity contract).
"""
from __future__ import annotations

import numpy as np
import torch


def labels_to_int(labels: torch.Tensor) -> torch.Tensor:
    lab = labels.long()
    return torch.where(lab == 255, torch.full_like(lab, -1), lab)


def batch_transform_2(images, labels, logits_1=None, logits_2=None, crop_size=(512, 512), scale_size=(0.8, 1.0), augmentation=True):
    return images, labels_to_int(labels), logits_1, logits_2


def batch_transform_3(images, labels1, labels2, logits_1=None, logits_2=None, crop_size=(512, 512), scale_size=(0.8, 1.0),
                      augmentation=True):
    return images, labels_to_int(labels1), labels_to_int(labels2), logits_1, logits_2


def batch_transform(images, labels, logits=None, crop_size=(512, 512), scale_size=(0.8, 1.0), augmentation=True):
    return images, labels_to_int(labels), logits


def cutout_box(h, w, ratio=2, rng=np.random):
    """generate_cutout_mask (VOC.py:518-534): returns (y0, y1, x0, x1) of the zero region."""
    area = h * w / ratio
    bw = rng.randint(w / ratio + 1, w)
    bh = np.round(area / bw)
    x0 = rng.randint(0, w - bw + 1)
    y0 = rng.randint(0, h - bh + 1)
    return int(y0), int(y0 + bh), int(x0), int(x0 + bw)


def _mix(tensors, mode, rng):
    image = tensors[0]
    b, _, h, w = image.shape
    if mode == "none":
        return tensors
    outs = [t.clone() for t in tensors]
    for i in range(b):
        y0, y1, x0, x1 = cutout_box(h, w, 2, rng)
        j = (i + 1) % b
        for k, (t, o) in enumerate(zip(tensors, outs)):
            if mode == "cutmix":
                o[i, ..., y0:y1, x0:x1] = t[j, ..., y0:y1, x0:x1]
            elif mode == "cutout":
                is_label = t.dtype == torch.int64
                o[i, ..., y0:y1, x0:x1] = -1 if is_label else 0
            else:
                raise ValueError("mode must be none, cutout or cutmix (classmix is not implemented on the device)")
    return outs


def generate_cut_gather_2(image, label, logits1, logits2, mode="cutout", rng=np.random):
    return tuple(_mix([image, label, logits1, logits2], mode, rng))


def generate_cut_gather_3(image, label1, label2, logits1, logits2, mode="cutout", rng=np.random):
    return tuple(_mix([image, label1, label2, logits1, logits2], mode, rng))


def generate_cut_gather(image, label, logits, mode="cutout", rng=np.random):
    return tuple(_mix([image, label, logits], mode, rng))


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


def _device_color_ops(img_q, params):
    """Colour jitter and Gaussian blur on the uint8 image planes (filled in by the colour stage of csrc/aug.hip)."""
    if any(p.jitter or p.blur for p in params):
        raise NotImplementedError("colour jitter / blur on the device: next stage")
    return img_q
