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
