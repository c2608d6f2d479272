"""Pascal-VOC split reader, dataset and the labeled-image CPU transform, plus the names the reference's step
wrappers import from this module (the in-step augmentation, which here runs on the device: gpu_aug.py).

Mirrors generalframeworks/dataset_helpers/VOC.py: ``get_pascal_idx_via_txt`` :48-62, ``VOC_BuildData`` :29-46,
``Pascal_VOC_Dataset`` :11-27, ``transform`` :64-124 - same constructor arguments, return values, directory layout
(``JPEGImages/<id>.jpg``, ``SegmentationClassAug/<id>.png``, ``<txt>/<label_num>/<seed>/{labeled,unlabeled,valid}_filename.txt``),
error behaviour (missing files raise from ``open`` / ``Image.open``) and random-draw laws, so ``mix_label.py:36-60`` builds its
three DataLoaders unchanged.  torchvision is not needed (pil_ops.py).  This is CPU-worker code by design: single images of
different sizes are decoded and cropped to the fixed crop here; everything batched happens on the MI355X.
"""
from __future__ import annotations

import os

import torch
import torch.utils.data as data
from PIL import Image

from . import pil_ops as P
from .gpu_aug import (batch_transform, batch_transform_2, batch_transform_3, generate_cut_gather,  # noqa: F401  (re-exports)
                      generate_cut_gather_2, generate_cut_gather_3)
from .pil_ops import Draws, DrawSource


def read_split(txt_root, label_num, seed):
    """The three id lists of one labeled-fraction split, in file order (blank-line handling = str.splitlines, like the reference)."""
    d = f"{txt_root}/{label_num}/{seed}"
    out = []
    for name in ("labeled_filename.txt", "unlabeled_filename.txt", "valid_filename.txt"):
        with open(f"{d}/{name}") as f:
            out.append(f.read().splitlines())
    return tuple(out)


def get_pascal_idx_via_txt(root, label_num, seed):
    """VOC.py:48-62."""
    return read_split(root, label_num, seed)


def draw(raw_hw, crop_size, scale_size, augmentation, src: DrawSource) -> Draws:
    """All random numbers of one ``transform`` call, consumed in the reference's order (see pil_ops.DrawSource)."""
    d = Draws()
    d.scale = src.uniform(scale_size[0], scale_size[1])
    rh, rw = int(raw_hw[0] * d.scale), int(raw_hw[1] * d.scale)
    ph, pw = max(rh, crop_size[0]), max(rw, crop_size[1])
    if not (ph == crop_size[0] and pw == crop_size[1]):        # RandomCrop.get_params draws nothing when nothing can move
        d.crop_i = src.randint(ph - crop_size[0] + 1)
        d.crop_j = src.randint(pw - crop_size[1] + 1)
    if augmentation:
        if src.rand() > 0.2:
            d.jitter = True
            src.jitter(d)
        if src.rand() > 0.5:
            d.blur = True
            d.sigma = src.uniform(0.15, 1.15)
        if src.rand() > 0.5:
            d.flip = True
    return d


def apply(image, label, logits, d: Draws, crop_size, augmentation):
    """The deterministic part of ``transform`` for fixed draws: rescale (BILINEAR image / NEAREST label, logits), pad to the
    crop size at the right / bottom (image reflect, label 255, logits 0), crop, colour jitter, blur, flip, to_tensor,
    label 255 -> -1, ImageNet normalisation."""
    raw_w, raw_h = image.size
    rh, rw = int(raw_h * d.scale), int(raw_w * d.scale)
    image = P.resize(image, (rh, rw), Image.BILINEAR)
    label = P.resize(label, (rh, rw), Image.NEAREST)
    if logits is not None:
        logits = P.resize(logits, (rh, rw), Image.NEAREST)
    ch, cw = crop_size
    if ch > rh or cw > rw:
        right, bottom = max(cw - rw, 0), max(ch - rh, 0)
        image = P.pad_right_bottom(image, right, bottom, "reflect")
        label = P.pad_right_bottom(label, right, bottom, "constant", 255)
        if logits is not None:
            logits = P.pad_right_bottom(logits, right, bottom, "constant", 0)
    image = P.crop(image, d.crop_i, d.crop_j, ch, cw)
    label = P.crop(label, d.crop_i, d.crop_j, ch, cw)
    if logits is not None:
        logits = P.crop(logits, d.crop_i, d.crop_j, ch, cw)
    if augmentation:
        if d.jitter:
            image = P.color_jitter(image, d.order, d.brightness, d.contrast, d.saturation, d.hue)
        if d.blur:
            image = P.gaussian_blur(image, d.sigma)
        if d.flip:
            image, label = P.hflip(image), P.hflip(label)
            if logits is not None:
                logits = P.hflip(logits)
    out_image = P.normalize(P.to_tensor(image))
    out_label = P.label_to_int(label)
    if logits is not None:
        return out_image, out_label, P.to_tensor(logits)
    return out_image, out_label


def transform(image, label, logits=None, crop_size=(512, 512), scale_size=(0.8, 1.0), augmentation=True, draws=None, source=None):
    """VOC.py:64-124.  image RGB PIL, label 8-bit PIL (class ids, 255 = ignore), optional logits 8-bit PIL ->
    (image fp32 [3,h,w] normalised, label int64 [1,h,w] with -1 = ignore[, logits fp32 [1,h,w]]).
    ``crop_size == -1`` keeps the reference's literal behaviour: the crop becomes (raw_w, raw_h) (VOC.py:138-139).
    ``draws`` injects the random numbers (tests); ``source`` a private generator pair (reproducible workers)."""
    raw_w, raw_h = image.size
    if crop_size == -1:
        crop_size = (raw_w, raw_h)
    crop_size = (int(crop_size[0]), int(crop_size[1]))
    if draws is None:
        draws = draw((raw_h, raw_w), crop_size, scale_size, augmentation, source or DrawSource())
    return apply(image, label, logits, draws, crop_size, augmentation)


class Pascal_VOC_Dataset(data.Dataset):
    """VOC.py:11-27: item = (image fp32 [3,h,w], label int64 [h,w])."""

    def __init__(self, root, idx_list, crop_size=(512, 512), scale_size=(0.5, 2.0), augmentation=True, train=True):
        self.root = os.path.expanduser(root)
        self.train = train
        self.crop_size = crop_size
        self.augmentation = augmentation
        self.scale_size = scale_size
        self.idx_list = idx_list

    def paths(self, index):
        name = self.idx_list[index]
        return f"{self.root}/JPEGImages/{name}.jpg", f"{self.root}/SegmentationClassAug/{name}.png"

    def __getitem__(self, index):
        ip, lp = self.paths(index)
        image, label = transform(Image.open(ip), Image.open(lp), None, crop_size=self.crop_size, scale_size=self.scale_size,
                                 augmentation=self.augmentation)
        return image, label.squeeze(0)

    def __len__(self):
        return len(self.idx_list)


class VOC_BuildData:
    """VOC.py:29-46: labeled set with scale 0.5-1.5 + augmentation, unlabeled and validation sets at scale 1 without."""

    def __init__(self, data_path, txt_path, label_num, seed, crop_size=[512, 512]):
        self.data_path = data_path
        self.txt_path = txt_path
        self.image_size = [513, 513]
        self.crop_size = crop_size
        self.num_segments = 21
        self.scale_size = (0.5, 1.5)
        self.train_l_idx, self.train_u_idx, self.test_idx = get_pascal_idx_via_txt(self.txt_path, label_num=label_num, seed=seed)

    def build(self):
        mk = Pascal_VOC_Dataset
        return (mk(self.data_path, self.train_l_idx, self.crop_size, self.scale_size, augmentation=True, train=True),
                mk(self.data_path, self.train_u_idx, self.crop_size, scale_size=(1.0, 1.0), augmentation=False, train=True),
                mk(self.data_path, self.test_idx, self.crop_size, scale_size=(1.0, 1.0), augmentation=False, train=False))
