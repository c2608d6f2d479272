"""The handful of torchvision-0.8.2 ``transforms.functional`` operations the reference's CPU data path calls on PIL images
(generalframeworks/dataset_helpers/VOC.py:64-124, Cityscapes.py:103-163), stated directly on PIL / numpy / torch so that the
loaders work without torchvision (not installed on the MI355X image).  Each function names the torchvision call it stands for.

This is host-side plumbing for DataLoader worker processes ("keep on CPU workers"): decode, geometry and colour ops
of single, differently-sized images.  The in-step, batched augmentation runs on the device (gpu_aug.py / csrc/aug.hip).
"""
from __future__ import annotations

import random as _random
from dataclasses import dataclass
from typing import Sequence, Tuple

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageFilter

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize(img: Image.Image, size_hw: Tuple[int, int], resample) -> Image.Image:
    """transforms_f.resize(img, (h, w), interpolation): PIL takes (w, h)."""
    return img.resize((int(size_hw[1]), int(size_hw[0])), resample)


def _rebuild(like: Image.Image, arr: np.ndarray) -> Image.Image:
    out = Image.fromarray(arr, like.mode)
    if like.mode == "P" and like.palette is not None:
        out.putpalette(like.getpalette())
    return out


def pad_right_bottom(img: Image.Image, right: int, bottom: int, mode: str, fill: int = 0) -> Image.Image:
    """transforms_f.pad(img, padding=(0, 0, right, bottom), fill=fill, padding_mode=mode) for mode 'reflect' | 'constant'.
    Palette ('P') label images keep their palette, as torchvision does."""
    a = np.asarray(img)
    widths = ((0, bottom), (0, right)) + (((0, 0),) if a.ndim == 3 else ())
    if mode == "constant":
        a = np.pad(a, widths, mode="constant", constant_values=fill)
    elif mode == "reflect":
        a = np.pad(a, widths, mode="reflect")
    else:
        raise ValueError(mode)
    return _rebuild(img, a)


def crop(img: Image.Image, top: int, left: int, h: int, w: int) -> Image.Image:
    """transforms_f.crop."""
    return img.crop((left, top, left + w, top + h))


def hflip(img: Image.Image) -> Image.Image:
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def adjust_hue(img: Image.Image, hue: float) -> Image.Image:
    """transforms_f.adjust_hue on an RGB PIL image: shift the 8-bit H plane by uint8(hue*255) with wrap-around."""
    if not -0.5 <= hue <= 0.5:
        raise ValueError("hue_factor is not in [-0.5, 0.5]")
    if img.mode in ("L", "1", "I", "F"):
        return img
    h, s, v = img.convert("HSV").split()
    nh = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        nh = nh + np.uint8(int(hue * 255) & 0xFF)
    return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert(img.mode)


def color_jitter(img: Image.Image, order: Sequence[int], brightness: float, contrast: float, saturation: float, hue: float):
    """transforms.ColorJitter.forward once the draws are made: the four adjustments in ``order``
    (0 brightness, 1 contrast, 2 saturation, 3 hue), each the PIL ImageEnhance blend torchvision delegates to."""
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(brightness)
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(contrast)
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(saturation)
        else:
            img = adjust_hue(img, hue)
    return img


def gaussian_blur(img: Image.Image, sigma: float) -> Image.Image:
    return img.filter(ImageFilter.GaussianBlur(radius=sigma))


def to_tensor(img: Image.Image) -> torch.Tensor:
    """transforms_f.to_tensor for 8-bit modes (RGB, L, P): bytes -> [C,H,W] float32 / 255."""
    a = np.asarray(img)
    if a.dtype != np.uint8:
        raise TypeError(f"8-bit image expected, got mode {img.mode}")
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255)


def normalize(t: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """transforms_f.normalize (out of place): (t - mean) / std in fp32, in that order."""
    m = torch.tensor(mean, dtype=t.dtype).view(-1, 1, 1)
    s = torch.tensor(std, dtype=t.dtype).view(-1, 1, 1)
    return (t - m) / s


def label_to_int(img: Image.Image) -> torch.Tensor:
    """(to_tensor(label) * 255).long() with 255 -> -1 (VOC.py:117-118): the float round trip x/255*255 is exact for all 256
    byte values after truncation EXCEPT that the reference's law is kept literally (long() truncates toward zero)."""
    lab = (to_tensor(img) * 255).long()
    lab[lab == 255] = -1
    return lab


# --------------------------------------------------------------------------
# the random draws of one labeled-image transform, with the reference's laws and draw ORDER
# --------------------------------------------------------------------------
@dataclass
class Draws:
    scale: float = 1.0
    crop_i: int = 0
    crop_j: int = 0
    jitter: bool = False
    order: Tuple[int, ...] = (0, 1, 2, 3)
    brightness: float = 1.0
    contrast: float = 1.0
    saturation: float = 1.0
    hue: float = 0.0
    blur: bool = False
    sigma: float = 0.0
    flip: bool = False


class DrawSource:
    """Random numbers in the order the reference consumes them (VOC.py:64-112): ``random.uniform`` for the scale and the blur
    sigma, ``torch.randint`` for the crop offsets (RandomCrop.get_params: no draw when the sizes are equal), ``torch.rand(1)``
    for the three gates, and ColorJitter-0.8.2's own stream (``torch.randperm(4)`` first, then one ``uniform_`` per adjustment
    AS THE PERMUTATION REACHES IT).  Both generators can be private (``seed``) so that a worker's draws are reproducible."""

    def __init__(self, seed=None):
        self.py = _random.Random(seed) if seed is not None else _random
        self.tg = torch.Generator().manual_seed(seed) if seed is not None else None

    def uniform(self, a, b):
        return self.py.uniform(a, b)

    def rand(self):
        return float(torch.rand(1, generator=self.tg))

    def randint(self, n):
        return int(torch.randint(0, n, (1,), generator=self.tg))

    def jitter(self, d: Draws):
        d.order = tuple(int(i) for i in torch.randperm(4, generator=self.tg))
        for op in d.order:
            lo, hi = ((0.75, 1.25), (0.75, 1.25), (0.75, 1.25), (-0.25, 0.25))[op]
            v = float(torch.empty(1).uniform_(lo, hi, generator=self.tg))
            if op == 0:
                d.brightness = v
            elif op == 1:
                d.contrast = v
            elif op == 2:
                d.saturation = v
            else:
                d.hue = v
