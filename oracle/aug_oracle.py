"""CPU restatement of the reference's IN-STEP augmentation, test infrastructure only.

Follows generalframeworks/dataset_helpers/VOC.py: tensor_to_pil_2 :284-291 (denormalise :309-316, 8-bit quantisation of the
image AND of the two confidence maps), transform_2 :126-196 (rescale -> pad -> crop -> colour jitter -> blur -> flip ->
to_tensor -> normalise), batch_transform_2 :339-352.  torchvision 0.8.2 is not installed here; its functional ops used by that
code are restated from their published definitions directly on PIL / numpy / torch:
  to_pil_image(float CHW) = pic.mul(255).byte() -> PIL;  to_tensor(PIL) = uint8 -> float32 .div(255);
  normalize = sub_(mean).div_(std);  resize = Image.resize((w, h), resample);  pad(reflect) = np.pad(mode='reflect'),
  pad(constant, fill) = constant border;  crop = Image.crop;  hflip = FLIP_LEFT_RIGHT;
  RandomCrop.get_params -> (i, j) uniform over the valid offsets.
The random draws (scale, crop offsets, jitter factors / order, blur sigma, the three Bernoulli gates) are INPUTS here
(``AugParams``) so that the HIP path can be compared on identical draws; their distributions are checked separately.
parity pinned by: PIL itself (the same library the reference calls) - no reference-side golden exists for this path."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch
from PIL import Image

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


@dataclass
class AugParams:
    scale: float = 1.0                 # random.uniform(scale_size)           VOC.py:129
    crop_i: int = 0                    # RandomCrop.get_params                VOC.py:154
    crop_j: int = 0
    jitter: bool = False               # torch.rand(1) > 0.2                  VOC.py:164
    order: Sequence[int] = (0, 1, 2, 3)   # ColorJitter's random permutation: 0 brightness 1 contrast 2 saturation 3 hue
    brightness: float = 1.0
    contrast: float = 1.0
    saturation: float = 1.0
    hue: float = 0.0
    blur: bool = False                 # torch.rand(1) > 0.5                  VOC.py:169
    sigma: float = 0.0                 # random.uniform(0.15, 1.15)
    flip: bool = False                 # torch.rand(1) > 0.5                  VOC.py:174


def denormalise(x: torch.Tensor) -> torch.Tensor:
    """VOC.py:309-314: two torchvision normalize calls (fp32, in this order)."""
    inv_std = torch.tensor([1 / 0.229, 1 / 0.224, 1 / 0.225], dtype=torch.float32).view(3, 1, 1)
    neg_mean = torch.tensor([-0.485, -0.456, -0.406], dtype=torch.float32).view(3, 1, 1)
    x = (x.clone() - torch.zeros(3, 1, 1)) / inv_std
    return (x - neg_mean) / torch.ones(3, 1, 1)


def to_u8(t: torch.Tensor) -> np.ndarray:
    """to_pil_image on a float tensor: pic.mul(255).byte().  Out-of-range values are clamped here (the reference's .byte() wraps:
    undefined for real images, which are in range)."""
    return t.mul(255).clamp(0, 255).to(torch.uint8).numpy()


def tensors_to_pil(image, label, logits1, logits2):
    """tensor_to_pil_2 (VOC.py:284-291)."""
    img = Image.fromarray(np.ascontiguousarray(to_u8(denormalise(image.float())).transpose(1, 2, 0)))
    # labels: the second call of the reference feeds -1 back in; (-1/255*255).byte() wraps to 255 there (and 255 -> -1 again)
    lf = label.float()
    lab = Image.fromarray(to_u8(torch.where(lf < 0, torch.full_like(lf, 255.0), lf) / 255.0))
    l1 = Image.fromarray(to_u8(logits1.float()))
    l2 = Image.fromarray(to_u8(logits2.float()))
    return img, lab, l1, l2


def padded_size(h, w, scale, crop):
    rh, rw = int(h * scale), int(w * scale)
    return rh, rw, max(rh, crop[0]), max(rw, crop[1])


def transform_2(image, label, logits1, logits2, p: AugParams, crop_size, augmentation):
    """transform_2 (VOC.py:126-196) on PIL images with injected draws."""
    raw_w, raw_h = image.size
    rh, rw = int(raw_h * p.scale), int(raw_w * p.scale)
    image = image.resize((rw, rh), Image.BILINEAR)
    label = label.resize((rw, rh), Image.NEAREST)
    logits1 = logits1.resize((rw, rh), Image.NEAREST)
    logits2 = logits2.resize((rw, rh), Image.NEAREST)
    if crop_size[0] > rh or crop_size[1] > rw:
        right, bottom = max(crop_size[1] - rw, 0), max(crop_size[0] - rh, 0)
        image = Image.fromarray(np.pad(np.asarray(image), ((0, bottom), (0, right), (0, 0)), mode="reflect"))
        label = Image.fromarray(np.pad(np.asarray(label), ((0, bottom), (0, right)), mode="constant", constant_values=255))
        logits1 = Image.fromarray(np.pad(np.asarray(logits1), ((0, bottom), (0, right)), mode="constant", constant_values=0))
        logits2 = Image.fromarray(np.pad(np.asarray(logits2), ((0, bottom), (0, right)), mode="constant", constant_values=0))
    i, j, h, w = p.crop_i, p.crop_j, crop_size[0], crop_size[1]
    box = (j, i, j + w, i + h)
    image, label, logits1, logits2 = image.crop(box), label.crop(box), logits1.crop(box), logits2.crop(box)
    if augmentation:
        if p.jitter:
            image = color_jitter(image, p)
        if p.blur:
            from PIL import ImageFilter
            image = image.filter(ImageFilter.GaussianBlur(radius=p.sigma))
        if p.flip:
            image, label = image.transpose(Image.FLIP_LEFT_RIGHT), label.transpose(Image.FLIP_LEFT_RIGHT)
            logits1, logits2 = logits1.transpose(Image.FLIP_LEFT_RIGHT), logits2.transpose(Image.FLIP_LEFT_RIGHT)
    img = torch.from_numpy(np.asarray(image).copy()).permute(2, 0, 1).float().div(255)
    lab = (torch.from_numpy(np.asarray(label).copy()).float().div(255) * 255).long()
    lab[lab == 255] = -1
    l1 = torch.from_numpy(np.asarray(logits1).copy()).float().div(255)
    l2 = torch.from_numpy(np.asarray(logits2).copy()).float().div(255)
    img = (img - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
    return img, lab, l1, l2


def color_jitter(image, p: AugParams):
    """ColorJitter((0.75,1.25),(0.75,1.25),(0.75,1.25),(-0.25,0.25)) with fixed draws: the four PIL adjustments in ``p.order``."""
    from PIL import ImageEnhance
    for op in p.order:
        if op == 0:
            image = ImageEnhance.Brightness(image).enhance(p.brightness)
        elif op == 1:
            image = ImageEnhance.Contrast(image).enhance(p.contrast)
        elif op == 2:
            image = ImageEnhance.Color(image).enhance(p.saturation)
        else:
            h, s, v = image.convert("HSV").split()
            nh = np.array(h, dtype=np.uint8)
            with np.errstate(over="ignore"):
                nh = nh + np.uint8(int(p.hue * 255) & 0xFF)          # uint8 wrap-around, as torchvision's adjust_hue does
            image = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
    return image


def batch_transform_2(images, labels, logits_1, logits_2, params: Sequence[AugParams], crop_size, augmentation):
    """batch_transform_2 (VOC.py:339-352) on CPU tensors: images [B,3,H,W] fp32 normalised, labels [B,H,W] (0..K-1, 255 or -1),
    logits [B,H,W] fp32 in [0,1]."""
    outs = [[], [], [], []]
    for k in range(images.shape[0]):
        pil = tensors_to_pil(images[k], labels[k], logits_1[k], logits_2[k])
        res = transform_2(*pil, params[k], tuple(crop_size), augmentation)
        for o, r in zip(outs, res):
            o.append(r.unsqueeze(0))
    return tuple(torch.cat(o) for o in outs)


# ---- generate_cut_gather* across ranks (VOC.py:354-477) -----------------------------------------------------------------
def cutout_mask(h, w, ratio, rng):
    """generate_cutout_mask (VOC.py:518-534): 1 outside the box, 0 inside; three numpy draws."""
    area = h * w / ratio
    bw = rng.randint(w / ratio + 1, w)
    bh = np.round(area / bw)
    x0 = rng.randint(0, w - bw + 1)
    y0 = rng.randint(0, h - bh + 1)
    m = torch.ones(h, w)
    m[int(y0): int(y0 + bh), int(x0): int(x0 + bw)] = 0
    return m


def cut_gather_ranks(per_rank, mode, rngs):
    """What each of W ranks returns from generate_cut_gather_2/3 (VOC.py:393-434 / 436-477), restated for a list of per-rank tensor
    tuples (image [B,3,H,W], label maps int64 [B,H,W] ..., confidence maps [B,H,W] ...): the tensors are all-gathered, EVERY rank
    walks all W*B gathered images drawing one mask each from its own numpy stream (``rngs[r]``), the partner of gathered image i is
    ``gathered[(i + 1) % B]`` with B the LOCAL batch size - i.e. always an image of rank 0 - and rank r keeps block r.
    cutmix / cutout only (classmix draws from torch's global stream)."""
    world, b = len(per_rank), per_rank[0][0].shape[0]
    h, w = per_rank[0][0].shape[-2:]
    gathered = [torch.cat([per_rank[r][k] for r in range(world)]) for k in range(len(per_rank[0]))]
    outs = []
    for r in range(world):
        if mode == "none":
            outs.append(tuple(t[r * b:(r + 1) * b].clone() for t in gathered))
            continue
        new = [[] for _ in gathered]
        for i in range(world * b):
            if mode == "cutout":
                m = cutout_mask(h, w, 2, rngs[r])
                for k, t in enumerate(gathered):
                    if t.dtype == torch.int64:
                        v = t[i].clone()
                        v[(1 - m).bool()] = -1
                    else:
                        v = t[i] * m
                    new[k].append(v.unsqueeze(0))
                continue
            m = cutout_mask(h, w, 2, rngs[r])
            j = (i + 1) % b
            for k, t in enumerate(gathered):
                v = t[i] * m + t[j] * (1 - m)
                new[k].append((v.long() if t.dtype == torch.int64 else v).unsqueeze(0))
        outs.append(tuple(torch.cat(n)[r * b:(r + 1) * b] for n in new))
    return outs
