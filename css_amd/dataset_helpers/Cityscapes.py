"""Cityscapes split reader, dataset and id mapping.

Mirrors generalframeworks/dataset_helpers/Cityscapes.py: ``get_cityscapes_idx_via_txt`` :87-101, ``City_BuildData`` :66-85,
``Cityscapes_Dataset`` :40-64 (``Cityscapes_Dataset_cache`` :10-38 is the same item law with two unused constructor arguments),
``image_root_transform`` :223-225, ``label_root_transform`` :219-221, ``cityscapes_class_map`` :194-217; ``transform`` is the
VOC one (the two reference copies are line-for-line identical).  Directory layout: ``leftImg8bit/<split>/<city>/<id>.png``,
``gtFine/<split>/<city>/<id minus '_leftImg8bit'>_gtFine_trainIds.png`` with split 'train' | 'val'.
"""
from __future__ import annotations

import os

import numpy as np
import torch.utils.data as data
from PIL import Image

from .VOC import read_split, transform  # noqa: F401
from .gpu_aug import batch_transform  # noqa: F401

# labelId -> trainId of the 19 evaluated classes (cityscapesScripts labels.py); void ids -> 255; ids the reference's table
# does not list (none occur in gtFine) -> 0, as its zeros_like default does
_VOID_IDS = (0, 1, 2, 3, 4, 5, 6, 9, 10, 14, 15, 16, 18, 29, 30)
_TRAIN_IDS = (7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33)
_LUT = np.zeros(256, dtype=np.int64)
_LUT[list(_VOID_IDS)] = 255
for _t, _i in enumerate(_TRAIN_IDS):
    _LUT[_i] = _t


def cityscapes_class_map(mask):
    """Cityscapes.py:194-217 as one table lookup; result has the dtype of ``mask``."""
    m = np.asarray(mask)
    idx = m.astype(np.int64)
    inside = (idx >= 0) & (idx < 256)
    out = np.where(inside, _LUT[np.clip(idx, 0, 255)], 0)
    return out.astype(m.dtype)


def get_cityscapes_idx_via_txt(root, label_num, seed):
    return read_split(root, label_num, seed)


def image_root_transform(root: str, mode: str):
    """'<city>_<seq>_<frame>_leftImg8bit' -> ('/leftImg8bit/<mode>/<city>/<id>.png', city)."""
    city = root[0: root.find('_')]
    return f"/leftImg8bit/{mode}/{city}/{root}.png", city


def label_root_transform(root: str, name: str, mode: str):
    """Drops the 12-character '_leftImg8bit' suffix (after strip()) and points at the trainIds map."""
    return f"/gtFine/{mode}/{name}/{root.strip()[0:-12]}_gtFine_trainIds.png"


class Cityscapes_Dataset(data.Dataset):
    def __init__(self, root, idx_list, crop_size=(512, 512), scale_size=(0.5, 2.0), augmentation=True, train=True):
        self.root = os.path.expanduser(root)
        self.train = train
        self.crop_size = crop_size
        self.augmentation = augmentation
        self.scale_size = scale_size
        self.idx_list = idx_list

    def paths(self, index):
        mode = 'train' if self.train else 'val'
        image_rel, city = image_root_transform(self.idx_list[index], mode=mode)
        return self.root + image_rel, self.root + label_root_transform(self.idx_list[index], city, mode=mode)

    def __getitem__(self, index):
        ip, lp = self.paths(index)
        image, label = transform(Image.open(ip), Image.open(lp), None, self.crop_size, self.scale_size, self.augmentation)
        return image, label.squeeze(0)

    def __len__(self):
        return len(self.idx_list)


class Cityscapes_Dataset_cache(Cityscapes_Dataset):
    def __init__(self, root, idx_list, crop_size=(512, 512), scale_size=(0.5, 2.0), augmentation=True, train=True,
                 apply_partial=None, partial_seed=None):
        super().__init__(root, idx_list, crop_size, scale_size, augmentation, train)
        self.apply_partial = apply_partial
        self.partial_seed = partial_seed


class City_BuildData:
    """Cityscapes.py:66-85: all three sets at scale 1; only the labeled one is augmented."""

    def __init__(self, data_path, txt_path, label_num, seed, crop_size=[512, 512]):
        self.data_path = data_path
        self.txt_path = txt_path
        self.label_num = label_num
        self.seed = seed
        self.im_size = [512, 1024]
        self.crop_size = crop_size
        self.num_segments = 19
        self.scale_size = (1.0, 1.0)
        self.train_l_idx, self.train_u_idx, self.test_idx = get_cityscapes_idx_via_txt(self.txt_path, self.label_num, self.seed)

    def build(self):
        mk = Cityscapes_Dataset
        return (mk(self.data_path, self.train_l_idx, self.crop_size, self.scale_size, augmentation=True, train=True),
                mk(self.data_path, self.train_u_idx, self.crop_size, scale_size=(1.0, 1.0), augmentation=False, train=True),
                mk(self.data_path, self.test_idx, self.crop_size, scale_size=(1.0, 1.0), augmentation=False, train=False))
