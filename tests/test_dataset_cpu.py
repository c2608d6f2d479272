"""split readers, datasets and the labeled-image CPU transform (css_amd/dataset_helpers/{VOC,Cityscapes,pil_ops}.py).

* torchvision-free reference functions: golden fixture tests/golden/dataset_helpers.json (made by importing the reference,
  tests/golden/make_golden.py::gen_dataset);
* ``transform``: against the PIL oracle (oracle/aug_oracle.py, the restatement of the reference's transform_2, whose image /
  label / logits legs are the same operations) on identical injected draws, bit-exact;
* the ORDER in which random numbers are consumed: replayed against the literal call sequence of VOC.py:64-112 +
  torchvision-0.8.2 RandomCrop.get_params / ColorJitter.forward on the global generators;
* end to end on a scratch VOC / Cityscapes tree through torch DataLoader.
"""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from css_amd.dataset_helpers import VOC, Cityscapes, pil_ops  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_helpers.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_cityscapes_class_map_matches_reference(gold):
    assert Cityscapes.cityscapes_class_map(np.arange(256, dtype=np.uint8)).tolist() == gold["class_map_u8"]
    odd = np.array(gold["odd_in"], dtype=np.int16)
    got = Cityscapes.cityscapes_class_map(odd)
    assert got.dtype == odd.dtype and got.tolist() == gold["class_map_odd"]
    tile = np.array(gold["tile_in"], dtype=np.uint8)
    got = Cityscapes.cityscapes_class_map(tile)
    assert got.shape == tile.shape and got.dtype == np.uint8 and got.tolist() == gold["class_map_tile"]


def test_cityscapes_paths_match_reference(gold):
    for p in gold["paths"]:
        img, city = Cityscapes.image_root_transform(p["name"], mode=p["mode"])
        assert (img, city) == (p["image"], p["city"])
        assert Cityscapes.label_root_transform(p["name"], city, mode=p["mode"]) == p["label"]


def _write_split(d, files):
    os.makedirs(f"{d}/662/3407")
    for k, v in files.items():
        with open(f"{d}/662/3407/{k}", "w", newline="") as f:
            f.write(v)


def test_split_readers_and_builders_match_reference(gold, tmp_path):
    d = str(tmp_path)
    _write_split(d, gold["split_files"])
    assert [list(x) for x in VOC.get_pascal_idx_via_txt(d, 662, 3407)] == gold["split_voc"]
    assert [list(x) for x in Cityscapes.get_cityscapes_idx_via_txt(d, "662", "3407")] == gold["split_city"]
    bd = VOC.VOC_BuildData(data_path="/data/voc", txt_path=d, label_num=662, seed=3407, crop_size=[321, 321])
    sets = [dict(root=s.root, n=len(s), crop=list(s.crop_size), scale=list(s.scale_size), aug=s.augmentation, train=s.train)
            for s in bd.build()]
    assert sets == gold["voc_sets"]
    assert dict(image_size=bd.image_size, num_segments=bd.num_segments, scale_size=list(bd.scale_size)) == gold["voc_attrs"]
    cd = Cityscapes.City_BuildData(data_path="~/city", txt_path=d, label_num=662, seed=3407, crop_size=[769, 769])
    csets = [dict(root=s.root.replace(os.path.expanduser("~"), gold["home"], 1), n=len(s), crop=list(s.crop_size),
                  scale=list(s.scale_size), aug=s.augmentation, train=s.train) for s in cd.build()]
    assert csets == gold["city_sets"]
    assert dict(im_size=cd.im_size, num_segments=cd.num_segments, scale_size=list(cd.scale_size)) == gold["city_attrs"]
    with pytest.raises(FileNotFoundError):
        VOC.get_pascal_idx_via_txt(d, 1, 1)


def _pil_inputs(h, w, seed, mode="L"):
    rng = np.random.RandomState(seed)
    # smooth-ish image so that resampling / blur differences would show, blocky labels with an ignore band
    base = rng.randint(0, 256, size=(h // 4 + 2, w // 4 + 2, 3)).astype(np.uint8)
    img = Image.fromarray(base).resize((w, h), Image.BICUBIC)
    lab = rng.randint(0, 21, size=(h // 8 + 1, w // 8 + 1)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w].copy()
    lab[:, : w // 10] = 255
    lab_img = Image.fromarray(lab, "L")
    if mode == "P":
        lab_img = lab_img.convert("P")
        lab_img.putpalette([v for i in range(256) for v in (i, (i * 7) % 256, (i * 13) % 256)])
        assert np.array_equal(np.asarray(lab_img), lab)
    logit = Image.fromarray(rng.randint(0, 256, size=(h, w)).astype(np.uint8), "L")
    return img, lab_img, logit


def _as_oracle_params(d):
    from oracle import aug_oracle as A
    return A.AugParams(scale=d.scale, crop_i=d.crop_i, crop_j=d.crop_j, jitter=d.jitter, order=tuple(d.order), brightness=d.brightness,
                       contrast=d.contrast, saturation=d.saturation, hue=d.hue, blur=d.blur, sigma=d.sigma, flip=d.flip)


@pytest.mark.parametrize("case", [
    dict(h=75, w=100, crop=(64, 64), scale=(0.5, 1.5), aug=True, mode="L"),
    dict(h=75, w=100, crop=(97, 129), scale=(0.5, 1.5), aug=True, mode="P"),      # always padded
    dict(h=60, w=91, crop=(60, 91), scale=(1.0, 1.0), aug=False, mode="L"),        # nothing can move: no crop draws
    dict(h=48, w=64, crop=(33, 80), scale=(0.5, 2.0), aug=True, mode="L"),
])
def test_transform_bit_exact_vs_pil_oracle(case):
    from oracle import aug_oracle as A
    for seed in range(12):
        img, lab, logit = _pil_inputs(case["h"], case["w"], seed, case["mode"])
        src = pil_ops.DrawSource(seed=1000 + seed)
        d = VOC.draw((case["h"], case["w"]), case["crop"], case["scale"], case["aug"], src)
        got = VOC.transform(img, lab, logit, crop_size=case["crop"], scale_size=case["scale"], augmentation=case["aug"], draws=d)
        lab_l = lab if lab.mode == "L" else Image.fromarray(np.asarray(lab), "L")
        exp = A.transform_2(img, lab_l, logit, logit, _as_oracle_params(d), case["crop"], case["aug"])
        assert got[0].shape == (3,) + case["crop"] and got[0].dtype == torch.float32
        assert got[1].shape == (1,) + case["crop"] and got[1].dtype == torch.int64
        assert got[2].shape == (1,) + case["crop"] and got[2].dtype == torch.float32
        assert torch.equal(got[0], exp[0]), (case, seed)
        assert torch.equal(got[1][0], exp[1]), (case, seed)
        assert torch.equal(got[2][0], exp[2]), (case, seed)
        assert set(torch.unique(got[1]).tolist()) <= set(range(-1, 21))
        two = VOC.transform(img, lab, None, crop_size=case["crop"], scale_size=case["scale"], augmentation=case["aug"], draws=d)
        assert len(two) == 2 and torch.equal(two[0], got[0]) and torch.equal(two[1], got[1])


def test_padding_conventions():
    """Image smaller than the crop: reflect padding for the image, ignore (-1) for the label, 0 for the logits (VOC.py:141-147)."""
    img, lab, logit = _pil_inputs(40, 50, 3)
    d = pil_ops.Draws(scale=1.0)
    im, lb, lg = VOC.transform(img, lab, logit, crop_size=(64, 70), scale_size=(1.0, 1.0), augmentation=False, draws=d)
    assert (lb[0, 40:, :] == -1).all() and (lb[0, :, 50:] == -1).all()
    assert (lg[0, 40:, :] == 0).all() and (lg[0, :, 50:] == 0).all()
    raw = pil_ops.normalize(pil_ops.to_tensor(img))
    assert torch.equal(im[:, :40, :50], raw)
    assert torch.equal(im[:, 40:64, :50], raw[:, 15:39, :].flip(1))      # rows 38..15 mirrored about the last row
    assert torch.equal(im[:, :40, 50:70], raw[:, :, 29:49].flip(2))


def _reference_draw_sequence(raw_hw, crop, scale_size, augmentation):
    """The literal sequence of generator calls of VOC.py:64-112 with torchvision 0.8.2 (global generators)."""
    out = {}
    out["scale"] = random.uniform(scale_size[0], scale_size[1])
    rh, rw = int(raw_hw[0] * out["scale"]), int(raw_hw[1] * out["scale"])
    h, w = max(rh, crop[0]), max(rw, crop[1])
    if w == crop[1] and h == crop[0]:
        out["ij"] = (0, 0)
    else:
        i = torch.randint(0, h - crop[0] + 1, size=(1,)).item()
        j = torch.randint(0, w - crop[1] + 1, size=(1,)).item()
        out["ij"] = (i, j)
    if augmentation:
        if torch.rand(1) > 0.2:
            fn_idx = torch.randperm(4)
            vals = {}
            for fn_id in fn_idx:
                lo, hi = ((0.75, 1.25), (0.75, 1.25), (0.75, 1.25), (-0.25, 0.25))[int(fn_id)]
                vals[int(fn_id)] = torch.tensor(1.0).uniform_(lo, hi).item()
            out["jitter"] = (tuple(int(i) for i in fn_idx), vals)
        if torch.rand(1) > 0.5:
            out["sigma"] = random.uniform(0.15, 1.15)
        out["flip"] = bool(torch.rand(1) > 0.5)
    return out


@pytest.mark.parametrize("aug", [True, False])
def test_draws_replay_the_reference_generator_sequence(aug):
    for seed in range(40):
        torch.manual_seed(seed)
        random.seed(seed)
        ref = _reference_draw_sequence((75, 100), (64, 64), (0.5, 1.5), aug)
        ref_next = (random.random(), float(torch.rand(1)))
        torch.manual_seed(seed)
        random.seed(seed)
        d = VOC.draw((75, 100), (64, 64), (0.5, 1.5), aug, pil_ops.DrawSource())
        assert (random.random(), float(torch.rand(1))) == ref_next          # both generators advanced identically
        assert d.scale == ref["scale"] and (d.crop_i, d.crop_j) == ref["ij"]
        assert d.jitter == ("jitter" in ref) and d.blur == ("sigma" in ref) and d.flip == ref.get("flip", False)
        if d.jitter:
            order, vals = ref["jitter"]
            assert tuple(d.order) == order
            assert (d.brightness, d.contrast, d.saturation, d.hue) == tuple(vals[k] for k in range(4))
        if d.blur:
            assert d.sigma == ref["sigma"]


def _make_voc_tree(root, ids, rng):
    os.makedirs(f"{root}/JPEGImages")
    os.makedirs(f"{root}/SegmentationClassAug")
    for n, name in enumerate(ids):
        h, w = 60 + 7 * n, 90 - 5 * n
        img, lab, _ = _pil_inputs(h, w, 100 + n)
        img.save(f"{root}/JPEGImages/{name}.jpg", quality=95)
        lab.save(f"{root}/SegmentationClassAug/{name}.png")


def test_voc_end_to_end_through_dataloader(tmp_path):
    root, txt = str(tmp_path / "voc"), str(tmp_path / "txt")
    ids = [f"2007_{i:06d}" for i in range(6)]
    _make_voc_tree(root, ids, None)
    _write_split(txt, {"labeled_filename.txt": "\n".join(ids[:2]), "unlabeled_filename.txt": "\n".join(ids[2:]),
                       "valid_filename.txt": "\n".join(ids[4:])})
    data = VOC.VOC_BuildData(data_path=root, txt_path=txt, label_num=662, seed=3407, crop_size=[65, 65])
    train_l, train_u, test = data.build()
    assert (len(train_l), len(train_u), len(test)) == (2, 4, 2)
    for ds in (train_l, train_u, test):
        loader = torch.utils.data.DataLoader(ds, batch_size=2, drop_last=True, num_workers=0)
        image, label = next(iter(loader))
        assert image.shape == (2, 3, 65, 65) and image.dtype == torch.float32
        assert label.shape == (2, 65, 65) and label.dtype == torch.int64
        assert int(label.min()) >= -1 and int(label.max()) <= 20
        assert torch.isfinite(image).all()
    # the unlabeled set: scale 1, no augmentation -> a pure crop of the padded decode, reproducible from the draws
    img = Image.open(train_u.paths(0)[0])
    lab = Image.open(train_u.paths(0)[1])
    torch.manual_seed(5)
    random.seed(5)
    a = train_u[0]
    torch.manual_seed(5)
    random.seed(5)
    d = VOC.draw((img.size[1], img.size[0]), (65, 65), (1.0, 1.0), False, pil_ops.DrawSource())
    b = VOC.transform(img, lab, None, crop_size=[65, 65], scale_size=(1.0, 1.0), augmentation=False, draws=d)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1][0])
    with pytest.raises(FileNotFoundError):
        VOC.Pascal_VOC_Dataset(root, ["missing"], (65, 65))[0]


def test_cityscapes_end_to_end(tmp_path):
    root, txt = str(tmp_path / "city"), str(tmp_path / "txt")
    names = {"train": ["aachen_000000_000019_leftImg8bit", "bochum_000000_000313_leftImg8bit"], "val": ["frankfurt_000000_000294_leftImg8bit"]}
    for mode, lst in names.items():
        for n, name in enumerate(lst):
            city = name.split("_")[0]
            os.makedirs(f"{root}/leftImg8bit/{mode}/{city}", exist_ok=True)
            os.makedirs(f"{root}/gtFine/{mode}/{city}", exist_ok=True)
            img, lab, _ = _pil_inputs(64, 128, 200 + n)
            arr = np.asarray(lab).copy()
            arr[arr > 18] = 255
            img.save(f"{root}/leftImg8bit/{mode}/{city}/{name}.png")
            Image.fromarray(arr, "L").save(f"{root}/gtFine/{mode}/{city}/{name[:-12]}_gtFine_trainIds.png")
    _write_split(txt, {"labeled_filename.txt": names["train"][0], "unlabeled_filename.txt": names["train"][1],
                       "valid_filename.txt": names["val"][0]})
    train_l, train_u, test = Cityscapes.City_BuildData(root, txt, 662, 3407, crop_size=[48, 96]).build()
    for ds in (train_l, train_u, test):
        image, label = ds[0]
        assert image.shape == (3, 48, 96) and label.shape == (48, 96) and label.dtype == torch.int64
        assert int(label.min()) >= -1 and int(label.max()) <= 18
    cached = Cityscapes.Cityscapes_Dataset_cache(root, names["train"], (48, 96), (1.0, 1.0), False, True, apply_partial=0.5, partial_seed=1)
    assert len(cached) == 2 and cached[1][0].shape == (3, 48, 96)


def test_compat_aliases_dataset_helpers():
    import css_amd.compat
    css_amd.compat.install()
    import importlib
    v = importlib.import_module("generalframeworks.dataset_helpers.VOC")
    c = importlib.import_module("generalframeworks.dataset_helpers.Cityscapes")
    assert v.VOC_BuildData is VOC.VOC_BuildData and c.City_BuildData is Cityscapes.City_BuildData
    for name in ("batch_transform_2", "batch_transform_3", "generate_cut_gather_2", "generate_cut_gather_3", "transform"):
        assert hasattr(v, name)
