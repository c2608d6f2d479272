"""SURVEY 8(f-4) on the GPU (VERDICT r02 missing 1): the data loaders feeding the device step - the path of
/root/reference/mix_label.py:36-60 (VOC_BuildData(...).build() -> three DataLoaders) into :162-196 (the train body).

A scratch VOC tree (JPEG images, palette-free PNG labels, split files) -> ``VOC_BuildData(...).build()`` -> ``DataLoader(num_workers=2,
drop_last=True)`` for the labeled and the unlabeled set -> one batch of each ->

* ``MixTrainer.step`` with the reference's in-step augmentation on the device (``device_aug='pil'``: random rescale, crop, cutmix, colour
  jitter, blur, flip - dataset_helpers/VOC.py:325-352,393-434): runs, finite positive losses, every parameter moved;
* the same batch with the identity augmentation and the oracle's recorded sampler draws injected, against ``oracle.train_step_mix`` on the
  CPU (fp32): supervised / unsupervised / contrastive loss, pseudo labels, prototypes.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import dev, rel_err  # noqa: E402


def _make_voc_tree(root, ids, size):
    """JPEG / PNG pairs with piecewise-constant labels (blocks of classes, some 255 = ignore): decodable by the loaders like VOC."""
    from PIL import Image
    os.makedirs(f"{root}/JPEGImages")
    os.makedirs(f"{root}/SegmentationClassAug")
    rng = np.random.RandomState(5)
    for n, name in enumerate(ids):
        h, w = size + 9 * (n % 3), size + 14 * ((n + 1) % 3)            # differently sized images: the CPU transform crops them
        img = (rng.rand(h, w, 3) * 255).astype(np.uint8)
        blocks = rng.randint(0, 21, ((h + 15) // 16, (w + 15) // 16)).astype(np.uint8)
        blocks[rng.rand(*blocks.shape) < 0.05] = 255
        lab = np.kron(blocks, np.ones((16, 16), np.uint8))[:h, :w]
        Image.fromarray(img).save(f"{root}/JPEGImages/{name}.jpg", quality=95)
        Image.fromarray(lab).save(f"{root}/SegmentationClassAug/{name}.png")


def _loaders(tmp_path, S, B):
    from css_amd.dataset_helpers import VOC
    root, txt = str(tmp_path / "voc"), str(tmp_path / "txt")
    ids = [f"2007_{i:06d}" for i in range(3 * B)]
    _make_voc_tree(root, ids, S)
    d = f"{txt}/662/3407"
    os.makedirs(d)
    for name, part in (("labeled_filename.txt", ids[:B]), ("unlabeled_filename.txt", ids[B:2 * B]), ("valid_filename.txt", ids[2 * B:])):
        with open(f"{d}/{name}", "w") as f:
            f.write("\n".join(part))
    data = VOC.VOC_BuildData(data_path=root, txt_path=txt, label_num=662, seed=3407, crop_size=[S, S])
    train_l, train_u, test = data.build()                                      # mix_label.py:36-41
    mk = lambda ds: torch.utils.data.DataLoader(ds, batch_size=B, drop_last=True, num_workers=2, shuffle=False)   # mix_label.py:42-59
    return mk(train_l), mk(train_u)


def _trainer(S, K, seed, gain, aug, mix, args, dtype=torch.float32):
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from css_amd.train_step import MixTrainer
    from oracle import css_oracle as O
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (0.5, 1.5) if aug == "pil" else (1.0, 1.0), "mix_mode": mix, "device_aug": aug}}
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.5)
    sd = O.init_state("tv", K, 256, seed, gain)
    m.model.load_state_dict(sd)
    m.ema_model.load_state_dict(sd)
    m = m.to(dev()).train().set_compute_dtype(dtype)
    return MixTrainer(m, K, lr=args["lr"], total_iter=100, num_queries=args["num_queries"], num_negatives=args["num_negatives"],
                      strong_threshold=args["strong_threshold"], weak_threshold=args["weak_threshold"], un_threshold=args["un_threshold"])


def test_dataloader_batch_through_the_device_step(tmp_path):
    from oracle import css_oracle as O
    K, S, B, seed, gain = 21, 65, 2, 7, 0.25
    torch.manual_seed(3)
    l_loader, u_loader = _loaders(tmp_path, S, B)
    l_img, l_lab = next(iter(l_loader))                                        # mix_label.py:162-165
    u_img, _ = next(iter(u_loader))
    assert l_img.shape == (B, 3, S, S) and l_lab.shape == (B, S, S) and l_lab.dtype == torch.int64 and u_img.shape == (B, 3, S, S)
    assert int(l_lab.min()) >= -1 and int(l_lab.max()) < K
    args = dict(lr=1e-3, temp_model=0.5, strong_threshold=0.8, weak_threshold=0.0, un_threshold=0.97, num_queries=64, num_negatives=128)

    # (1) the reference's pipeline: loader batch -> .cuda() -> step with the in-step augmentation on the device
    tr = _trainer(S, K, seed, gain, "pil", "cutmix", args)
    np.random.seed(0)
    p0 = tr.flat_p.clone()
    r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()))
    torch.cuda.synchronize()
    losses = {k: float(v) for k, v in r.items() if k != "pseudo"}
    print("loader -> step (device_aug=pil, cutmix):", losses)
    assert all(math.isfinite(v) for v in losses.values()) and bool(torch.isfinite(tr.flat_p).all())
    assert losses["sup"] > 0 and losses["contrast"] > 0
    assert (tr.flat_p != p0).float().mean().item() > 0.99
    del tr

    # (2) the same loader batch, identity augmentation, oracle draws injected: against the CPU oracle
    st = O.MixState("tv", K, 256, seed, gain)
    rec = {}
    torch.manual_seed(0)
    np.random.seed(0)
    ro = O.train_step_mix(st, l_img, l_lab, u_img, record=rec, **args)
    tr = _trainer(S, K, seed, gain, "identity", "none", args)
    r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()), _injected=dict(anchor=rec["anchor"], negative=rec["negative"]))
    for key in ("sup", "unsup", "contrast"):
        a, b = float(r[key]), float(ro[key])
        print(f"loader batch, {key}: hip {a:.6f} oracle {b:.6f}")
        if math.isnan(b):
            assert math.isnan(a), (key, a, b)
        else:
            assert abs(a - b) < 1e-3 * max(1.0, abs(b)), (key, a, b)
    assert (r["pseudo"].cpu() != ro["pseudo"]).float().mean().item() < 5e-3
    assert rel_err(tr.prototypes.cpu(), st.prototypes) < 1e-3
