"""Device-resident in-step augmentation against the PIL restatement of the reference's pipeline
(oracle/aug_oracle.py) on identical random draws.  Byte work: bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev  # noqa: E402


def _inputs(b, h, w, seed, k=21):
    from oracle import aug_oracle as A
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand(b, 3, h, w, generator=g) - torch.tensor(A.MEAN).view(1, 3, 1, 1)) / torch.tensor(A.STD).view(1, 3, 1, 1)
    lab = torch.randint(0, k, (b, h, w), generator=g).float()
    lab[torch.rand(b, h, w, generator=g) < 0.1] = 255.0          # teacher/indicator disagreement
    l1, l2 = torch.rand(b, h, w, generator=g), torch.rand(b, h, w, generator=g)
    return img, lab, l1, l2


GEOM_CASES = [
    # H, W, crop, scales per image
    (33, 41, (33, 41), [1.0, 0.5, 0.73, 1.5]),            # identity, strongest shrink (pad on both sides), shrink, enlarge
    (65, 65, (65, 65), [0.8, 1.0, 1.27, 2.0]),
    (40, 56, (32, 48), [0.9, 0.61, 1.0, 1.13]),           # crop smaller than the image
    (37, 29, (37, 29), [0.999, 1.001, 0.5001, 1.9999]),   # sizes that differ by one pixel from the input
]


@pytest.mark.parametrize("case", GEOM_CASES)
def test_rescale_pad_crop_quantise_bit_exact(case):
    from css_amd.dataset_helpers import gpu_aug
    from oracle import aug_oracle as A
    h, w, crop, scales = case
    img, lab, l1, l2 = _inputs(len(scales), h, w, seed=h * 100 + w)
    g = torch.Generator().manual_seed(7)
    ps_o, ps_d = [], []
    for s in scales:
        rh, rw = int(h * s), int(w * s)
        ph, pw = max(rh, crop[0]), max(rw, crop[1])
        ci = int(torch.randint(0, ph - crop[0] + 1, (1,), generator=g))
        cj = int(torch.randint(0, pw - crop[1] + 1, (1,), generator=g))
        ps_o.append(A.AugParams(scale=s, crop_i=ci, crop_j=cj))
        ps_d.append(gpu_aug.AugParams(scale=s, crop_i=ci, crop_j=cj))
    want = A.batch_transform_2(img, lab, l1, l2, ps_o, crop, augmentation=False)
    got = gpu_aug.device_batch_transform_2(img.to(dev()), lab.to(dev()), l1.to(dev()), l2.to(dev()), crop, None, False, params=ps_d)
    names = ("image", "label", "logits_cls", "logits_rep")
    for n, a, b in zip(names, got, want):
        assert a.shape == b.shape and a.dtype == b.dtype, n
        assert torch.equal(a.cpu(), b), (n, (a.cpu() != b).float().mean().item())


def test_flip_and_label_conventions():
    from css_amd.dataset_helpers import gpu_aug
    from oracle import aug_oracle as A
    img, lab, l1, l2 = _inputs(3, 21, 34, seed=3)
    lab[1] = -1.0                                            # second pass of the reference feeds -1 back in (wraps to 255 -> -1)
    ps_o = [A.AugParams(flip=True), A.AugParams(flip=False), A.AugParams(scale=1.2, crop_i=2, crop_j=3, flip=True)]
    ps_d = [gpu_aug.AugParams(**{k: getattr(p, k) for k in ("scale", "crop_i", "crop_j", "flip")}) for p in ps_o]
    want = A.batch_transform_2(img, lab, l1, l2, ps_o, (21, 34), augmentation=True)
    got = gpu_aug.device_batch_transform_2(img.to(dev()), lab.to(dev()), l1.to(dev()), l2.to(dev()), (21, 34), None, True, params=ps_d)
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    assert int(got[1][1].max()) == -1


def test_draw_laws():
    """The host-side draws follow the reference's distributions (VOC.py:129,154,162-181)."""
    from css_amd.dataset_helpers import gpu_aug
    import random
    rng, trng = random.Random(0), torch.Generator().manual_seed(0)
    ps = [gpu_aug.draw_params(64, 80, (64, 80), (0.5, 1.5), True, rng, trng) for _ in range(4000)]
    sc = torch.tensor([p.scale for p in ps])
    assert 0.5 <= float(sc.min()) and float(sc.max()) <= 1.5 and abs(float(sc.mean()) - 1.0) < 0.02
    assert abs(sum(p.jitter for p in ps) / 4000 - 0.8) < 0.03
    assert abs(sum(p.blur for p in ps) / 4000 - 0.5) < 0.03 and abs(sum(p.flip for p in ps) / 4000 - 0.5) < 0.03
    jit = [p for p in ps if p.jitter]
    for name, lo, hi in (("brightness", 0.75, 1.25), ("contrast", 0.75, 1.25), ("saturation", 0.75, 1.25), ("hue", -0.25, 0.25)):
        v = torch.tensor([getattr(p, name) for p in jit])
        assert lo <= float(v.min()) and float(v.max()) <= hi and abs(float(v.mean()) - (lo + hi) / 2) < 0.02
    assert len({p.order for p in jit}) == 24
    sg = torch.tensor([p.sigma for p in ps if p.blur])
    assert 0.15 <= float(sg.min()) and float(sg.max()) <= 1.15
    for p in ps:                                            # crop offsets stay inside the padded, rescaled image
        rh, rw = int(64 * p.scale), int(80 * p.scale)
        assert 0 <= p.crop_i <= max(rh, 64) - 64 and 0 <= p.crop_j <= max(rw, 80) - 80


def _copy_params(p):
    from css_amd.dataset_helpers import gpu_aug
    return gpu_aug.AugParams(**{k: getattr(p, k) for k in ("scale", "crop_i", "crop_j", "jitter", "order", "brightness", "contrast",
                                                           "saturation", "hue", "blur", "sigma", "flip")})


def test_colour_jitter_each_op_bit_exact():
    """Brightness / contrast / saturation (PIL blends) and the 8-bit HSV hue shift, one op at a time and all four orders mixed."""
    from css_amd.dataset_helpers import gpu_aug
    from oracle import aug_oracle as A
    import itertools
    import random
    rng = random.Random(5)
    orders = list(itertools.permutations(range(4)))
    ps = []
    for k in range(12):
        ps.append(A.AugParams(jitter=True, order=orders[(5 * k + 1) % 24], brightness=rng.uniform(0.75, 1.25), contrast=rng.uniform(0.75, 1.25),
                              saturation=rng.uniform(0.75, 1.25), hue=rng.uniform(-0.25, 0.25)))
    ps.append(A.AugParams(jitter=True, brightness=1.25, contrast=0.75, saturation=1.25, hue=-0.25))      # range ends
    ps.append(A.AugParams(jitter=True, brightness=0.75, contrast=1.25, saturation=0.75, hue=0.25))
    img, lab, l1, l2 = _inputs(len(ps), 24, 31, seed=9)
    img[0, :, :4] = img[0, 0:1, :4]                          # grey pixels (minc == maxc) and saturated ones
    img[1] = (torch.randint(0, 2, (3, 24, 31)).float() - torch.tensor(A.MEAN).view(3, 1, 1)) / torch.tensor(A.STD).view(3, 1, 1)
    want = A.batch_transform_2(img, lab, l1, l2, ps, (24, 31), augmentation=True)
    got = gpu_aug.device_batch_transform_2(img.to(dev()), lab.to(dev()), l1.to(dev()), l2.to(dev()), (24, 31), None, True,
                                           params=[_copy_params(p) for p in ps])
    diff = (got[0].cpu() != want[0])
    assert not diff.any(), (diff.float().mean().item(), diff.flatten(1).any(1).nonzero().flatten().tolist())


def test_gaussian_blur_bit_exact_and_full_pipeline():
    from css_amd.dataset_helpers import gpu_aug
    from oracle import aug_oracle as A
    import random
    rng = random.Random(8)
    ps = [A.AugParams(blur=True, sigma=s) for s in (0.15, 0.4, 0.77, 1.0, 1.15)]
    # everything at once, as the second call of the reference does (scale 1, augmentation on)
    for k in range(5):
        ps.append(A.AugParams(jitter=True, order=tuple(rng.sample(range(4), 4)), brightness=rng.uniform(0.75, 1.25), contrast=rng.uniform(0.75, 1.25),
                              saturation=rng.uniform(0.75, 1.25), hue=rng.uniform(-0.25, 0.25), blur=True, sigma=rng.uniform(0.15, 1.15),
                              flip=bool(k % 2)))
    img, lab, l1, l2 = _inputs(len(ps), 19, 27, seed=12)
    want = A.batch_transform_2(img, lab, l1, l2, ps, (19, 27), augmentation=True)
    got = gpu_aug.device_batch_transform_2(img.to(dev()), lab.to(dev()), l1.to(dev()), l2.to(dev()), (19, 27), None, True,
                                           params=[_copy_params(p) for p in ps])
    for n, a, b in zip(("image", "label", "logits_cls", "logits_rep"), got, want):
        d = a.cpu() != b
        assert not d.any(), (n, d.float().mean().item(), d.flatten(1).any(1).nonzero().flatten().tolist())


def test_model_mix_with_device_augmentation():
    """Model_mix.forward with config device_aug='pil': random rescale + crop + colour ops run on the device between teacher and
    student; outputs keep the reference's shapes / dtypes / label convention and the step is differentiable."""
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    K, S = 21, 65
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (0.5, 1.5), "mix_mode": "cutmix", "device_aug": "pil"}}
    torch.manual_seed(4)
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.25).to(dev())
    m.model.train()
    m.ema_model.train()
    g = torch.Generator().manual_seed(6)
    l, u = torch.randn(2, 3, S, S, generator=g).to(dev()), torch.randn(2, 3, S, S, generator=g).to(dev())
    proto = torch.randn(K, 256, generator=g).to(dev())
    pl, pu, u_lab, u_lc, u_lr, rep_all, prob_all = m(l, u, proto)
    assert pl.shape == (2, K, S, S) and pu.shape == (2, K, S, S) and rep_all.shape[0] == 4
    assert u_lab.dtype == torch.int64 and u_lab.shape == (2, S, S) and int(u_lab.min()) >= -1 and int(u_lab.max()) < K
    for t in (u_lc, u_lr):                                   # confidence maps went through the 8-bit round trip
        assert t.dtype == torch.float32 and float(t.min()) >= 0 and float(t.max()) <= 1
        assert float(((t * 255) - (t * 255).round()).abs().max()) < 1e-4
    (pl.mean() + pu.mean() + rep_all.pow(2).mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.model.parameters())


def test_mixing_modes_on_device():
    """cutmix / cutout boxes follow generate_cutout_mask (half the area, VOC.py:518-534); classmix keeps a random half of the
    image's classes and takes the partner (i+1) % B elsewhere (VOC.py:505-510,430-437)."""
    import numpy as np
    from css_amd.dataset_helpers import gpu_aug
    g = torch.Generator().manual_seed(1)
    b, h, w = 3, 20, 28
    img = torch.randn(b, 3, h, w, generator=g).to(dev())
    lab = torch.randint(0, 6, (b, h, w), generator=g).to(dev())
    l1, l2 = torch.rand(b, h, w, generator=g).to(dev()), torch.rand(b, h, w, generator=g).to(dev())
    for mode in ("cutmix", "cutout"):
        oi, ol, o1, o2 = gpu_aug.generate_cut_gather_2(img, lab, l1, l2, mode=mode, rng=np.random.RandomState(3))
        for i in range(b):
            changed = (ol[i] != lab[i]) | (o1[i] != l1[i])
            ys, xs = changed.nonzero(as_tuple=True)
            area = (int(ys.max()) - int(ys.min()) + 1) * (int(xs.max()) - int(xs.min()) + 1)
            assert abs(area - h * w / 2) <= w                      # the box covers half the image (up to rounding of its height)
            if mode == "cutout":
                assert int(ol[i][changed].max()) == -1 and float(oi[i][:, changed].abs().max()) == 0
            else:
                j = (i + 1) % b
                assert torch.equal(oi[i][:, changed], img[j][:, changed]) and torch.equal(ol[i][changed], lab[j][changed])
    torch.manual_seed(5)
    oi, ol, o1, o2 = gpu_aug.generate_cut_gather_2(img, lab, l1, l2, mode="classmix")
    for i in range(b):
        j = (i + 1) % b
        own = (ol[i] == lab[i]) & (o1[i] == l1[i])
        kept = torch.unique(lab[i][(o1[i] == l1[i]) & (o1[i] != l1[j])])
        assert len(kept) == len(torch.unique(lab[i])) // 2            # half of the image's classes survive
        other = ~((o1[i] == l1[i]) & (o1[i] != l1[j]))
        assert torch.equal(o2[i][other], l2[j][other]) and torch.equal(oi[i][:, other], img[j][:, other])


def test_mix_boxes_kernel_equals_the_indexing_path():
    """css_mix_boxes (one launch per tensor for the boxes of the whole batch) against the per-image indexing path the CPU tensors take
    (generate_cut_gather*, VOC.py:354-477): the same draws, the same bits - cutmix and cutout, the 2-, 3- and 1-label forms, a batch whose
    size does not divide anything, int64 labels with -1."""
    import numpy as np
    from css_amd.dataset_helpers import gpu_aug
    g = torch.Generator().manual_seed(4)
    b, h, w = 5, 37, 45
    img = torch.randn(b, 3, h, w, generator=g)
    lab = torch.randint(-1, 21, (b, h, w), generator=g)
    l1, l2 = torch.rand(b, h, w, generator=g), torch.rand(b, h, w, generator=g)
    for mode in ("cutmix", "cutout"):
        for fn, args in ((gpu_aug.generate_cut_gather_2, (img, lab, l1, l2)), (gpu_aug.generate_cut_gather_3, (img, lab, lab.flip(0), l1, l2)),
                         (gpu_aug.generate_cut_gather, (img, lab, l1))):
            want = fn(*[t.clone() for t in args], mode=mode, rng=np.random.RandomState(11))
            got = fn(*[t.to(dev()) for t in args], mode=mode, rng=np.random.RandomState(11))
            for a, c in zip(want, got):
                assert a.dtype == c.dtype and torch.equal(a, c.cpu()), (mode, fn.__name__)
