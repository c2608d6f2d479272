"""Model_cross.forward (ddp_model.py:184-239, 8 outputs, both pseudo-label maps, no agreement mask) and
Model_ori_pseudo.forward (:32-70, 7 outputs, class-space pseudo labels only) against the oracle's composition of the same steps
(teacher passes in train mode, similarity / soft-max / arg-max at full resolution, identity augmentation, student passes)."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import dev, rel_err  # noqa: E402

K, S, B = 21, 65, 2


def _setup(cls, **kw):
    from css_amd.networks import resnet
    from oracle import css_oracle as O
    cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [1.0, 1.0], "mix_mode": "none", "device_aug": "identity"}}
    m = cls(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, **kw)
    sd = O.init_state("tv", K, 256, 31, 0.25)          # well-conditioned random weights (bn3 gains x0.25), as the damped fixtures
    m.model.load_state_dict(sd)
    m.ema_model.load_state_dict(sd)
    m = m.to(dev())
    m.model.train()
    m.ema_model.train()
    g = torch.Generator().manual_seed(8)
    l, u = torch.randn(B, 3, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
    proto = torch.randn(K, 256, generator=g)
    return m, sd, l, u, proto


def _close(a, b, tol=1e-3):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert rel_err(a.float().cpu(), b) < tol, rel_err(a.float().cpu(), b)


def test_model_cross_forward_matches_oracle():
    from css_amd.networks.ddp_model import Model_cross
    from oracle import css_oracle as O
    m, sd, l, u, proto = _setup(Model_cross, temp=0.1)
    out = m(l.to(dev()), u.to(dev()), proto.to(dev()))
    assert len(out) == 8
    sd_t, sd_s = copy.deepcopy(sd), copy.deepcopy(sd)
    O.deeplab_forward(sd_t, l, "tv", True, K, 256)                       # teacher on the labeled batch: BN running statistics only
    pred_u, rep_u = O.deeplab_forward(sd_t, u, "tv", True, K, 256)
    logits_rep, labels_rep, logits_cls, labels_cls, _ = O.pseudo_labels_mix(pred_u, rep_u, proto, 0.1, (S, S), K)
    pred_l, rep_l = O.deeplab_forward(sd_s, l, "tv", True, K, 256)
    pred_u2, rep_u2 = O.deeplab_forward(sd_s, u, "tv", True, K, 256)    # identity augmentation, mix_mode none
    _close(out[0], F.interpolate(pred_l, size=(S, S), mode="bilinear", align_corners=True))
    _close(out[1], F.interpolate(pred_u2, size=(S, S), mode="bilinear", align_corners=True))
    assert out[2].dtype == torch.int64 and out[3].dtype == torch.int64
    assert (out[2].cpu() == labels_cls).float().mean() > 0.995          # arg-max flips only at near-ties of the random-init logits
    assert (out[3].cpu() == labels_rep).float().mean() > 0.995
    _close(out[4], logits_cls, 2e-3)
    _close(out[5], logits_rep, 2e-3)
    rep_all = torch.cat((rep_l, rep_u2))
    _close(out[6], rep_all)
    _close(out[7], O.prob_all_from_rep(rep_all, proto, 0.1), 2e-3)
    # teacher's running statistics moved twice (labeled, unlabeled pass), the student's twice as well
    _close(m.ema_model.resnet_bn1.running_mean, sd_t["resnet_bn1.running_mean"], 1e-4)
    _close(m.model.resnet_bn1.running_var, sd_s["resnet_bn1.running_var"], 1e-4)


def test_model_ori_pseudo_forward_matches_oracle():
    from css_amd.networks.ddp_model import Model_ori_pseudo
    from oracle import css_oracle as O
    m, sd, l, u, _ = _setup(Model_ori_pseudo)
    out = m(l.to(dev()), u.to(dev()))
    assert len(out) == 7
    sd_t, sd_s = copy.deepcopy(sd), copy.deepcopy(sd)
    pred_u, _ = O.deeplab_forward(sd_t, u, "tv", True, K, 256)          # the teacher sees the unlabeled batch only (ddp_model.py:35)
    raw = F.interpolate(pred_u, size=(S, S), mode="bilinear", align_corners=True)
    logits, labels = torch.max(torch.softmax(raw, dim=1), dim=1)
    pred_l, rep_l = O.deeplab_forward(sd_s, l, "tv", True, K, 256)
    pred_u2, rep_u2 = O.deeplab_forward(sd_s, u, "tv", True, K, 256)
    _close(out[0], F.interpolate(pred_l, size=(S, S), mode="bilinear", align_corners=True))
    _close(out[1], F.interpolate(pred_u2, size=(S, S), mode="bilinear", align_corners=True))
    assert (out[2].cpu() == labels).float().mean() > 0.995
    _close(out[3], logits, 2e-3)
    _close(out[4], torch.cat((rep_l, rep_u2)))
    _close(out[5], torch.cat((pred_l, pred_u2)))
    _close(out[6], raw)
    _close(m.ema_model.resnet_bn1.running_mean, sd_t["resnet_bn1.running_mean"], 1e-4)
