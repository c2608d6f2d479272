"""The train bodies of the reference's two OTHER entry scripts, run VERBATIM through the drop-in surface (no fused trainer):
cross_label.py:162-198 (warm-up branch: class-predictor pseudo labels) and ori_pseudo.py:158-187 - Model_cross /
Model_ori_pseudo forward, CrossEntropyLoss(ignore_index=-1), Attention_Threshold_Loss, label_onehot + nearest down-sampling,
Contrast_Loss(rep, label, mask, prob, prototypes), zero_grad / backward / torch.optim.SGD(nesterov).step / ema_update / PolyLR -
against one iteration captured from the reference itself (tests/golden/train_trace_{cross,ori}.npz, made by
make_golden.py::gen_train_trace_w5 with identity augmentation and the reference's recorded sampler draws injected).  fp32 path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import dev, rel_err  # noqa: E402

K, S = 21, 65
PROBES = ["resnet_conv1.weight", "resnet_layer3.10.conv2.weight", "classifier.3.weight", "representation.3.bias"]


def T(a):
    return torch.from_numpy(np.asarray(a))


def probe_slice(t):
    return t.detach().flatten()[:: max(1, t.numel() // 2048)][:2048]


def _injection(g, rep_all, label_all, mask_all, prob_all, protos):
    """Line the recorded draws (one entry per class that HAD hard pixels in the reference run) up with the valid classes."""
    from oracle import css_oracle as O
    rec = {}
    O.contrast_loss(rep_all.detach().float().cpu(), label_all.cpu(), mask_all.cpu(), prob_all.detach().float().cpu(), protos.cpu().clone(),
                    64, 128, 0.5, 0.8, 0.99, record=rec)
    anchors, negs, j = [], [], 0
    for hn in rec["hard_num"]:
        if hn > 0:
            anchors.append(g[f"anchor{j}"].astype(np.int64))
            negs.append(g[f"negative{j}"].astype(np.int64))
            j += 1
        else:
            anchors.append(None)
            negs.append(None)
    assert j == int(g["n_anchor"]), (j, int(g["n_anchor"]))
    return dict(anchor=anchors, negative=negs)


@pytest.mark.parametrize("kind", ["cross", "ori"])
def test_entry_script_train_body_vs_reference_trace(golden, kind):
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_cross, Model_ori_pseudo
    from css_amd.loss.loss import Attention_Threshold_Loss, Contrast_Loss, CrossEntropyLoss
    from css_amd.scheduler.my_lr_scheduler import PolyLR
    from css_amd.utils import label_onehot
    from oracle import css_oracle as O
    g = golden(f"train_trace_{kind}")
    seed, gain, weak = int(g["seed"]), float(g["residual_gain"]), float(g["weak"])
    config = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none", "device_aug": "identity"}, "Loss": {"weak_threshold": weak}}
    if kind == "cross":
        model = Model_cross(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=config, temp=0.5)
    else:
        model = Model_ori_pseudo(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=config)
    sd = O.init_state("tv", K, 256, seed, gain)
    model.model.load_state_dict(sd, strict=True)
    model.ema_model.load_state_dict(sd, strict=True)
    model = model.to(dev())
    model.model.train()
    model.ema_model.train()
    criterion = {"ce_loss": CrossEntropyLoss(ignore_index=-1), "unsup_loss": Attention_Threshold_Loss(0.97),
                 "contrast_loss": Contrast_Loss(strong_threshold=0.8, num_queries=64, num_negatives=128, temp=0.5, alpha=0.99)}
    optimizer = torch.optim.SGD(model.model.parameters(), lr=6.4e-3, weight_decay=5e-4, momentum=0.9, nesterov=True)
    scheduler = PolyLR(optimizer, 100, min_lr=1e-4)
    prototypes = torch.zeros(K, 256, device=dev())
    train_l_image, train_u_image = T(g["l_img"]).to(dev()), T(g["u_img"]).to(dev())
    train_l_label = T(g["l_lab"]).long().to(dev())
    num_class = K

    # ---- the loop body, as in the reference -------------------------------------------------------------------
    if kind == "cross":
        (pred_l_large, pred_u_large, train_u_aug_label_cls, train_u_aug_label_rep, train_u_aug_logits_cls, train_u_aug_logits_rep, rep_all,
         pred_all) = model(train_l_image, train_u_image, prototypes)
        train_u_aug_label, train_u_aug_logits = train_u_aug_label_cls, train_u_aug_logits_cls          # epoch < args.warmup
    else:
        pred_l_large, pred_u_large, train_u_aug_label, train_u_aug_logits, rep_all, pred_all, pred_u_large_raw = model(train_l_image, train_u_image)
    sup_loss = criterion["ce_loss"](pred_l_large, train_l_label)
    unsup_loss = criterion["unsup_loss"](pred_u_large, train_u_aug_label, train_u_aug_logits)
    with torch.no_grad():
        train_u_aug_mask = train_u_aug_logits.ge(weak).float()
        mask_all = torch.cat(((train_l_label.unsqueeze(1) >= 0).float(), train_u_aug_mask.unsqueeze(1)))
        mask_all = F.interpolate(mask_all, size=pred_all.shape[2:], mode="nearest")
        label_l = F.interpolate(label_onehot(train_l_label, num_class), size=pred_all.shape[2:], mode="nearest")
        label_u = F.interpolate(label_onehot(train_u_aug_label, num_class), size=pred_all.shape[2:], mode="nearest")
        label_all = torch.cat((label_l, label_u))
        prob_all = pred_all if kind == "cross" else torch.softmax(pred_all, dim=1)
    inj = _injection(g, rep_all, label_all, mask_all, prob_all, prototypes)
    contrast_loss = criterion["contrast_loss"](rep_all, label_all, mask_all, prob_all, prototypes, _injected=inj)
    total_loss = sup_loss + unsup_loss + contrast_loss
    optimizer.zero_grad()
    total_loss.backward()
    optimizer.step()
    model.ema_update()
    scheduler.step()
    # -----------------------------------------------------------------------------------------------------------

    fails = []
    for name, val, tol in (("sup", sup_loss, 2e-3), ("unsup", unsup_loss, 3e-2), ("con", contrast_loss, 2e-3)):
        ref = float(g[name])
        print(f"{kind} {name}: hip {val.item():.6f} reference {ref:.6f}")
        if not abs(val.item() - ref) < tol * max(1.0, abs(ref)):
            fails.append((name, val.item(), ref))
    assert not fails, fails
    mism = (train_u_aug_label.cpu() != T(g["ulab"]).long()).float().mean().item()
    print(f"{kind}: pseudo-label mismatch fraction {mism:.2e}")
    assert mism < 1e-3
    if kind == "cross":
        assert (train_u_aug_label_rep.cpu() != T(g["ulab_rep"]).long()).float().mean().item() < 1e-3
    assert rel_err(train_u_aug_logits.float().cpu(), T(g["ulc"])) < 2e-3
    e = rel_err(prototypes.cpu(), T(g["protos"]))
    print(f"{kind}: prototypes rel err {e:.2e}")
    assert e < 2e-3
    assert abs(optimizer.param_groups[0]["lr"] - float(g["lr_next"])) < 1e-9
    sdm, sde = model.model.state_dict(), model.ema_model.state_dict()
    for p in PROBES:
        es = rel_err(probe_slice(sdm[p]).float().cpu(), T(g[f"student::{p}"]))
        et = rel_err(probe_slice(sde[p]).float().cpu(), T(g[f"teacher::{p}"]))
        print(f"{kind} {p}: student {es:.2e} teacher {et:.2e}")
        # parameters after the update are dominated by lr*grad; the gradient carries the ReLU-flip noise of test_network_gpu.py
        assert es < 3e-2 and et < 3e-2
    assert rel_err(sde["resnet_bn1.running_mean"].float().cpu(), T(g["teacher_rm::resnet_bn1"])) < 1e-3
    assert all(torch.isfinite(q).all() for q in model.model.parameters())


def _build(kind, g):
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_cross, Model_ori_pseudo
    from oracle import css_oracle as O
    seed, gain, weak = int(g["seed"]), float(g["residual_gain"]), float(g["weak"])
    config = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none", "device_aug": "identity"}, "Loss": {"weak_threshold": weak}}
    if kind == "cross":
        model = Model_cross(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=config, temp=0.5)
    else:
        model = Model_ori_pseudo(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=config)
    sd = O.init_state("tv", K, 256, seed, gain)
    model.model.load_state_dict(sd, strict=True)
    model.ema_model.load_state_dict(sd, strict=True)
    model = model.to(dev())
    model.model.train()
    model.ema_model.train()
    return model, weak


@pytest.mark.parametrize("kind", ["cross", "ori"])
def test_fused_trainers_vs_reference_trace(golden, kind):
    """CrossTrainer / OriTrainer (css_amd/train_step.py: the same bodies with the fused losses, the class-id map instead of one-hot
    tensors, direct gradient accumulation and the fused SGD+EMA kernel) against the same captured iteration."""
    from css_amd.train_step import CrossTrainer, OriTrainer
    from css_amd.utils import label_onehot
    g = golden(f"train_trace_{kind}")
    l_img, u_img, l_lab = T(g["l_img"]).to(dev()), T(g["u_img"]).to(dev()), T(g["l_lab"]).long().to(dev())
    # the sampler draws of the reference run, lined up through a plain forward of a twin model
    twin, weak = _build(kind, g)
    with torch.no_grad():
        protos0 = torch.zeros(K, 256, device=dev())
        if kind == "cross":
            _, _, ulab, _, ulc, _, rep_all, prob_all = twin(l_img, u_img, protos0)
        else:
            _, _, ulab, ulc, rep_all, pred_all, _ = twin(l_img, u_img)
            prob_all = torch.softmax(pred_all, dim=1)
        mask_all = torch.cat(((l_lab.unsqueeze(1) >= 0).float(), ulc.ge(weak).float().unsqueeze(1)))
        mask_all = F.interpolate(mask_all, size=prob_all.shape[2:], mode="nearest")
        label_all = torch.cat((F.interpolate(label_onehot(l_lab, K), size=prob_all.shape[2:], mode="nearest"),
                               F.interpolate(label_onehot(ulab, K), size=prob_all.shape[2:], mode="nearest")))
        inj = _injection(g, rep_all, label_all, mask_all, prob_all, protos0)
    del twin
    model, weak = _build(kind, g)
    cls = CrossTrainer if kind == "cross" else OriTrainer
    tr = cls(model, K, lr=6.4e-3, total_iter=100, min_lr=1e-4, num_queries=64, num_negatives=128, strong_threshold=0.8,
             weak_threshold=weak, un_threshold=0.97)
    r = tr.step(l_img, l_lab, u_img, ramp=1.0, _injected=inj)
    fails = []
    for name, key, tol in (("sup", "sup", 2e-3), ("unsup", "unsup", 3e-2), ("con", "contrast", 2e-3)):
        ref = float(g[name])
        print(f"fused {kind} {name}: hip {r[key].item():.6f} reference {ref:.6f}")
        if not abs(r[key].item() - ref) < tol * max(1.0, abs(ref)):
            fails.append((name, r[key].item(), ref))
    assert not fails, fails
    assert (r["pseudo"].cpu() != T(g["ulab"]).long()).float().mean().item() < 1e-3
    e = rel_err(tr.prototypes.cpu(), T(g["protos"]))
    print(f"fused {kind}: prototypes rel err {e:.2e}")
    assert e < 2e-3
    assert abs(tr.lr - float(g["lr_next"])) < 1e-9
    sdm, sde = model.model.state_dict(), model.ema_model.state_dict()
    for p in PROBES:
        es = rel_err(probe_slice(sdm[p]).float().cpu(), T(g[f"student::{p}"]))
        et = rel_err(probe_slice(sde[p]).float().cpu(), T(g[f"teacher::{p}"]))
        print(f"fused {kind} {p}: student {es:.2e} teacher {et:.2e}")
        assert es < 3e-2 and et < 3e-2
    assert rel_err(sde["resnet_bn1.running_mean"].float().cpu(), T(g["teacher_rm::resnet_bn1"])) < 1e-3


def test_cityscapes_shaped_step_vs_reference_trace(golden):
    """MixTrainer in its Cityscapes configuration (deep-stem ResNet-101, K=19, ProbOhemCrossEntropy2d thresh 0.7 with the min_kept-th
    smallest probability as threshold) against the iteration captured from the reference (train_trace_city.npz)."""
    import copy
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from css_amd.train_step import MixTrainer
    from oracle import css_oracle as O
    g = golden("train_trace_city")
    seed, gain, Kc = int(g["seed"]), float(g["residual_gain"]), 19
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none", "device_aug": "identity"}}
    m = Model_mix(resnet.resnet101(), num_classes=Kc, output_dim=256, config=cfg, temp=0.5)
    sd = O.init_state("stem", Kc, 256, seed, gain)
    m.model.load_state_dict(sd, strict=True)
    m.ema_model.load_state_dict(sd, strict=True)
    m = m.to(dev()).train()
    tr = MixTrainer(m, Kc, lr=6.4e-3, total_iter=100, min_lr=1e-4, num_queries=64, num_negatives=128, strong_threshold=0.8,
                    weak_threshold=0.0, un_threshold=0.97, sup="ohem", ohem_min_kept=3000)
    l_img, l_lab, u_img = T(g["0::l_img"]), T(g["0::l_lab"]).long(), T(g["0::u_img"])
    args = dict(lr=6.4e-3, temp_model=0.5, strong_threshold=0.8, weak_threshold=0.0, un_threshold=0.97, num_queries=64, num_negatives=128,
                sup="ohem", ohem_min_kept=3000)
    rec = {}
    O.train_step_mix(O.MixState("stem", Kc, 256, seed, gain), l_img, l_lab, u_img, record=rec, **args)
    anchors, negs, j = [], [], 0
    for hn in rec["hard_num"]:
        if hn > 0:
            anchors.append(g[f"0::anchor{j}"].astype(np.int64))
            negs.append(g[f"0::negative{j}"].astype(np.int64))
            j += 1
        else:
            anchors.append(None)
            negs.append(None)
    assert j == int(g["0::n_anchor"])
    r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()), ramp=1.0, _injected=dict(anchor=anchors, negative=negs))
    fails = []
    for key, gk, tol in (("sup", "sup", 2e-3), ("unsup", "unsup", 3e-2), ("contrast", "con", 2e-3)):
        ref = float(g[f"0::{gk}"])
        print(f"city {key}: hip {r[key].item():.6f} reference {ref:.6f}")
        if not abs(r[key].item() - ref) < tol * max(1.0, abs(ref)):
            fails.append((key, r[key].item(), ref))
    assert not fails, fails
    assert (r["pseudo"].cpu() != T(g["0::ulab"]).long()).float().mean().item() < 1e-3
    assert rel_err(tr.prototypes.cpu(), T(g["0::protos"])) < 2e-3
    sdm, sde = m.model.state_dict(), m.ema_model.state_dict()
    for p in ["resnet_conv1.0.weight", "resnet_layer3.10.conv2.weight", "classifier.3.weight", "representation.3.bias"]:
        es = rel_err(probe_slice(sdm[p]).float().cpu(), T(g[f"0::student::{p}"]))
        et = rel_err(probe_slice(sde[p]).float().cpu(), T(g[f"0::teacher::{p}"]))
        print(f"city {p}: student {es:.2e} teacher {et:.2e}")
        assert es < 3e-2 and et < 3e-2
