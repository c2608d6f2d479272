#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the fixtures it
writes (*.npz: inputs + expected outputs, no reference source) are committed
and travel to the GPU box.  Shims, none of which edits the
reference: stub ``torchvision`` modules, ``Tensor.cuda`` -> identity, a
1-process gloo group for ``concat_all_gather``.

    python tests/golden/make_golden.py            # regenerate everything
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# ---- shims ---------------------------------------------------------------
for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional",
             "torchvision.models"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self

import torch.distributed as dist  # noqa: E402

if not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("gloo", rank=0, world_size=1)

from generalframeworks.networks import resnet as ref_resnet  # noqa: E402
from generalframeworks.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep  # noqa: E402
from generalframeworks.networks import ddp_model as ref_ddp  # noqa: E402
from generalframeworks.loss import loss as ref_loss  # noqa: E402
from generalframeworks import utils as ref_utils  # noqa: E402
from generalframeworks.scheduler.my_lr_scheduler import PolyLR  # noqa: E402
from generalframeworks.scheduler.rampscheduler import RampdownScheduler  # noqa: E402

from oracle import css_oracle as O  # noqa: E402  (only for init_state: the seeded weights)


class TVResNet101(nn.Module):
    """torchvision-0.8.2-shaped ResNet-101 assembled from the reference's own
    Bottleneck / conv1x1 (torchvision itself is not installed here)."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 23, 2)
        self.layer4 = self._make(512, 3, 2)

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(ref_resnet.conv1x1(self.inplanes, planes * 4, stride), nn.BatchNorm2d(planes * 4))
        layers = [ref_resnet.Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(ref_resnet.Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)


def build_ref_net(backbone, K, seed, residual_gain=1.0):
    bb = TVResNet101() if backbone == "tv" else ref_resnet.resnet101(pretrained=False)
    net = DeepLabv3Plus_with_rep(bb, dilate_scale=8, num_classes=K, output_dim=256)
    sd = O.init_state(backbone, K, 256, seed, residual_gain)
    net.load_state_dict(sd, strict=True)
    return net, sd


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e3:.0f} KB")


PROBES = {
    "tv": ["resnet_conv1.weight", "resnet_layer1.0.conv2.weight", "resnet_layer2.0.downsample.0.weight",
           "resnet_layer3.5.bn2.weight", "resnet_layer4.2.conv2.weight", "ASPP.convs.2.0.weight",
           "ASPP.convs.4.1.weight", "project.0.weight", "classifier.3.bias", "representation.0.weight",
           "resnet_layer3.22.bn3.bias"],
    "stem": ["resnet_conv1.0.weight", "resnet_conv1.6.weight", "resnet_layer1.0.conv2.weight",
             "resnet_layer3.5.bn2.weight", "resnet_layer4.2.conv2.weight", "ASPP.convs.3.0.weight",
             "representation.3.weight"],
}


def probe_slice(t):
    return t.detach().flatten()[:: max(1, t.numel() // 2048)][:2048].clone()


def gen_network(backbone, size, K, seed, tag, residual_gain=1.0, batch=2):
    torch.manual_seed(seed)
    net, sd = build_ref_net(backbone, K, seed, residual_gain)
    x = torch.randn(batch, 3, size, size)
    net.train()
    pred, rep = net(x)
    wp = torch.randn_like(pred)
    wr = torch.randn_like(rep)
    loss = (pred * wp).sum() + (rep * wr).sum()
    loss.backward()
    named = dict(net.named_parameters())
    out = dict(x=x, pred=pred, rep=rep, wp=wp, wr=wr, seed=seed, K=K, residual_gain=residual_gain)
    for p in PROBES[backbone]:
        out["grad::" + p] = probe_slice(named[p].grad)
    bufs = dict(net.named_buffers())
    out["rm::resnet_bn1"] = bufs["resnet_bn1.running_mean"]
    out["rv::resnet_bn1"] = bufs["resnet_bn1.running_var"]
    out["rm::ASPP.convs.4.2"] = bufs["ASPP.convs.4.2.running_mean"]
    out["rv::ASPP.convs.4.2"] = bufs["ASPP.convs.4.2.running_var"]
    # eval mode with the (now once-updated) running stats
    net.eval()
    with torch.no_grad():
        pe, re_ = net(x)
    out["pred_eval"] = pe
    out["rep_eval_sub"] = re_[:, ::16]
    save(tag, **out)


def gen_pseudo(seed=5):
    g = torch.Generator().manual_seed(seed)
    K, C, h, H = 21, 256, 17, 65
    cases = {}
    for name, zero_proto in (("rand", False), ("zero", True)):
        pred_u = torch.randn(2, K, h, h, generator=g) * 3
        rep_u = torch.randn(2, C, h, h, generator=g)
        protos = torch.zeros(K, C) if zero_proto else torch.randn(K, C, generator=g)
        if not zero_proto:
            protos[3] = 0  # an unseen class row
        m = types.SimpleNamespace(num_classes=K, temp=0.5)
        # ddp_model.py:104-118, executed through the reference's own ops
        rep_b, rep_dim, rep_w, rep_h = rep_u.shape
        nr = F.normalize(rep_u.permute(0, 2, 3, 1), dim=-1).reshape(rep_b * rep_w * rep_h, rep_dim)
        npt = F.normalize(protos, dim=-1).permute(1, 0)
        sim = torch.mm(nr, npt).reshape(rep_b, rep_w, rep_h, K).permute(0, 3, 1, 2)
        siml = F.interpolate(sim, size=(H, H), mode="bilinear", align_corners=True)
        lg_rep, lb_rep = torch.max(F.softmax(siml / m.temp, dim=1), dim=1)
        pl = F.interpolate(pred_u, size=(H, H), mode="bilinear", align_corners=True)
        lg_cls, lb_cls = torch.max(torch.softmax(pl, dim=1), dim=1)
        lm = (~lb_cls.eq(lb_rep)).float()
        pseudo = lb_cls - lm * K
        pseudo[pseudo < 0] = 255
        prob_all = F.softmax(sim / m.temp, dim=1)
        cases.update({f"{name}::pred_u": pred_u, f"{name}::rep_u": rep_u, f"{name}::protos": protos,
                      f"{name}::sim": sim, f"{name}::lg_rep": lg_rep, f"{name}::lb_rep": lb_rep,
                      f"{name}::lg_cls": lg_cls, f"{name}::lb_cls": lb_cls, f"{name}::pseudo": pseudo,
                      f"{name}::prob_all": prob_all})
    save("pseudo_labels", **cases)


class Recorder:
    """Wraps the three RNG call sites of Contrast_Loss (loss.py:127,137,414)."""

    def __init__(self):
        self.anchor, self.negative = [], []
        self._randint = torch.randint
        self._sampler = ref_loss.negative_index_sampler

    def __enter__(self):
        def randint(*a, **k):
            r = self._randint(*a, **k)
            self.anchor.append(r.tolist())
            return r

        def sampler(samp_num, seg):
            r = self._sampler(samp_num, seg)
            self.negative.append(list(r))
            return r

        torch.randint = randint
        ref_loss.negative_index_sampler = sampler
        return self

    def __exit__(self, *a):
        torch.randint = self._randint
        ref_loss.negative_index_sampler = self._sampler


def make_loss_inputs(g, B2, K, C, h, present, ignore_frac=0.1, hard_free=()):
    """Piecewise-constant label map over ``present`` classes, ~10 % masked."""
    lab = torch.zeros(B2, h, h, dtype=torch.long)
    blk = 4
    nb = (h + blk - 1) // blk
    cls = torch.tensor(present)[torch.randint(len(present), (B2, nb, nb), generator=g)]
    lab = cls.repeat_interleave(blk, 1).repeat_interleave(blk, 2)[:, :h, :h]
    label = F.one_hot(lab, K).permute(0, 3, 1, 2).float()
    mask = (torch.rand(B2, 1, h, h, generator=g) > ignore_frac).float()
    rep = torch.randn(B2, C, h, h, generator=g)
    prob = torch.softmax(torch.randn(B2, K, h, h, generator=g) * 2, dim=1)
    for c in hard_free:   # class with valid pixels but no hard ones
        prob[:, c] = 0.999
    return rep, label, mask, prob


def gen_contrast(seed=11):
    K, C = 21, 256
    cases = [
        # name, B2, h, present, Q, N, proto_init, hard_free
        ("first", 4, 17, [0, 2, 5, 7, 20], 64, 128, "zero", ()),
        ("ema", 4, 17, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], 128, 256, "rand", ()),
        ("nohard", 4, 17, [1, 4, 9], 32, 64, "mixed", (4,)),
        ("single", 2, 9, [6], 16, 32, "zero", ()),
        ("stress", 2, 9, [0, 3, 8, 15], 1024, 2048, "rand", ()),
    ]
    out = {}
    for name, B2, h, present, Q, N, pinit, hard_free in cases:
        g = torch.Generator().manual_seed(seed + len(name))
        torch.manual_seed(seed)
        np.random.seed(seed)
        rep, label, mask, prob = make_loss_inputs(g, B2, K, C, h, present, hard_free=hard_free)
        if pinit == "zero":
            protos = torch.zeros(K, C)
        elif pinit == "rand":
            protos = torch.randn(K, C, generator=g)
        else:
            protos = torch.randn(K, C, generator=g)
            protos[present[0]] = 0
        protos_in = protos.clone()
        rep.requires_grad_(True)
        crit = ref_loss.Contrast_Loss(num_queries=Q, num_negatives=N, temp=0.5, strong_threshold=0.8, alpha=0.99)
        with Recorder() as rec:
            loss = crit(rep, label, mask, prob, protos)
        loss.backward()
        out.update({f"{name}::rep": rep.detach(), f"{name}::label_idx": label.argmax(1).to(torch.int16),
                    f"{name}::mask": mask.to(torch.uint8), f"{name}::prob": prob.to(torch.float32),
                    f"{name}::protos_in": protos_in, f"{name}::protos_out": protos,
                    f"{name}::loss": loss.detach(), f"{name}::QN": np.array([Q, N]),
                    f"{name}::n_anchor": len(rec.anchor)})
        # grads are sparse: store the non-zero pixel rows only
        gr = rep.grad.permute(0, 2, 3, 1).reshape(-1, C)
        nz = (gr.abs().sum(1) > 0).nonzero().flatten()
        out[f"{name}::grad_rows"] = nz.to(torch.int32)
        out[f"{name}::grad_vals"] = gr[nz]
        if name == "stress":
            # indices not stored (32 MB): the oracle replays the same three RNG streams from
            # torch.manual_seed(seed) / np.random.seed(seed) and must reproduce the loss.
            out[f"{name}::rng_seed"] = seed
            continue
        for i, (a, n) in enumerate(zip(rec.anchor, rec.negative)):
            out[f"{name}::anchor{i}"] = np.asarray(a, dtype=np.int16)
            out[f"{name}::negative{i}"] = np.asarray(n, dtype=np.int16)
    out["rng_seed"] = seed
    save("contrast_loss", **out)


def gen_losses(seed=21):
    g = torch.Generator().manual_seed(seed)
    K, H = 21, 33
    out = {}
    # Attention_Threshold_Loss incl. the all-ignored NaN case
    for name in ("normal", "allignored"):
        pred = (torch.randn(3, K, H, H, generator=g) * 2).requires_grad_(True)
        lab = torch.randint(0, K, (3, H, H), generator=g)
        lab[torch.rand(3, H, H, generator=g) < 0.2] = -1
        if name == "allignored":
            lab[:] = -1
        logits = torch.rand(3, H, H, generator=g)
        loss = ref_loss.Attention_Threshold_Loss(0.7)(pred, lab, logits)
        loss.backward()
        out.update({f"att_{name}::pred": pred.detach(), f"att_{name}::lab": lab.to(torch.int16),
                    f"att_{name}::logits": logits, f"att_{name}::loss": loss.detach(),
                    f"att_{name}::grad": pred.grad})
    # plain CE (nn.CrossEntropyLoss(ignore_index=-1))
    pred = (torch.randn(2, K, H, H, generator=g) * 2).requires_grad_(True)
    lab = torch.randint(0, K, (2, H, H), generator=g)
    lab[torch.rand(2, H, H, generator=g) < 0.1] = -1
    loss = nn.CrossEntropyLoss(ignore_index=-1)(pred, lab)
    loss.backward()
    out.update({"ce::pred": pred.detach(), "ce::lab": lab.to(torch.int16), "ce::loss": loss.detach(), "ce::grad": pred.grad})
    # OHEM: min_kept <= #valid with threshold raised / not raised, and min_kept > #valid
    for name, min_kept, scale in (("raise", 1500, 6.0), ("keep", 200, 0.5), ("toofew", 5000, 2.0)):
        pred = (torch.randn(2, K, H, H, generator=g) * scale).requires_grad_(True)
        lab = torch.randint(0, K, (2, H, H), generator=g)
        lab[torch.rand(2, H, H, generator=g) < 0.1] = -1
        crit = ref_loss.ProbOhemCrossEntropy2d(ignore_label=-1, thresh=0.7, min_kept=min_kept)
        loss = crit(pred, lab.clone())
        loss.backward()
        out.update({f"ohem_{name}::pred": pred.detach(), f"ohem_{name}::lab": lab.to(torch.int16),
                    f"ohem_{name}::min_kept": min_kept, f"ohem_{name}::loss": loss.detach(),
                    f"ohem_{name}::grad": pred.grad})
    save("losses", **out)


def gen_labelmask(seed=31):
    g = torch.Generator().manual_seed(seed)
    K = 21
    out = {}
    for H, h in ((65, 17), (129, 33), (97, 25)):
        l_lab = torch.randint(-1, K, (2, H, H), generator=g)
        u_lab = torch.randint(-1, K, (2, H, H), generator=g)
        u_logits = torch.rand(2, H, H, generator=g)
        u_mask = u_logits.ge(0.7).float()
        mask_all = torch.cat(((l_lab.unsqueeze(1) >= 0).float(), u_mask.unsqueeze(1)))
        mask_all = F.interpolate(mask_all, size=(h, h), mode="nearest")
        label_l = F.interpolate(ref_utils.label_onehot(l_lab, K), size=(h, h), mode="nearest")
        label_u = F.interpolate(ref_utils.label_onehot_2(u_lab, K), size=(h, h), mode="nearest")[:, 1:]
        label_all = torch.cat((label_l, label_u))
        out.update({f"{H}::l_lab": l_lab.to(torch.int16), f"{H}::u_lab": u_lab.to(torch.int16),
                    f"{H}::u_logits": u_logits, f"{H}::mask_all": mask_all.to(torch.uint8),
                    f"{H}::label_all": label_all.to(torch.uint8)})
    save("label_mask", **out)


def gen_eval(seed=51):
    """Eval path of mix_label.py:199-225: bilinear(align_corners=True) to the label size -> argmax -> ConfMatrix.update
    (util/meter.py:39-48) over two batches -> mean_intersection_over_union (util/miou.py:3-9)."""
    from generalframeworks.util.meter import ConfMatrix
    from generalframeworks.util.miou import mean_intersection_over_union
    g = torch.Generator().manual_seed(seed)
    out = {}
    for tag, (K, h, H) in {"voc": (21, 17, 65), "city": (19, 25, 97)}.items():
        meter = ConfMatrix(num_classes=K, fmt=":6.4f", name="test_miou")
        for bi in range(2):
            pred = torch.randn(2, K, h, h, generator=g)
            # make the prediction correlate with the label so that the matrix has a real diagonal
            lab = torch.randint(0, K, (2, H, H), generator=g)
            lab_small = F.interpolate(lab[:, None].float(), size=(h, h), mode="nearest")[:, 0].long()
            pred = pred + 2.0 * F.one_hot(lab_small, K).permute(0, 3, 1, 2).float() * (torch.rand(2, 1, h, h, generator=g) > 0.4)
            lab[torch.rand(2, H, H, generator=g) < 0.07] = -1          # ignore label after the 255 -> -1 mapping
            if bi == 1:
                lab[0, :5] = 255                                       # and a raw 255 band: >= K is dropped as well
            up = F.interpolate(pred, size=lab.shape[1:], mode="bilinear", align_corners=True)
            meter.update(up.argmax(1).flatten(), lab.flatten())
            out.update({f"{tag}::pred{bi}": pred, f"{tag}::lab{bi}": lab.to(torch.int16), f"{tag}::argmax{bi}": up.argmax(1).to(torch.uint8)})
        out[f"{tag}::mat"] = meter.mat
        out[f"{tag}::miou"] = np.float64(mean_intersection_over_union(meter.mat))
    save("eval", **out)


def gen_schedules():
    opt = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=6.4e-3)
    sch = PolyLR(opt, 1000, min_lr=1e-4)
    lrs = []
    for _ in range(1000):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    ramp = RampdownScheduler(0, 200, 0, 1.0, 0, -5.0)
    rv = []
    for _ in range(210):
        rv.append(ramp.value)
        ramp.step()
    # EMA decay table through the reference's Model_mix.ema_update on a 1-param stand-in
    class _M:  # noqa: D401 - tiny stand-in exposing what ema_update touches
        pass
    m = _M()
    m.step, m.alpha = 0, 0.99
    m.model = nn.Linear(1, 1, bias=False)
    m.ema_model = nn.Linear(1, 1, bias=False)
    with torch.no_grad():
        m.model.weight.fill_(1.0)
        m.ema_model.weight.fill_(0.0)
    ema = []
    for _ in range(150):
        ref_ddp.Model_mix.ema_update(m)
        ema.append(float(m.ema_model.weight))
    # SGD nesterov trajectory as configured at mix_label.py:96-97
    p = nn.Parameter(torch.tensor([1.0, -2.0, 0.5]))
    o = torch.optim.SGD([p], lr=0.01, weight_decay=5e-4, momentum=0.9, nesterov=True)
    traj = []
    gg = torch.Generator().manual_seed(1)
    grads = torch.randn(5, 3, generator=gg)
    for i in range(5):
        p.grad = grads[i].clone()
        o.step()
        traj.append(p.detach().clone())
    save("schedules", poly=np.array(lrs), ramp=np.array(rv), ema=np.array(ema), sgd_grads=grads,
         sgd_traj=torch.stack(traj))


def gen_train_trace(seed=41, gain=1.0, tag="train_trace", backbone="tv", K=21, ohem_min_kept=None, iters=2):
    """Two iterations of the mix_label.train body (mix_label.py:162-196) driven through the
    reference's Model_mix / Contrast_Loss / Attention_Threshold_Loss with identity augmentation
    (batch_transform_2 patched to label 255 -> -1, mix_mode 'none')."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    S, B = 65, 2
    bb = TVResNet101() if backbone == "tv" else ref_resnet.resnet101(pretrained=False)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_ddp.Model_mix(bb, num_classes=K, output_dim=256, config={"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none"}}, temp=0.5)
    sd = O.init_state(backbone, K, 256, seed, gain)
    model.model.load_state_dict(sd, strict=True)
    model.ema_model.load_state_dict(sd, strict=True)

    def ident(images, labels, l1=None, l2=None, crop_size=None, scale_size=None, augmentation=True):
        lab = labels.long().clone()
        lab[lab == 255] = -1
        return images, lab, l1, l2

    ref_ddp.batch_transform_2 = ident
    model.train()
    crit_c = ref_loss.Contrast_Loss(strong_threshold=0.8, num_queries=64, num_negatives=128, temp=0.5, alpha=0.99)
    crit_u = ref_loss.Attention_Threshold_Loss(0.97)
    crit_s = nn.CrossEntropyLoss(ignore_index=-1)
    if ohem_min_kept is not None:      # the Cityscapes criterion of mix_label.py:82 (thresh 0.7)
        crit_s = ref_loss.ProbOhemCrossEntropy2d(ignore_label=-1, thresh=0.7, min_kept=ohem_min_kept)
    opt = torch.optim.SGD(model.model.parameters(), lr=6.4e-3, weight_decay=5e-4, momentum=0.9, nesterov=True)
    sch = PolyLR(opt, 100, min_lr=1e-4)
    protos = torch.zeros(K, 256)
    g = torch.Generator().manual_seed(seed)
    out = dict(seed=seed, residual_gain=gain)
    probes = ["resnet_conv1.weight" if backbone == "tv" else "resnet_conv1.0.weight", "resnet_layer3.10.conv2.weight", "classifier.3.weight",
              "representation.3.bias"]
    for it in range(iters):
        l_img = torch.randn(B, 3, S, S, generator=g)
        u_img = torch.randn(B, 3, S, S, generator=g)
        blk = torch.randint(0, K, (B, 5, 5), generator=g)
        l_lab = blk.repeat_interleave(13, 1).repeat_interleave(13, 2)[:, :S, :S].clone()
        l_lab[torch.rand(B, S, S, generator=g) < 0.05] = -1
        with Recorder() as rec:
            pl, pu, ulab, ulc, ulr, rep_all, prob_all = model(l_img, u_img, protos)
            sup = crit_s(pl, l_lab)
            unsup = crit_u(pu, ulab, ulc)
            with torch.no_grad():
                umask = ulc.ge(0.0).float()      # weak_threshold 0 so that pseudo-labelled pixels take part
                mask_all = torch.cat(((l_lab.unsqueeze(1) >= 0).float(), umask.unsqueeze(1)))
                mask_all = F.interpolate(mask_all, size=prob_all.shape[2:], mode="nearest")
                label_l = F.interpolate(ref_utils.label_onehot(l_lab, K), size=prob_all.shape[2:], mode="nearest")
                label_u = F.interpolate(ref_utils.label_onehot_2(ulab, K), size=prob_all.shape[2:], mode="nearest")[:, 1:]
                label_all = torch.cat((label_l, label_u))
            con = crit_c(rep_all, label_all, mask_all, prob_all, protos)
        total = sup + unsup + con * 1.0
        opt.zero_grad()
        total.backward()
        opt.step()
        model.ema_update()
        sch.step()
        out.update({f"{it}::l_img": l_img, f"{it}::u_img": u_img, f"{it}::l_lab": l_lab.to(torch.int16),
                    f"{it}::sup": sup.detach(), f"{it}::unsup": unsup.detach(), f"{it}::con": con.detach(),
                    f"{it}::ulab": ulab.to(torch.int16), f"{it}::protos": protos.clone(),
                    f"{it}::n_anchor": len(rec.anchor)})
        for i, (a, n) in enumerate(zip(rec.anchor, rec.negative)):
            out[f"{it}::anchor{i}"] = np.asarray(a, dtype=np.int32)
            out[f"{it}::negative{i}"] = np.asarray(n, dtype=np.int32)
        sdm, sde = model.model.state_dict(), model.ema_model.state_dict()
        for p in probes:
            out[f"{it}::student::{p}"] = probe_slice(sdm[p])
            out[f"{it}::teacher::{p}"] = probe_slice(sde[p])
        out[f"{it}::teacher_rm::resnet_bn1"] = sde["resnet_bn1.running_mean"].clone()
    save(tag, **out)


def gen_train_trace_w5(kind, seed, gain=0.25):
    """ONE iteration of the train body of cross_label.py:162-198 (kind 'cross', warm-up branch: class-predictor pseudo labels) or
    ori_pseudo.py:158-187 (kind 'ori') driven through the reference's Model_cross / Model_ori_pseudo, Contrast_Loss,
    Attention_Threshold_Loss, label_onehot, SGD(nesterov) and PolyLR, with identity augmentation (the batch_transform* used by the
    model patched to the label convention 255 -> -1, mix_mode 'none')."""
    import io
    import contextlib
    torch.manual_seed(seed)
    np.random.seed(seed)
    K, S, B, weak = 21, 65, 2, 0.0
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none"}, "Loss": {"weak_threshold": weak}}
    with contextlib.redirect_stdout(io.StringIO()):
        if kind == "cross":
            model = ref_ddp.Model_cross(TVResNet101(), num_classes=K, output_dim=256, config=cfg, temp=0.5)
        else:
            model = ref_ddp.Model_ori_pseudo(TVResNet101(), num_classes=K, output_dim=256, config=cfg)
    sd = O.init_state("tv", K, 256, seed, gain)
    model.model.load_state_dict(sd, strict=True)
    model.ema_model.load_state_dict(sd, strict=True)

    def fix(lab):
        lab = lab.long().clone()
        lab[lab == 255] = -1
        return lab

    def ident1(images, labels, logits=None, crop_size=None, scale_size=None, augmentation=True):
        return images, fix(labels), logits

    def ident3(images, l1, l2, g1=None, g2=None, crop_size=None, scale_size=None, augmentation=True):
        return images, fix(l1), fix(l2), g1, g2

    ref_ddp.batch_transform = ident1
    ref_ddp.batch_transform_3 = ident3
    # the reference's generate_cut_gather / _3 have no 'none' mode (VOC.py:390,468 raise): no mixing = identity
    ref_ddp.generate_cut_gather = lambda image, label, logits, mode=None: (image, label, logits)
    ref_ddp.generate_cut_gather_3 = lambda image, l1, l2, g1, g2, mode=None: (image, l1, l2, g1, g2)
    model.train()
    crit_c = ref_loss.Contrast_Loss(strong_threshold=0.8, num_queries=64, num_negatives=128, temp=0.5, alpha=0.99)
    crit_u = ref_loss.Attention_Threshold_Loss(0.97)
    crit_s = nn.CrossEntropyLoss(ignore_index=-1)
    opt = torch.optim.SGD(model.model.parameters(), lr=6.4e-3, weight_decay=5e-4, momentum=0.9, nesterov=True)
    sch = PolyLR(opt, 100, min_lr=1e-4)
    protos = torch.zeros(K, 256)
    g = torch.Generator().manual_seed(seed)
    l_img = torch.randn(B, 3, S, S, generator=g)
    u_img = torch.randn(B, 3, S, S, generator=g)
    blk = torch.randint(0, K, (B, 5, 5), generator=g)
    l_lab = blk.repeat_interleave(13, 1).repeat_interleave(13, 2)[:, :S, :S].clone()
    l_lab[torch.rand(B, S, S, generator=g) < 0.05] = -1
    out = dict(seed=seed, residual_gain=gain, l_img=l_img, u_img=u_img, l_lab=l_lab.to(torch.int16), weak=weak)
    with Recorder() as rec:
        if kind == "cross":
            pl, pu, ulab_cls, ulab_rep, ulc, ulr, rep_all, pred_all = model(l_img, u_img, protos)
            ulab = ulab_cls
            out["ulab_rep"] = ulab_rep.to(torch.int16)
        else:
            pl, pu, ulab, ulc, rep_all, pred_all, pu_raw = model(l_img, u_img)
        sup = crit_s(pl, l_lab)
        unsup = crit_u(pu, ulab, ulc)                       # cross_label.py:174-175 (epoch < warmup) / ori_pseudo.py:167
        with torch.no_grad():
            umask = ulc.ge(weak).float()
            mask_all = torch.cat(((l_lab.unsqueeze(1) >= 0).float(), umask.unsqueeze(1)))
            mask_all = F.interpolate(mask_all, size=pred_all.shape[2:], mode="nearest")
            label_l = F.interpolate(ref_utils.label_onehot(l_lab, K), size=pred_all.shape[2:], mode="nearest")
            label_u = F.interpolate(ref_utils.label_onehot(ulab, K), size=pred_all.shape[2:], mode="nearest")
            label_all = torch.cat((label_l, label_u))
            prob_all = pred_all if kind == "cross" else torch.softmax(pred_all, dim=1)      # ori_pseudo.py:178
        con = crit_c(rep_all, label_all, mask_all, prob_all, protos)
    total = sup + unsup + con
    opt.zero_grad()
    total.backward()
    opt.step()
    model.ema_update()
    sch.step()
    out.update(sup=sup.detach(), unsup=unsup.detach(), con=con.detach(), ulab=ulab.to(torch.int16), ulc=ulc, protos=protos.clone(),
               n_anchor=len(rec.anchor), lr_next=opt.param_groups[0]["lr"])
    for i, (a, n) in enumerate(zip(rec.anchor, rec.negative)):
        out[f"anchor{i}"] = np.asarray(a, dtype=np.int32)
        out[f"negative{i}"] = np.asarray(n, dtype=np.int32)
    sdm, sde = model.model.state_dict(), model.ema_model.state_dict()
    for p in ["resnet_conv1.weight", "resnet_layer3.10.conv2.weight", "classifier.3.weight", "representation.3.bias"]:
        out[f"student::{p}"] = probe_slice(sdm[p])
        out[f"teacher::{p}"] = probe_slice(sde[p])
    out["teacher_rm::resnet_bn1"] = sde["resnet_bn1.running_mean"].clone()
    save(f"train_trace_{kind}", **out)


def gen_dataset(seed=61):
    """the torchvision-free functions of the reference's dataset helpers (split readers, Cityscapes path and
    id mapping) run on a scratch tree; recorded as JSON (strings + small integer tables)."""
    import json
    import tempfile
    from generalframeworks.dataset_helpers import VOC as ref_voc
    from generalframeworks.dataset_helpers import Cityscapes as ref_city
    rng = np.random.RandomState(seed)
    ids_u8 = np.arange(256, dtype=np.uint8)
    odd = np.array([-1, 0, 7, 33, 34, 255, 256, 300, -7], dtype=np.int16)
    tile = rng.randint(0, 34, size=(6, 9)).astype(np.uint8)
    names = ["aachen_000000_000019_leftImg8bit", "frankfurt_000001_083852_leftImg8bit ", "x_1_leftImg8bit", "nounderscore"]
    paths = []
    for nme in names:
        for mode in ("train", "val"):
            img, city = ref_city.image_root_transform(nme, mode=mode)
            paths.append(dict(name=nme, mode=mode, image=img, city=city, label=ref_city.label_root_transform(nme, city, mode=mode)))
    with tempfile.TemporaryDirectory() as d:
        files = {"labeled_filename.txt": "2007_000032\n2007_000039\n\n2007_000063", "unlabeled_filename.txt": "a\nb\r\nc\n",
                 "valid_filename.txt": ""}
        os.makedirs(f"{d}/662/3407")
        for k, v in files.items():
            with open(f"{d}/662/3407/{k}", "w", newline="") as f:
                f.write(v)
        split_voc = [list(x) for x in ref_voc.get_pascal_idx_via_txt(d, 662, 3407)]
        split_city = [list(x) for x in ref_city.get_cityscapes_idx_via_txt(d, "662", "3407")]
        bd = ref_voc.VOC_BuildData(data_path="/data/voc", txt_path=d, label_num=662, seed=3407, crop_size=[321, 321])
        sets = [dict(root=s.root, n=len(s), crop=list(s.crop_size), scale=list(s.scale_size), aug=s.augmentation, train=s.train)
                for s in bd.build()]
        cd = ref_city.City_BuildData(data_path="~/city", txt_path=d, label_num=662, seed=3407, crop_size=[769, 769])
        csets = [dict(root=s.root, n=len(s), crop=list(s.crop_size), scale=list(s.scale_size), aug=s.augmentation, train=s.train)
                 for s in cd.build()]
        battr = dict(image_size=bd.image_size, num_segments=bd.num_segments, scale_size=list(bd.scale_size))
        cattr = dict(im_size=cd.im_size, num_segments=cd.num_segments, scale_size=list(cd.scale_size))
    out = dict(class_map_u8=ref_city.cityscapes_class_map(ids_u8).tolist(), odd_in=odd.tolist(),
               class_map_odd=ref_city.cityscapes_class_map(odd).tolist(), tile_in=tile.tolist(),
               class_map_tile=ref_city.cityscapes_class_map(tile).tolist(), paths=paths, split_files=files, split_voc=split_voc,
               split_city=split_city, voc_sets=sets, city_sets=csets, voc_attrs=battr, city_attrs=cattr,
               home=os.path.expanduser("~"))
    with open(os.path.join(HERE, "dataset_helpers.json"), "w") as f:
        json.dump(out, f, separators=(',', ':'))
    print("dataset_helpers.json written")


if __name__ == "__main__":
    which = sys.argv[1:] or ["net", "pseudo", "contrast", "losses", "labelmask", "sched", "trace", "eval", "dataset", "trace_w5", "trace_city"]
    if "eval" in which:
        gen_eval()
    if "dataset" in which:
        gen_dataset()
    if "trace_city" in which:
        # Cityscapes-shaped step: deep-stem ResNet-101, K=19, OHEM with a min_kept that makes the k-th smallest probability the threshold
        gen_train_trace(45, 0.25, "train_trace_city", backbone="stem", K=19, ohem_min_kept=3000, iters=1)
    if "trace_w5" in which:
        gen_train_trace_w5("cross", 47)
        gen_train_trace_w5("ori", 49)
    if "net" in which:
        gen_network("tv", 65, 21, 101, "net_tv_65")
        gen_network("stem", 65, 19, 102, "net_stem_65")
        gen_network("tv", 97, 21, 103, "net_tv_97")
        gen_network("tv", 65, 21, 104, "net_tv_65_damped", residual_gain=0.25, batch=3)
        gen_network("stem", 65, 19, 105, "net_stem_65_damped", residual_gain=0.25, batch=3)
    if "pseudo" in which:
        gen_pseudo()
    if "contrast" in which:
        gen_contrast()
    if "losses" in which:
        gen_losses()
    if "labelmask" in which:
        gen_labelmask()
    if "sched" in which:
        gen_schedules()
    if "trace" in which:
        gen_train_trace()
        gen_train_trace(43, 0.25, "train_trace_damped")
