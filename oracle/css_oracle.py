"""CPU oracle for the CSS hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU (fp32) *restatement* of the algorithms on the
reference's data-parallel hot path.  It is the checker the
HIP product is compared against.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; nothing under
``css_amd/`` does, and the product fails loudly when the HIP library is
missing rather than falling back to this code.

Parity pin: the reference ships no tests / golden vectors, so
the oracle is pinned against outputs of the reference itself, produced in the
build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference`` with three shims and stores inputs + expected outputs as
small ``.npz`` fixtures).  ``tests/test_oracle_golden.py`` checks every
function below against those fixtures.

Third-party arithmetic not in the reference tree:
torchvision==0.8.2 ``models.resnet101`` (the default backbone).  Its published
structure (7x7/s2 stem conv, BN, ReLU, MaxPool(3,2,1), Bottleneck x [3,4,23,3]
with the stride on conv2) is restated here as backbone ``"tv"``; the in-tree
deep-stem ResNet (``networks/resnet.py:142-291``) is backbone ``"stem"``.

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).  All tensors are NCHW fp32 like the reference's.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Architecture description
# --------------------------------------------------------------------------

BN_EPS = 1e-5       # nn.BatchNorm2d default, used everywhere in the reference
BN_MOMENTUM = 0.1


def _conv(name, cin, cout, k, stride=1, pad=0, dil=1, bias=False):
    return dict(kind="conv", name=name, cin=cin, cout=cout, k=k, stride=stride,
                pad=pad, dil=dil, bias=bias)


def _bn(name, c):
    return dict(kind="bn", name=name, c=c)


def backbone_spec(backbone: str) -> dict:
    """Layer table of the ResNet-101 trunk *after* ``_nostride_dilate`` with
    ``dilate_scale=8`` has rewritten it (generalframeworks/networks/deeplabv3/
    deeplabv3.py:93-96,135-149).

    ``tv``   : torchvision-0.8.2-shaped ResNet-101 (mix_label.py:68).
    ``stem`` : generalframeworks/networks/resnet.py:142-291 (deep stem,
               ceil_mode max-pool, inplanes 128); the multi-grid dilations it
               is built with are overwritten by ``_nostride_dilate`` so every
               3x3 in layer3 ends at d=2 and in layer4 at d=4.
    """
    spec = {}
    if backbone == "tv":
        spec["stem"] = [_conv("resnet_conv1", 3, 64, 7, 2, 3)]
        spec["stem_bn"] = _bn("resnet_bn1", 64)
        spec["maxpool_ceil"] = False
        inplanes = 64
    elif backbone == "stem":
        # resnet.py:177-185 conv3x3(3,64,s2) BN ReLU conv3x3(64,64) BN ReLU conv3x3(64,128)
        spec["stem"] = [
            _conv("resnet_conv1.0", 3, 64, 3, 2, 1), _bn("resnet_conv1.1", 64),
            _conv("resnet_conv1.3", 64, 64, 3, 1, 1), _bn("resnet_conv1.4", 64),
            _conv("resnet_conv1.6", 64, 128, 3, 1, 1),
        ]
        spec["stem_bn"] = _bn("resnet_bn1", 128)
        spec["maxpool_ceil"] = True      # resnet.py:188-190
        inplanes = 128
    else:
        raise ValueError(backbone)

    layers = []
    cfg = [(64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)]
    for li, (planes, nblocks, stride) in enumerate(cfg, start=1):
        blocks = []
        for bi in range(nblocks):
            pre = f"resnet_layer{li}.{bi}"
            s = stride if bi == 0 else 1
            # conv2 as built (stride s, dilation 1, pad 1), then _nostride_dilate
            c2_stride, c2_dil = s, 1
            ds_stride = s
            if li in (3, 4):
                dilate = 2 if li == 3 else 4
                if c2_stride == 2:                 # deeplabv3.py:139-143
                    c2_stride, c2_dil = 1, dilate // 2
                else:                              # deeplabv3.py:146-149
                    c2_dil = dilate
                ds_stride = 1
            if backbone == "stem" and li in (3, 4):
                # ResNet_Stem builds layer3/4 with stride 1 already
                # (replace_stride_with_dilation=[False,True,True], resnet.py:150,237-239),
                # so no conv has stride 2 and every 3x3 takes the "other" branch.
                c2_stride, c2_dil = 1, (2 if li == 3 else 4)
            blk = dict(
                conv1=_conv(pre + ".conv1", inplanes, planes, 1),
                bn1=_bn(pre + ".bn1", planes),
                conv2=_conv(pre + ".conv2", planes, planes, 3, c2_stride, c2_dil, c2_dil),
                bn2=_bn(pre + ".bn2", planes),
                conv3=_conv(pre + ".conv3", planes, planes * 4, 1),
                bn3=_bn(pre + ".bn3", planes * 4),
                downsample=None,
            )
            if bi == 0 and (stride != 1 or inplanes != planes * 4):
                blk["downsample"] = (
                    _conv(pre + ".downsample.0", inplanes, planes * 4, 1, ds_stride),
                    _bn(pre + ".downsample.1", planes * 4),
                )
            blocks.append(blk)
            inplanes = planes * 4
        layers.append(blocks)
    spec["layers"] = layers
    return spec


def head_spec(num_classes=21, output_dim=256) -> dict:
    """ASPP + decoder (deeplabv3.py:113-133, aspp.py:41-65)."""
    h = {}
    h["aspp0"] = (_conv("ASPP.convs.0.0", 2048, 256, 1), _bn("ASPP.convs.0.1", 256))
    h["aspp_d"] = [(_conv(f"ASPP.convs.{i}.0", 2048, 256, 3, 1, d, d), _bn(f"ASPP.convs.{i}.1", 256))
                   for i, d in zip((1, 2, 3), (12, 24, 36))]
    h["aspp_pool"] = (_conv("ASPP.convs.4.1", 2048, 256, 1), _bn("ASPP.convs.4.2", 256))
    h["aspp_proj"] = (_conv("ASPP.project.0", 1280, 256, 1), _bn("ASPP.project.1", 256))
    h["project"] = (_conv("project.0", 256, 48, 1), _bn("project.1", 48))
    h["classifier"] = (_conv("classifier.0", 304, 256, 3, 1, 1), _bn("classifier.1", 256),
                       _conv("classifier.3", 256, num_classes, 1, bias=True))
    h["representation"] = (_conv("representation.0", 304, 256, 3, 1, 1), _bn("representation.1", 256),
                           _conv("representation.3", 256, output_dim, 1, bias=True))
    return h


def _walk(spec_part):
    if isinstance(spec_part, dict) and "kind" in spec_part:
        yield spec_part
    elif isinstance(spec_part, dict):
        for v in spec_part.values():
            yield from _walk(v)
    elif isinstance(spec_part, (list, tuple)):
        for v in spec_part:
            yield from _walk(v)


def all_layers(backbone: str, num_classes=21, output_dim=256) -> List[dict]:
    """Every conv/bn entry in module-registration order of the reference
    (deeplabv3.py:103-133): trunk first, then ASPP, project, classifier, representation."""
    bs = backbone_spec(backbone)
    out = list(_walk(bs["stem"])) + [bs["stem_bn"]]
    for blocks in bs["layers"]:
        for blk in blocks:
            for key in ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3"):
                out.append(blk[key])
            if blk["downsample"] is not None:
                out.extend(blk["downsample"])
    hs = head_spec(num_classes, output_dim)
    for key in ("aspp0",):
        out.extend(hs[key])
    for pair in hs["aspp_d"]:
        out.extend(pair)
    for key in ("aspp_pool", "aspp_proj", "project", "classifier", "representation"):
        out.extend(hs[key])
    return out


def init_state(backbone: str, num_classes=21, output_dim=256, seed=0, residual_gain=1.0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic, non-degenerate state_dict with the reference's key names
    (SURVEY section 5 checkpoint row).  Conv: Kaiming-normal fan_out; BN gamma~U(.5,1.5),
    beta~N(0,.1), running_mean~N(0,.1), running_var~U(.5,1.5) (the reference's
    own init zeroes every bn3.weight, resnet.py:218-223, which would hide the
    residual branches from a parity check).  ``residual_gain`` scales every ``bn3.weight``: with 1.0 the
    random network amplifies a 1-ulp input perturbation ~1000x by the ASPP output (fp32 itself is then only
    good to ~1e-3 against fp64); 0.25 gives the well-conditioned regime of a trained network."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for L in all_layers(backbone, num_classes, output_dim):
        n = L["name"]
        if L["kind"] == "conv":
            fan_out = L["cout"] * L["k"] * L["k"]
            w = torch.randn(L["cout"], L["cin"], L["k"], L["k"], generator=g) * math.sqrt(2.0 / fan_out)
            sd[n + ".weight"] = w
            if L["bias"]:
                sd[n + ".bias"] = torch.randn(L["cout"], generator=g) * 0.1
        else:
            c = L["c"]
            sd[n + ".weight"] = torch.rand(c, generator=g) + 0.5
            if n.endswith(".bn3"):
                sd[n + ".weight"] *= residual_gain
            sd[n + ".bias"] = torch.randn(c, generator=g) * 0.1
            sd[n + ".running_mean"] = torch.randn(c, generator=g) * 0.1
            sd[n + ".running_var"] = torch.rand(c, generator=g) + 0.5
            sd[n + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return sd


def param_names(backbone: str, num_classes=21, output_dim=256) -> List[str]:
    """Names of the trainable parameters in ``module.parameters()`` order."""
    names = []
    for L in all_layers(backbone, num_classes, output_dim):
        names.append(L["name"] + ".weight")
        if L["kind"] == "bn" or L.get("bias"):
            names.append(L["name"] + ".bias")
    return names


# --------------------------------------------------------------------------
# Network forward (N2-N7)
# --------------------------------------------------------------------------

def _apply_conv(sd, L, x):
    return F.conv2d(x, sd[L["name"] + ".weight"], sd.get(L["name"] + ".bias") if L["bias"] else None,
                    stride=L["stride"], padding=L["pad"], dilation=L["dil"])


def _apply_bn(sd, L, x, train, stats_group=None):
    """nn.BatchNorm2d forward (train: batch statistics over N*H*W, biased var
    for normalisation, unbiased into running_var, momentum 0.1; eval: running
    statistics).  ``stats_group`` is None for a single process."""
    n = L["name"]
    rm, rv = sd[n + ".running_mean"], sd[n + ".running_var"]
    if train:
        sd[n + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[n + ".weight"], sd[n + ".bias"], training=train,
                        momentum=BN_MOMENTUM, eps=BN_EPS)


def _bottleneck(sd, blk, x, train):
    """Bottleneck.forward, generalframeworks/networks/resnet.py:119-139."""
    out = F.relu(_apply_bn(sd, blk["bn1"], _apply_conv(sd, blk["conv1"], x), train))
    out = F.relu(_apply_bn(sd, blk["bn2"], _apply_conv(sd, blk["conv2"], out), train))
    out = _apply_bn(sd, blk["bn3"], _apply_conv(sd, blk["conv3"], out), train)
    identity = x
    if blk["downsample"] is not None:
        identity = _apply_bn(sd, blk["downsample"][1], _apply_conv(sd, blk["downsample"][0], x), train)
    return F.relu(out + identity)


def deeplab_forward(sd, x, backbone="tv", train=True, num_classes=21, output_dim=256,
                    return_intermediates=False):
    """DeepLabv3Plus_with_rep.forward, deeplabv3.py:151-169 (+ ASPP.forward
    aspp.py:67-72, ASPPPooling.forward aspp.py:35-38).  ``sd`` is a state_dict
    as produced by :func:`init_state`; BN running statistics in it are updated
    in place when ``train`` is true."""
    bs = backbone_spec(backbone)
    hs = head_spec(num_classes, output_dim)
    inter = {}
    # stem  (deeplabv3.py:152-153)
    h = x
    for L in bs["stem"]:
        if L["kind"] == "conv":
            h = _apply_conv(sd, L, h)
        else:
            h = F.relu(_apply_bn(sd, L, h, train))
    h = F.relu(_apply_bn(sd, bs["stem_bn"], h, train))
    h = F.max_pool2d(h, 3, 2, 1, ceil_mode=bs["maxpool_ceil"])
    inter["stem"] = h
    feats = []
    for li, blocks in enumerate(bs["layers"]):
        for blk in blocks:
            h = _bottleneck(sd, blk, h, train)
        feats.append(h)
        inter[f"layer{li + 1}"] = h
    x_low, x4 = feats[0], feats[3]
    # ASPP  (aspp.py:67-72)
    res = [F.relu(_apply_bn(sd, hs["aspp0"][1], _apply_conv(sd, hs["aspp0"][0], x4), train))]
    for c, b in hs["aspp_d"]:
        res.append(F.relu(_apply_bn(sd, b, _apply_conv(sd, c, x4), train)))
    c, b = hs["aspp_pool"]
    p = F.adaptive_avg_pool2d(x4, 1)
    p = F.relu(_apply_bn(sd, b, _apply_conv(sd, c, p), train))
    p = F.interpolate(p, size=x4.shape[-2:], mode="bilinear", align_corners=False)
    res.append(p)
    cat = torch.cat(res, dim=1)
    c, b = hs["aspp_proj"]
    feature = F.relu(_apply_bn(sd, b, _apply_conv(sd, c, cat), train))
    inter["aspp"] = feature
    # decoder (deeplabv3.py:163-166)
    c, b = hs["project"]
    xl = F.relu(_apply_bn(sd, b, _apply_conv(sd, c, x_low), train))
    up = F.interpolate(feature, size=xl.shape[2:], mode="bilinear", align_corners=True)
    dec = torch.cat([xl, up], dim=1)
    outs = []
    for key in ("classifier", "representation"):
        c0, b0, c1 = hs[key]
        t = F.relu(_apply_bn(sd, b0, _apply_conv(sd, c0, dec), train))
        outs.append(_apply_conv(sd, c1, t))
    if return_intermediates:
        return outs[0], outs[1], inter
    return outs[0], outs[1]


# --------------------------------------------------------------------------
# Step-wrapper pieces (W2-W4)
# --------------------------------------------------------------------------

def similarity(rep, prototypes):
    """cos(pixel embedding, class prototype): ddp_model.py:104-110 / :147-153.
    rep [B,C,h,w], prototypes [K,C] -> [B,K,h,w]."""
    b, c, h, w = rep.shape
    nr = F.normalize(rep.permute(0, 2, 3, 1), dim=-1).reshape(b * h * w, c)
    npt = F.normalize(prototypes, dim=-1).permute(1, 0)
    sim = torch.mm(nr, npt)
    return sim.reshape(b, h, w, prototypes.shape[0]).permute(0, 3, 1, 2)


def pseudo_labels_mix(pred_u, rep_u, prototypes, temp, out_size, num_classes):
    """Teacher half of Model_mix.forward, ddp_model.py:104-118."""
    sim = similarity(rep_u, prototypes)
    sim_large = F.interpolate(sim, size=out_size, mode="bilinear", align_corners=True)
    logits_rep, labels_rep = torch.max(F.softmax(sim_large / temp, dim=1), dim=1)
    pred_large = F.interpolate(pred_u, size=out_size, mode="bilinear", align_corners=True)
    logits_cls, labels_cls = torch.max(torch.softmax(pred_large, dim=1), dim=1)
    label_mask = (~labels_cls.eq(labels_rep)).float()
    pseudo = labels_cls - label_mask * num_classes
    pseudo[pseudo < 0] = 255
    return logits_rep, labels_rep, logits_cls, labels_cls, pseudo


def prob_all_from_rep(rep_all, prototypes, temp):
    """ddp_model.py:147-154."""
    return F.softmax(similarity(rep_all, prototypes) / temp, dim=1)


def identity_aug_label(pseudo_labels):
    """What ``batch_transform_2`` does to a label map when geometry is the
    identity: PIL round trip keeps integer classes, 255 -> -1
    (dataset_helpers/VOC.py:184-185).  Returns int64."""
    lab = pseudo_labels.long().clone()
    lab[lab == 255] = -1
    return lab


def ema_decay(step, alpha):
    """ddp_model.py:94."""
    return min(1 - 1 / (step + 1), alpha)


def ema_update(ema_params, params, step, alpha=0.99):
    """Model_mix.ema_update, ddp_model.py:93-97 (parameters only, not buffers)."""
    d = ema_decay(step, alpha)
    for e, p in zip(ema_params, params):
        e.data = d * e.data + (1 - d) * p.data
    return step + 1


# --------------------------------------------------------------------------
# Label / mask assembly (A1), schedules (A2, A3)
# --------------------------------------------------------------------------

def label_onehot(inputs, num_class):
    """generalframeworks/utils.py:116-125 (relu maps -1 -> class 0)."""
    b, h, w = inputs.shape
    inputs = torch.relu(inputs)
    out = torch.zeros([b, num_class, h, w])
    return out.scatter_(1, inputs.unsqueeze(1), 1.0)


def label_onehot_2(inputs, num_class):
    """generalframeworks/utils.py:127-136 (+1 shift, K+1 channels)."""
    b, h, w = inputs.shape
    inputs = inputs + 1
    out = torch.zeros([b, num_class + 1, h, w])
    return out.scatter_(1, inputs.unsqueeze(1), 1.0)


def build_label_mask(l_label, u_label, u_logits_cls, weak_threshold, num_class, out_hw):
    """mix_label.py:175-183."""
    u_mask = u_logits_cls.ge(weak_threshold).float()
    mask_all = torch.cat(((l_label.unsqueeze(1) >= 0).float(), u_mask.unsqueeze(1)))
    mask_all = F.interpolate(mask_all, size=out_hw, mode="nearest")
    label_l = F.interpolate(label_onehot(l_label, num_class), size=out_hw, mode="nearest")
    label_u = F.interpolate(label_onehot_2(u_label, num_class), size=out_hw, mode="nearest")
    label_u = label_u[:, 1:, :, :]
    return torch.cat((label_l, label_u)), mask_all


def poly_lr(base_lr, it, max_iters, power=0.9, min_lr=1e-4):
    """scheduler/my_lr_scheduler.py:11-13 (``it`` = last_epoch)."""
    return max(base_lr * (1 - it / max_iters) ** power, min_lr)


def rampdown_value(epoch, begin_epoch=0, max_epoch=200, max_value=1.0, min_value=0.0, mult=-5.0):
    """scheduler/rampscheduler.py:41-54."""
    if epoch < begin_epoch:
        v = 0.0
    elif epoch >= max_epoch:
        v = min_value
    else:
        v = max_value * float(np.exp(mult * (float(epoch - begin_epoch) / (max_epoch - begin_epoch)) ** 2))
    return max(v, min_value)


def sgd_nesterov_step(params, grads, bufs, lr, momentum=0.9, weight_decay=5e-4):
    """torch.optim.SGD(nesterov=True) as configured at mix_label.py:96-97.
    ``bufs`` entries may be None on the first step (buf = grad)."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        d = g + weight_decay * p.data
        if bufs[i] is None:
            bufs[i] = d.clone()
        else:
            bufs[i].mul_(momentum).add_(d)
        d = d + momentum * bufs[i]
        p.data.add_(d, alpha=-lr)


# --------------------------------------------------------------------------
# Losses (L1-L3)
# --------------------------------------------------------------------------

def attention_threshold_loss(pred, pseudo_label, logits, strong_threshold):
    """Attention_Threshold_Loss.forward, loss/loss.py:53-64."""
    b = pred.shape[0]
    valid_mask = (pseudo_label >= 0).float()
    weighting = logits.view(b, -1).ge(strong_threshold).sum(-1) / (valid_mask.view(b, -1).sum(-1))
    loss = F.cross_entropy(pred, pseudo_label, reduction="none", ignore_index=-1)
    return torch.mean(torch.masked_select(weighting[:, None, None] * loss, loss > 0))


def ce_loss(pred, target):
    """nn.CrossEntropyLoss(ignore_index=-1), mix_label.py:81,169."""
    return F.cross_entropy(pred, target, ignore_index=-1)


def prob_ohem_ce(pred, target, ignore_label=-1, thresh=0.7, min_kept=256):
    """ProbOhemCrossEntropy2d.forward, loss/loss.py:19-46."""
    b, c, h, w = pred.size()
    target = target.reshape(-1).clone()
    valid_mask = target.ne(ignore_label)
    target = target * valid_mask.long()
    num_valid = valid_mask.sum()
    prob = F.softmax(pred, dim=1)
    prob = (prob.transpose(0, 1)).reshape(c, -1)
    if min_kept > num_valid:
        pass                                     # loss.py:28-29 (prints, keeps all valid)
    elif num_valid > 0:
        prob = prob.masked_fill(~valid_mask, 1)
        mask_prob = prob[target, torch.arange(len(target), dtype=torch.long)]
        threshold = thresh
        if min_kept > 0:
            index = mask_prob.argsort()
            threshold_index = index[min(len(index), min_kept) - 1]
            if mask_prob[threshold_index] > thresh:
                threshold = mask_prob[threshold_index]
            kept_mask = mask_prob.le(threshold)
            target = target * kept_mask.long()
            valid_mask = valid_mask * kept_mask
    target = target.masked_fill(~valid_mask, ignore_label).view(b, h, w)
    return F.cross_entropy(pred, target, ignore_index=ignore_label)


def negative_index_sampler(samp_num, seg_num_list, rng=np.random):
    """loss/loss.py:410-418."""
    negative_index = []
    for i in range(samp_num.shape[0]):
        for j in range(samp_num.shape[1]):
            negative_index += rng.randint(low=sum(seg_num_list[:j]), high=sum(seg_num_list[:j + 1]),
                                          size=int(samp_num[i, j])).tolist()
    return negative_index


def contrast_loss(rep, label, mask, prob, prototypes, num_queries, num_negatives, temp=0.5,
                  strong_threshold=0.97, alpha=0.99, injected: Optional[dict] = None,
                  gather=None, record: Optional[dict] = None):
    """Contrast_Loss.forward, loss/loss.py:75-149.

    ``prototypes`` [K,C] is updated IN PLACE like the reference (:105,:108).
    ``gather(t)`` stands for ``concat_all_gather`` (ddp_model.py:241-251);
    None = single process.  ``injected`` carries the sampler outputs per locally
    present class index v: ``{"anchor": [V][Q] int, "negative": [V][Q*N] int}``
    (positions into the hard list of class v / into the concatenation of the
    other classes' valid lists in cyclic order v+1..v-1).  Without it the three
    RNG calls of the reference (:127, :137, :414) are made here; ``record``
    receives what was drawn.
    """
    rep_prt = gather(rep.detach()) if gather is not None else rep.detach()
    batch_size, num_feat, rep_w, rep_h = rep.shape
    num_segments = label.shape[1]
    valid_pixel_all = label * mask
    valid_pixel_all_prt = gather(valid_pixel_all) if gather is not None else valid_pixel_all
    rep = rep.permute(0, 2, 3, 1)
    rep_prt = rep_prt.permute(0, 2, 3, 1)

    rep_all_list, rep_hard_list, num_list, proto_rep_list, present = [], [], [], [], []
    for i in range(num_segments):
        valid_pixel = valid_pixel_all[:, i]
        valid_pixel_gather = valid_pixel_all_prt[:, i]
        if valid_pixel.sum() == 0:
            continue
        prob_seg = prob[:, i, :, :]
        rep_mask_hard = (prob_seg < strong_threshold) * valid_pixel.bool()
        with torch.no_grad():
            proto_rep_ = torch.mean(rep_prt[valid_pixel_gather.bool()], dim=0, keepdim=True)
            if prototypes[i].sum() == torch.tensor(0.0):
                proto_rep_list.append(proto_rep_)
                prototypes[i] = proto_rep_
            else:
                prototypes[i] = alpha * prototypes[i] + (1 - alpha) * proto_rep_
                proto_rep_list.append(prototypes[i].unsqueeze(0))
        rep_all_list.append(rep[valid_pixel.bool()])
        rep_hard_list.append(rep[rep_mask_hard])
        num_list.append(int(valid_pixel.sum().item()))
        present.append(i)

    if record is not None:
        record.update(present=present, num_list=list(num_list),
                      hard_num=[len(r) for r in rep_hard_list], anchor=[], negative=[])
    if len(num_list) <= 1:
        return torch.tensor(0.0) + 0 * rep.sum()
    loss = torch.tensor(0.0)
    proto_rep = torch.cat(proto_rep_list)
    valid_num = len(num_list)
    seg_len = torch.arange(valid_num)
    for i in range(valid_num):
        if len(rep_hard_list[i]) == 0:
            if record is not None:
                record["anchor"].append(None)
                record["negative"].append(None)
            continue
        if injected is not None:
            # (modulo: a no-op for draws recorded on these very lists; lets the draws of a neighbouring evaluation - a 1-ulp-perturbed re-run
            # whose hard list is one pixel shorter - be replayed, with the rule css_contrast_resolve applies on the device)
            sample_idx = torch.as_tensor(injected["anchor"][i], dtype=torch.long) % len(rep_hard_list[i])
        else:
            sample_idx = torch.randint(len(rep_hard_list[i]), size=(num_queries,))
        anchor_rep = rep_hard_list[i][sample_idx]
        with torch.no_grad():
            id_mask = torch.cat(([seg_len[i:], seg_len[:i]]))
            if injected is not None:
                negative_index = list(injected["negative"][i])
            else:
                proto_sim = torch.cosine_similarity(proto_rep[id_mask[0]].unsqueeze(0), proto_rep[id_mask[1:]], dim=1)
                proto_prob = torch.softmax(proto_sim / temp, dim=0)
                dist = torch.distributions.categorical.Categorical(probs=proto_prob)
                samp_class = dist.sample(sample_shape=[num_queries, num_negatives])
                samp_num = torch.stack([(samp_class == c).sum(1) for c in range(len(proto_prob))], dim=1)
                negative_num_list = num_list[i + 1:] + num_list[:i]
                negative_index = negative_index_sampler(samp_num, negative_num_list)
            negative_rep_all = torch.cat(rep_all_list[i + 1:] + rep_all_list[:i])
            if injected is not None:
                negative_index = (torch.as_tensor(negative_index, dtype=torch.long) % len(negative_rep_all)).tolist()
            negative_rep = negative_rep_all[negative_index].reshape(num_queries, num_negatives, num_feat)
            positive_rep = proto_rep[i].unsqueeze(0).unsqueeze(0).repeat(num_queries, 1, 1)
            all_rep = torch.cat((positive_rep, negative_rep), dim=1)
        if record is not None:
            record["anchor"].append(sample_idx.tolist())
            record["negative"].append(list(negative_index))
        logits = torch.cosine_similarity(anchor_rep.unsqueeze(1), all_rep, dim=2)
        loss = loss + F.cross_entropy(logits / temp, torch.zeros(num_queries).long())
    return loss / valid_num


def class_negative_probs(proto_rep, v, temp):
    """Categorical distribution over the other present classes for anchor class
    ``v`` in cyclic order v+1..v-1 (loss.py:133-135).  proto_rep [V,C]."""
    V = proto_rep.shape[0]
    order = list(range(v + 1, V)) + list(range(0, v))
    sim = torch.cosine_similarity(proto_rep[v].unsqueeze(0), proto_rep[order], dim=1)
    return torch.softmax(sim / temp, dim=0), order


# --------------------------------------------------------------------------
# Whole step (mix_label.train body, mix_label.py:162-196) with identity aug
# --------------------------------------------------------------------------

class MixState:
    """Student + EMA teacher + optimiser state for the oracle's train step."""

    def __init__(self, backbone="tv", num_classes=21, output_dim=256, seed=0, residual_gain=1.0):
        self.backbone, self.num_classes, self.output_dim = backbone, num_classes, output_dim
        self.student = init_state(backbone, num_classes, output_dim, seed, residual_gain)
        self.teacher = OrderedDict((k, v.clone()) for k, v in self.student.items())   # copy.deepcopy, ddp_model.py:85
        self.pnames = param_names(backbone, num_classes, output_dim)
        self.mom = [None] * len(self.pnames)
        self.step = 0
        self.prototypes = torch.zeros(num_classes, output_dim)


def train_step_mix(st: MixState, l_img, l_lab, u_img, *, lr, temp_model=0.5, strong_threshold=0.8,
                   weak_threshold=0.7, un_threshold=0.97, num_queries=256, num_negatives=512,
                   temp_loss=0.5, alpha_proto=0.99, ema_alpha=0.99, ramp=1.0, sup="ce",
                   ohem_min_kept=None, injected=None, record=None):
    """One iteration of mix_label.train (mix_label.py:162-196) with the
    augmentation replaced by the identity (``mix_mode='none'``, VOC.py:404-405;
    label 255 -> -1, VOC.py:184-185).  Returns a dict of scalars / tensors."""
    K = st.num_classes
    H, W = l_img.shape[2:]
    for n in st.pnames:
        st.student[n].requires_grad_(True)
        st.student[n].grad = None
    with torch.no_grad():
        # teacher on labeled images: outputs unused but BN running stats move
        deeplab_forward(st.teacher, l_img, st.backbone, True, K, st.output_dim)
        pred_u, rep_u = deeplab_forward(st.teacher, u_img, st.backbone, True, K, st.output_dim)
        lg_rep, lb_rep, lg_cls, lb_cls, pseudo = pseudo_labels_mix(pred_u, rep_u, st.prototypes, temp_model, (H, W), K)
        u_aug_label = identity_aug_label(pseudo)
        u_aug_img, u_aug_lg_cls, u_aug_lg_rep = u_img, lg_cls, lg_rep
    pred_l, rep_l = deeplab_forward(st.student, l_img, st.backbone, True, K, st.output_dim)
    pred_l_large = F.interpolate(pred_l, size=(H, W), mode="bilinear", align_corners=True)
    pred_u2, rep_u2 = deeplab_forward(st.student, u_aug_img, st.backbone, True, K, st.output_dim)
    pred_u_large = F.interpolate(pred_u2, size=(H, W), mode="bilinear", align_corners=True)
    rep_all = torch.cat((rep_l, rep_u2))
    with torch.no_grad():
        prob_all = prob_all_from_rep(rep_all, st.prototypes, temp_model)

    if sup == "ce":
        sup_loss = ce_loss(pred_l_large, l_lab)
    else:
        sup_loss = prob_ohem_ce(pred_l_large, l_lab, -1, 0.7, ohem_min_kept)
    unsup_loss = attention_threshold_loss(pred_u_large, u_aug_label, u_aug_lg_cls, un_threshold)
    with torch.no_grad():
        label_all, mask_all = build_label_mask(l_lab, u_aug_label, u_aug_lg_cls, weak_threshold, K, rep_all.shape[2:])
    c_loss = contrast_loss(rep_all, label_all, mask_all, prob_all, st.prototypes, num_queries, num_negatives,
                           temp_loss, strong_threshold, alpha_proto, injected=injected, record=record)
    total = sup_loss + unsup_loss + c_loss * ramp
    total.backward()
    params = [st.student[n] for n in st.pnames]
    grads = [p.grad for p in params]
    with torch.no_grad():
        sgd_nesterov_step(params, grads, st.mom, lr)
        st.step = ema_update([st.teacher[n] for n in st.pnames], params, st.step, ema_alpha)
    for n in st.pnames:
        st.student[n].requires_grad_(False)
    return dict(sup=float(sup_loss), unsup=float(unsup_loss), contrast=float(c_loss), total=float(total),
                pred_l=pred_l.detach(), rep_all=rep_all.detach(), pseudo=u_aug_label, grads=grads)


def build_label_mask_relu(l_label, u_label, u_logits, weak_threshold, num_class, out_hw):
    """cross_label.py:180-188 / ori_pseudo.py:170-177: as build_label_mask, but the unlabeled half goes through ``label_onehot``
    (ReLU: an ignored -1 pseudo label becomes class 0) instead of ``label_onehot_2``."""
    u_mask = u_logits.ge(weak_threshold).float()
    mask_all = torch.cat(((l_label.unsqueeze(1) >= 0).float(), u_mask.unsqueeze(1)))
    mask_all = F.interpolate(mask_all, size=out_hw, mode="nearest")
    label_l = F.interpolate(label_onehot(l_label, num_class), size=out_hw, mode="nearest")
    label_u = F.interpolate(label_onehot(u_label, num_class), size=out_hw, mode="nearest")
    return torch.cat((label_l, label_u)), mask_all


def train_step_w5(st: MixState, kind, l_img, l_lab, u_img, *, lr, temp_model=0.5, strong_threshold=0.8, weak_threshold=0.7,
                  un_threshold=0.97, num_queries=256, num_negatives=512, temp_loss=0.5, alpha_proto=0.99, ema_alpha=0.99, ramp=1.0,
                  warmup=True, injected=None, record=None):
    """One iteration of the train body of cross_label.py:162-198 (``kind='cross'``: Model_cross.forward ddp_model.py:184-239;
    ``warmup`` = epoch < args.warmup selects the class-predictor pseudo labels for the unsupervised loss, :174-177) or of
    ori_pseudo.py:158-187 (``kind='ori'``: Model_ori_pseudo.forward ddp_model.py:32-70, teacher on the unlabeled batch only,
    prob_all = softmax of the student's logits :178, no ramp), identity augmentation as in train_step_mix."""
    K = st.num_classes
    H, W = l_img.shape[2:]
    for n in st.pnames:
        st.student[n].requires_grad_(True)
        st.student[n].grad = None
    with torch.no_grad():
        if kind == "cross":
            deeplab_forward(st.teacher, l_img, st.backbone, True, K, st.output_dim)           # ddp_model.py:187 (running stats)
            pred_u, rep_u = deeplab_forward(st.teacher, u_img, st.backbone, True, K, st.output_dim)
            lg_rep, lb_rep, lg_cls, lb_cls, _ = pseudo_labels_mix(pred_u, rep_u, st.prototypes, temp_model, (H, W), K)
            lab_cls, lab_rep = identity_aug_label(lb_cls), identity_aug_label(lb_rep)
        else:
            pred_u, _ = deeplab_forward(st.teacher, u_img, st.backbone, True, K, st.output_dim)   # ddp_model.py:35
            pred_large = F.interpolate(pred_u, size=(H, W), mode="bilinear", align_corners=True)
            lg_cls, lb_cls = torch.max(torch.softmax(pred_large, dim=1), dim=1)
            lab_cls = identity_aug_label(lb_cls)
    pred_l, rep_l = deeplab_forward(st.student, l_img, st.backbone, True, K, st.output_dim)
    pred_l_large = F.interpolate(pred_l, size=(H, W), mode="bilinear", align_corners=True)
    pred_u2, rep_u2 = deeplab_forward(st.student, u_img, st.backbone, True, K, st.output_dim)
    pred_u_large = F.interpolate(pred_u2, size=(H, W), mode="bilinear", align_corners=True)
    rep_all = torch.cat((rep_l, rep_u2))
    with torch.no_grad():
        if kind == "cross":
            prob_all = prob_all_from_rep(rep_all, st.prototypes, temp_model)
        else:
            prob_all = torch.softmax(torch.cat((pred_l, pred_u2)), dim=1)                   # ori_pseudo.py:178 on ddp_model.py:61
    sup_loss = ce_loss(pred_l_large, l_lab)
    if kind == "cross" and not warmup:
        unsup_loss = attention_threshold_loss(pred_u_large, lab_rep, lg_rep, un_threshold)
    else:
        unsup_loss = attention_threshold_loss(pred_u_large, lab_cls, lg_cls, un_threshold)
    with torch.no_grad():
        label_all, mask_all = build_label_mask_relu(l_lab, lab_cls, lg_cls, weak_threshold, K, rep_all.shape[2:])
    c_loss = contrast_loss(rep_all, label_all, mask_all, prob_all, st.prototypes, num_queries, num_negatives,
                           temp_loss, strong_threshold, alpha_proto, injected=injected, record=record)
    total = sup_loss + unsup_loss + c_loss * (ramp if kind == "cross" else 1.0)
    total.backward()
    params = [st.student[n] for n in st.pnames]
    grads = [p.grad for p in params]
    with torch.no_grad():
        sgd_nesterov_step(params, grads, st.mom, lr)
        st.step = ema_update([st.teacher[n] for n in st.pnames], params, st.step, ema_alpha)
    for n in st.pnames:
        st.student[n].requires_grad_(False)
    return dict(sup=float(sup_loss), unsup=float(unsup_loss), contrast=float(c_loss), total=float(total), pseudo=lab_cls,
                logits_cls=lg_cls, pseudo_rep=lab_rep if kind == "cross" else None)


# ---------------------------------------------------------------------------
# evaluation path: mix_label.py:199-225, util/meter.py:39-48, util/miou.py:3-9
# ---------------------------------------------------------------------------
def eval_confusion(pred, label, num_classes):
    """pred [B,K,h,w] fp32 logits, label [B,H,W] int -> (K x K int64 matrix (row = target, col = prediction), argmax [B,H,W]).
    mix_label.py:214-216: bilinear(align_corners=True) to the label size, argmax(1); meter.py:44-47: targets outside [0,K)
    are dropped, inds = K*target + pred, bincount."""
    up = F.interpolate(pred.float(), size=label.shape[1:], mode="bilinear", align_corners=True)
    am = up.argmax(1)
    t, p = label.reshape(-1).long(), am.reshape(-1)
    k = (t >= 0) & (t < num_classes)
    mat = torch.bincount(num_classes * t[k] + p[k], minlength=num_classes ** 2).reshape(num_classes, num_classes)
    return mat, am


def mean_iou(mat):
    """util/miou.py:3-9 (no epsilon: a class absent from both prediction and target gives NaN, like the reference)."""
    h = mat.float()
    iu = torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))
    return torch.mean(iu).item()
