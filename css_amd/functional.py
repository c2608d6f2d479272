"""Functional layer over the similarity / pseudo-label / class-map kernels (no autograd: the reference runs all of
these under ``torch.no_grad()`` -- ddp_model.py:101-118,147-154; mix_label.py:175-183)."""
from __future__ import annotations

import torch

from ._lib import call, dev_stream, dtype_code


def nhwc(t: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor -> contiguous [N,H,W,C] view (zero-copy when the memory is already channels_last)."""
    return t.permute(0, 2, 3, 1).contiguous()


def normalized_prototypes(prototypes: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """F.normalize(prototypes, dim=-1) (ddp_model.py:107) as [32, C] in the compute dtype (rows >= K are zero)."""
    k, c = prototypes.shape
    p = prototypes.detach().float().contiguous()
    out = torch.empty((32, c), dtype=dtype, device=p.device)
    dev, st = dev_stream(p)
    call("css_proto_normalize", p, out, k, c, dtype_code(dtype), dev, st)
    return out


def similarity(rep_nhwc: torch.Tensor, prototypes: torch.Tensor, temp: float, want_sim=False, want_prob=False,
               cls: torch.Tensor = None, strong_threshold: float = 0.0):
    """cos(embedding, prototype) for every pixel on MFMA.  rep_nhwc [B,h,w,C] (fp32/bf16), prototypes [K,C] fp32.
    Returns (sim [B,h,w,K] f32 | None, prob = softmax(sim/temp) [B,h,w,K] f32 | None, hard [B*h*w] u8 | None)."""
    rep = rep_nhwc.detach()
    assert rep.is_contiguous()
    b, h, w, c = rep.shape
    k = prototypes.shape[0]
    p = b * h * w
    pn = normalized_prototypes(prototypes, rep.dtype)
    dev, st = dev_stream(rep)
    sim = torch.empty((b, h, w, k), dtype=torch.float32, device=rep.device) if want_sim else None
    prob = torch.empty((b, h, w, k), dtype=torch.float32, device=rep.device) if want_prob else None
    hard = torch.empty((p,), dtype=torch.uint8, device=rep.device) if cls is not None else None
    call("css_similarity", rep, c, pn, sim, prob, cls, hard, p, k, c, float(temp), float(strong_threshold),
         dtype_code(rep.dtype), dev, st)
    return sim, prob, hard


def pseudo_labels(sim: torch.Tensor, pred_nhwc: torch.Tensor, temp: float, out_hw):
    """Teacher pseudo-labelling (ddp_model.py:111-118): bilinear x4 (align_corners) + softmax + max in rep space and cls
    space + agreement mask.  Returns (logits_rep f32, labels_rep i64, logits_cls f32, labels_cls i64, pseudo f32 with
    255 where the two label maps disagree), all [B,H,W]."""
    b, h, w, k = sim.shape
    hh, ww = out_hw
    pred = pred_nhwc.detach()
    assert pred.is_contiguous() and sim.is_contiguous() and pred.shape[:3] == sim.shape[:3]
    f = dict(device=sim.device)
    lr = torch.empty((b, hh, ww), dtype=torch.float32, **f)
    lc = torch.empty((b, hh, ww), dtype=torch.float32, **f)
    ar = torch.empty((b, hh, ww), dtype=torch.int64, **f)
    ac = torch.empty((b, hh, ww), dtype=torch.int64, **f)
    ps = torch.empty((b, hh, ww), dtype=torch.float32, **f)
    dev, st = dev_stream(sim)
    call("css_pseudo_label", sim, pred, pred.shape[-1], b, h, w, k, hh, ww, float(temp), lr, ar, lc, ac, ps,
         dtype_code(pred.dtype), dev, st)
    return lr, ar, lc, ac, ps


def class_map(l_lab: torch.Tensor, u_lab: torch.Tensor, u_logits: torch.Tensor, weak_threshold: float, out_hw):
    """mask_all / label_all of mix_label.py:175-183 folded into one class-id map at embedding resolution:
    int32 [2B*h*w], -1 = not a valid pixel of any class (nearest down-sampling like F.interpolate(mode='nearest'))."""
    b, hh, ww = l_lab.shape
    h, w = out_hw
    cls = torch.empty((2 * b * h * w,), dtype=torch.int32, device=l_lab.device)
    dev, st = dev_stream(l_lab)
    call("css_class_map", l_lab.contiguous(), u_lab.contiguous(), u_logits.float().contiguous(), float(weak_threshold), b, hh, ww,
         h, w, cls, dev, st)
    return cls
