"""HIP-backed losses with the reference's call signatures (generalframeworks/loss/loss.py):

    Contrast_Loss(num_queries, num_negatives, temp=0.5, mean=False, strong_threshold=0.97, alpha=0.99)
        .forward(rep, label, mask, prob, prototypes) -> 0-dim tensor        loss.py:66-149
    Attention_Threshold_Loss(strong_threshold).forward(pred, pseudo_label, logits)   loss.py:48-64
    ProbOhemCrossEntropy2d(ignore_label, reduction, thresh, min_kept, ...).forward(pred, target)   loss.py:8-46
    CrossEntropyLoss(ignore_index=-1)   (what mix_label.py:81 builds from torch.nn)

No host synchronisation anywhere: class presence, counts and the "fewer than two classes" early-out
(loss.py:116-117) are resolved on the device.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _lib, ops
from .._lib import call, dev_stream, dtype_code, query
from ..functional import nhwc


# --------------------------------------------------------------------------
# pixel-wise cross-entropy family
# --------------------------------------------------------------------------
class _PixelCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, conf, conf_thr, mode, ohem):
        b, k, h, w = pred.shape
        x = nhwc(pred.detach().float())                  # [B,H,W,K] fp32 (zero-copy for channels_last fp32 input)
        label = label.contiguous()
        p = b * h * w
        dev, st = dev_stream(x)
        f64 = dict(dtype=torch.int64, device=x.device)      # integer accumulators (include/css_hip.h: order-independent sums)
        stats = torch.zeros(b * 4, **f64)
        keep = None
        if ohem is not None:
            min_kept, thresh = ohem
            gtprob = torch.empty(p, dtype=torch.float32, device=x.device)
            call("css_ce_fwd", x, label, None, 0.0, None, k, p, h * w, stats, gtprob, dev, st)
            state = torch.zeros(query("css_ohem_state_bytes"), dtype=torch.uint8, device=x.device)
            call("css_ohem_threshold", gtprob, p, stats, b, int(min_kept), float(thresh), state, dev, st)
            keep = state[query("css_ohem_thr_offset"):]
            stats = torch.zeros(b * 4, **f64)
            call("css_ce_fwd", x, label, None, 0.0, keep, k, p, h * w, stats, None, dev, st)
        else:
            call("css_ce_fwd", x, label, conf, float(conf_thr), None, k, p, h * w, stats, None, dev, st)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        coef = torch.empty(b, dtype=torch.float32, device=x.device)
        call("css_ce_finalize", stats, b, int(mode), loss, coef, dev, st)
        ctx.save_for_backward(x, label, coef, keep)
        ctx.cfg = (mode, pred.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, label, coef, keep = ctx.saved_tensors
        mode, in_dtype = ctx.cfg
        b, h, w, k = x.shape
        dev, st = dev_stream(x)
        gs = g.detach().float().reshape(1).contiguous()
        dx = torch.empty_like(x)
        call("css_ce_bwd", x, label, keep, k, b * h * w, h * w, coef, gs, int(mode == 1), dx, dev, st)
        return dx.permute(0, 3, 1, 2).to(in_dtype), None, None, None, None, None


class _PixelCESmall(torch.autograd.Function):
    """The same losses from the LOW-resolution NHWC logits ``small`` [B,h,w,K]: the bilinear(align_corners=True) up-sampling to the
    label size (ddp_model.py:141,144) is applied on the fly in forward and backward (css_ce_small_*), so the [B,K,H,W] fp32
    logits and their gradient (2 x 354 MB each way at c2) are never written."""

    @staticmethod
    def forward(ctx, small, label, conf, conf_thr, mode, ohem):
        b, h, w, k = small.shape
        x = small.detach()
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        label = label.contiguous()
        hh, ww = label.shape[1], label.shape[2]
        p = b * hh * ww
        dev, st = dev_stream(x)
        dc = dtype_code(x.dtype)
        f64 = dict(dtype=torch.int64, device=x.device)      # integer accumulators (include/css_hip.h: order-independent sums)
        stats = torch.zeros(b * 4, **f64)
        keep = None
        if ohem is not None:
            min_kept, thresh = ohem
            gtprob = torch.empty(p, dtype=torch.float32, device=x.device)
            call("css_ce_small_fwd", x, k, b, h, w, label, None, 0.0, None, k, hh, ww, stats, gtprob, dc, dev, st)
            state = torch.zeros(query("css_ohem_state_bytes"), dtype=torch.uint8, device=x.device)
            call("css_ohem_threshold", gtprob, p, stats, b, int(min_kept), float(thresh), state, dev, st)
            keep = state[query("css_ohem_thr_offset"):]
            stats = torch.zeros(b * 4, **f64)
            call("css_ce_small_fwd", x, k, b, h, w, label, None, 0.0, keep, k, hh, ww, stats, None, dc, dev, st)
        else:
            call("css_ce_small_fwd", x, k, b, h, w, label, conf, float(conf_thr), None, k, hh, ww, stats, None, dc, dev, st)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        coef = torch.empty(b, dtype=torch.float32, device=x.device)
        call("css_ce_finalize", stats, b, int(mode), loss, coef, dev, st)
        ctx.save_for_backward(x, label, coef, keep)
        ctx.cfg = (mode, small.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, label, coef, keep = ctx.saved_tensors
        mode, in_dtype = ctx.cfg
        b, h, w, k = x.shape
        hh, ww = label.shape[1], label.shape[2]
        dev, st = dev_stream(x)
        gs = g.detach().float().reshape(1).contiguous()
        dx = torch.zeros((b, h, w, k), dtype=torch.float32, device=x.device)
        call("css_ce_small_bwd", x, k, b, h, w, label, keep, k, hh, ww, coef, gs, int(mode == 1), dx, dtype_code(x.dtype), dev, st)
        return dx.to(in_dtype), None, None, None, None, None


def fused_upsample_ok(small_hw, label_hw, num_classes=21):
    """css_ce_small_bwd's tile footprint assumes an up-sampling factor >= 2 (513/129, 769/193 in the reference's configs: exactly 4);
    above a factor of 4, or with more than 24 classes, its adjoint inside a workgroup falls back to LDS float atomics - not
    bit-reproducible - so the trainer takes the (ordered) up-sample + full-resolution loss there."""
    (h, w), (hh, ww) = small_hw, label_hw
    return (2 * (h - 1) <= hh - 1 and 2 * (w - 1) <= ww - 1 and 4 * (h - 1) >= hh - 1 and 4 * (w - 1) >= ww - 1 and num_classes <= 24)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(ignore_index=-1) of mix_label.py:81 (mean over non-ignored pixels)."""

    def __init__(self, ignore_index=-1):
        super().__init__()
        assert ignore_index < 0
        self.ignore_index = ignore_index

    def forward(self, pred, target):
        return _PixelCE.apply(pred, target, None, 0.0, 0, None)

    def forward_small(self, small, target):
        """small: the network's NHWC logits [B,h,w,K] before up-sampling; == forward(F.interpolate(.., align_corners=True), target)."""
        return _PixelCESmall.apply(small, target, None, 0.0, 0, None)


class Attention_Threshold_Loss(nn.Module):
    def __init__(self, strong_threshold):
        super().__init__()
        self.strong_threshold = strong_threshold

    def forward(self, pred, pseudo_label, logits):
        return _PixelCE.apply(pred, pseudo_label, logits.detach().float().contiguous(), self.strong_threshold, 1, None)

    def forward_small(self, small, pseudo_label, logits):
        return _PixelCESmall.apply(small, pseudo_label, logits.detach().float().contiguous(), self.strong_threshold, 1, None)


class ProbOhemCrossEntropy2d(nn.Module):
    def __init__(self, ignore_label, reduction="mean", thresh=0.6, min_kept=256, down_ratio=1, use_weight=False):
        super().__init__()
        assert reduction == "mean" and ignore_label < 0
        self.ignore_label, self.thresh, self.min_kept, self.down_ratio = ignore_label, float(thresh), int(min_kept), down_ratio

    def forward(self, pred, target):
        return _PixelCE.apply(pred, target, None, 0.0, 0, (self.min_kept, self.thresh))

    def forward_small(self, small, target):
        return _PixelCESmall.apply(small, target, None, 0.0, 0, (self.min_kept, self.thresh))


# --------------------------------------------------------------------------
# contrastive loss
# --------------------------------------------------------------------------
class _ContrastCore(torch.autograd.Function):
    """rep [P, C] rows (NHWC-flattened), cls int32 [P], hard uint8 [P]; prototypes updated in place."""

    @staticmethod
    def forward(ctx, rep, cls, hard, prototypes, K, Q, N, temp, alpha, seed, offset, injected, group_sync):
        rep_d = rep.detach()
        assert rep_d.is_contiguous() and prototypes.dtype == torch.float32 and prototypes.is_contiguous()
        P, C = rep_d.shape
        dev, st = dev_stream(rep_d)
        dc = dtype_code(rep_d.dtype)
        d = rep_d.device
        i32 = dict(dtype=torch.int32, device=d)
        meta = torch.zeros(query("css_contrast_meta_bytes"), dtype=torch.uint8, device=d)
        sums = torch.empty(K * C + K, dtype=torch.float64, device=d)
        ws = torch.empty(query("css_contrast_class_sums_ws_bytes", P, K, C) // 4, dtype=torch.float32, device=d)
        call("css_contrast_class_sums", rep_d, C, cls, P, K, C, sums, ws, dc, dev, st)
        nch = query("css_contrast_nchunks", P)
        chunkhist = torch.empty(nch * 64, **i32)
        listV, listH = torch.empty(P, **i32), torch.empty(P, **i32)
        call("css_contrast_compact", cls, hard, P, K, chunkhist, listV, listH, meta, dev, st)
        if group_sync and ops.collectives_on():
            # replaces the two all_gathers of loss.py:77,81 (545 MB/rank) by K*(C+1) numbers: mean = sum/count
            dist.all_reduce(sums)
        call("css_contrast_proto_update", prototypes, sums, K, C, float(alpha), meta, dev, st)
        anchor_pix = torch.zeros(K * Q, **i32)
        neg_pix = torch.zeros(K * Q * N, **i32)
        if injected is None:
            cdf = torch.zeros(32 * 32, dtype=torch.float32, device=d)
            call("css_contrast_sample", prototypes, C, meta, float(temp), cdf, listV, listH, Q, N, int(seed), int(offset),
                 anchor_pix, neg_pix, dev, st)
        else:
            a_idx, n_idx = injected
            call("css_contrast_resolve", meta, listV, listH, Q, N, a_idx, n_idx, anchor_pix, neg_pix, dev, st)
        loss_vq = torch.zeros(K * Q, dtype=torch.float32, device=d)
        gradbuf = torch.empty(K * Q * C, dtype=torch.float32, device=d)
        loss = torch.empty(1, dtype=torch.float32, device=d)
        call("css_contrast_loss", rep_d, C, prototypes, K, C, meta, anchor_pix, neg_pix, Q, N, float(temp), loss_vq, gradbuf, loss,
             dc, dev, st)
        ctx.save_for_backward(gradbuf, anchor_pix, meta)
        ctx.cfg = (P, C, K, Q, rep_d.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        gradbuf, anchor_pix, meta = ctx.saved_tensors
        P, C, K, Q, dt = ctx.cfg
        dev, st = dev_stream(gradbuf)
        drep = torch.zeros((P, C), dtype=dt, device=gradbuf.device)
        gs = g.detach().float().reshape(1).contiguous()
        call("css_contrast_scatter_grad", gradbuf, anchor_pix, meta, K, Q, gs, drep, C, dtype_code(dt), dev, st)
        return (drep,) + (None,) * 12


def _flat_strides(t: torch.Tensor):
    """(batch stride, channel stride, pixel stride) of a [B,K,h,w] tensor whose (h,w) plane is jointly flattenable."""
    sb, sk, sh, sw = t.stride()
    if sh != t.shape[3] * sw:
        return None
    return sb, sk, sw


class Contrast_Loss(nn.Module):
    def __init__(self, num_queries, num_negatives, temp=0.5, mean=False, strong_threshold=0.97, alpha=0.99):
        super().__init__()
        self.temp, self.mean = temp, mean
        self.num_queries, self.num_negatives = num_queries, num_negatives
        self.strong_threshold, self.alpha = strong_threshold, alpha
        self._calls = 0
        self.last = None       # debug handles of the last call (device tensors), used by the parity tests

    def _seed(self):
        self._calls += 1
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        return (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1)) & 0xFFFFFFFFFFFFFFFF, self._calls

    def _injected_tensors(self, injected, device):
        if injected is None:
            return None
        Q, N = self.num_queries, self.num_negatives
        a = torch.zeros((32, Q), dtype=torch.int32)
        n = torch.zeros((32, Q * N), dtype=torch.int32)
        for v, (ai, ni) in enumerate(zip(injected["anchor"], injected["negative"])):
            if ai is not None:
                a[v] = torch.as_tensor(ai, dtype=torch.int32)
                n[v] = torch.as_tensor(ni, dtype=torch.int32)
        return a.to(device).contiguous(), n.to(device).contiguous()

    def forward(self, rep, label, mask, prob, prototypes, _injected=None):
        """Reference signature (loss.py:75).  rep [2B,C,h,w] (grad), label [2B,K,h,w] one-hot, mask [2B,1,h,w],
        prob [2B,K,h,w], prototypes [K,C] fp32 -- updated IN PLACE."""
        b2, c, h, w = rep.shape
        k = label.shape[1]
        p = b2 * h * w
        rep_rows = rep.permute(0, 2, 3, 1)
        if not rep_rows.is_contiguous():
            rep_rows = rep_rows.contiguous()
        rep_rows = rep_rows.reshape(p, c)
        label, prob = label.detach().float(), prob.detach().float()
        ls, ps = _flat_strides(label), _flat_strides(prob)
        if ls is None:
            label = label.contiguous()
            ls = _flat_strides(label)
        if ps is None:
            prob = prob.contiguous()
            ps = _flat_strides(prob)
        mask = mask.detach().float().contiguous()
        d = rep.device
        cls = torch.empty(p, dtype=torch.int32, device=d)
        hard = torch.empty(p, dtype=torch.uint8, device=d)
        meta_err = torch.zeros(_lib.query("css_contrast_meta_bytes"), dtype=torch.uint8, device=d)
        dev, st = dev_stream(rep_rows)
        call("css_contrast_classify", label, mask, prob, ls[0], ls[1], ls[2], ps[0], ps[1], ps[2], p, h * w, k,
             float(self.strong_threshold), cls, hard, meta_err, dev, st)
        return self.forward_fused(rep_rows, cls, hard, prototypes, k, _injected)

    def forward_fused(self, rep_rows, cls, hard, prototypes, num_classes, _injected=None):
        """Fast path of the trainer: class-id map (css_amd.functional.class_map) and hard flags (similarity kernel)
        instead of the dense one-hot label / mask / prob tensors."""
        seed, off = self._seed()
        inj = self._injected_tensors(_injected, rep_rows.device)
        loss = _ContrastCore.apply(rep_rows, cls, hard, prototypes, num_classes, self.num_queries, self.num_negatives,
                                   self.temp, self.alpha, seed, off, inj, True)
        return loss
