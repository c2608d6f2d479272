"""``ConfMatrix`` / ``AverageMeter`` with the reference's constructor and ``update`` signatures
(generalframeworks/util/meter.py:4-60).  The K x K confusion matrix (int64, row = target, column = prediction) lives on the
device and is accumulated by HIP kernels: ``update(pred, target)`` takes class indices like the reference;
``update_from_logits(pred_small, target)`` additionally fuses the bilinear(align_corners=True) up-sampling and the argmax of
``test()`` (mix_label.py:213-216) so the full-resolution logits are never materialised."""
from __future__ import annotations

import torch

from .._lib import CssHipError, call, dev_stream
from ..ops import dtype_code


class AverageMeter(object):
    """Computes and stores the average and current value (util/meter.py:4-25)."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


def _iou_mean(mat: torch.Tensor) -> float:
    h = mat.float()
    iu = torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))
    return torch.mean(iu).item()


class ConfMatrix(object):
    def __init__(self, num_classes, fmt=":6.4f", name="miou"):
        self.name, self.fmt, self.num_classes = name, fmt, num_classes
        self.mat = None
        self.temp_mat = None
        self.val = 0
        self.avg = 0

    def _ensure(self, device):
        if not torch.device(device).type == "cuda":
            raise CssHipError("css_amd.util.ConfMatrix runs on the MI355X only (no CPU path)")
        n = self.num_classes
        if self.mat is None:
            self.mat = torch.zeros((n, n), dtype=torch.int64, device=device)
        self.temp_mat = torch.zeros((n, n), dtype=torch.int64, device=device)

    @torch.no_grad()
    def update(self, pred, target):
        """pred, target: flat class-index tensors; targets outside [0, K) are ignored (util/meter.py:39-48)."""
        self._ensure(pred.device)
        p = pred.reshape(-1).to(torch.int64).contiguous()
        t = target.reshape(-1).to(torch.int64).contiguous()
        if p.numel() != t.numel():
            raise ValueError("pred and target must have the same number of elements")
        dev, st = dev_stream(p)
        call("css_confusion_bincount", p, t, p.numel(), self.num_classes, self.temp_mat, dev, st)
        self.mat += self.temp_mat

    @torch.no_grad()
    def update_from_logits(self, pred, target, return_argmax=False, channels_last=False):
        """pred: the network's low-resolution logits, logical [B,K,h,w] (channels_last memory, as the HIP model returns them), or -
        with ``channels_last=True`` - an NHWC tensor [B,h,w,K]; the layout is never guessed from the shape (h or w may equal K).
        target: int64 [B,H,W].  Equivalent to
        ``update(F.interpolate(pred, target.shape[1:], mode='bilinear', align_corners=True).argmax(1).flatten(), target.flatten())``."""
        self._ensure(pred.device)
        k = self.num_classes
        if pred.dim() != 4:
            raise ValueError("pred must be 4-d")
        if pred.shape[-1 if channels_last else 1] != k:
            raise ValueError(f"pred has {pred.shape[-1 if channels_last else 1]} classes on its {'last' if channels_last else 'second'} "
                             f"dimension, the meter has {k} (pass channels_last=True for [B,h,w,K] tensors)")
        nhwc = pred if channels_last else pred.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
        if nhwc.dtype not in (torch.float32, torch.bfloat16):
            nhwc = nhwc.float()
        b, h, w, _ = nhwc.shape
        t = target.to(torch.int64).contiguous()
        hh, ww = t.shape[1], t.shape[2]
        am = torch.empty((b, hh, ww), dtype=torch.uint8, device=nhwc.device) if return_argmax else None
        dev, st = dev_stream(nhwc)
        call("css_eval_confusion", nhwc, k, t, b, h, w, k, hh, ww, self.temp_mat, am, dtype_code(nhwc.dtype), dev, st)
        self.mat += self.temp_mat
        return am

    def __str__(self):
        self.avg = _iou_mean(self.mat)
        self.val = _iou_mean(self.temp_mat)
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)
