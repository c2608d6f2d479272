"""Student + EMA-teacher step wrappers with the reference's constructor / forward signatures
(generalframeworks/networks/ddp_model.py): ``Model_mix`` (:73-156), ``Model_cross`` (:158-239),
``Model_ori_pseudo`` (:8-70), ``concat_all_gather`` (:241-251).

Differences that are NOT visible through the interface:
  * all arithmetic runs in HIP kernels; student and teacher parameters live in two flat fp32 buffers so that
    ``ema_update`` is ONE kernel over 59.5 M elements instead of ~340 tiny ones (same formula, ddp_model.py:93-97);
  * the in-step augmentation is the GPU stand-in of ``css_amd.dataset_helpers.gpu_aug`` (see that module);
  * ``set_compute_dtype(torch.bfloat16)`` switches both networks to the bf16 MFMA path (default fp32 = parity path).
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from .._lib import call, dev_stream
from ..dataset_helpers.gpu_aug import (aug_mode, batch_transform, batch_transform_2, batch_transform_3, generate_cut_gather,
                                       generate_cut_gather_2, generate_cut_gather_3, prefetch_partner_image)
from .deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep


def flatten_parameters(module: nn.Module) -> torch.Tensor:
    """Re-home every parameter of ``module`` into one flat fp32 buffer (16-byte aligned slices, conv weights keep their
    channels_last physical layout) and return it.  Idempotent per device."""
    params = list(module.parameters())
    if not params:
        return None
    dev = params[0].device
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 7) // 8 * 8      # 16-byte aligned slices in the bf16 mirror too (ops.prepare_flat_weights)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    for p, o in zip(params, offs):
        n = p.numel()
        if p.dim() == 4:
            co, ci, r, s = p.shape
            view = flat[o:o + n].view(co, r, s, ci).permute(0, 3, 1, 2)
        else:
            view = flat[o:o + n].view(p.shape)
        with torch.no_grad():
            view.copy_(p.data)
        p.data = view
    module._css_flat = flat
    module._css_flat_offsets = offs
    return flat


class _StudentTeacher(nn.Module):
    def _init_common(self, base_encoder, num_classes, output_dim, ema_alpha, config):
        self.model = DeepLabv3Plus_with_rep(base_encoder, num_classes=num_classes, output_dim=output_dim, dilate_scale=8)
        self.num_classes = num_classes
        self.step = 0
        self.ema_model = copy.deepcopy(self.model)
        for p in self.ema_model.parameters():
            p.requires_grad = False
        self.alpha = ema_alpha
        print("EMA model has been prepared. Alpha = {}".format(self.alpha))
        self.config = config
        self._flat = None

    def _device_aug(self):
        """In-step augmentation mode.  The reference's YAML files have no such key: with the key ABSENT the drop-in path must train
        like the reference does, i.e. with its whole PIL pipeline (random rescale, crop, colour jitter, blur, flip, 8-bit
        quantisation of the confidence maps: VOC.py:325-352) - the device restatement 'pil'.  'identity' (geometry and colours
        untouched) is for parity traces and benchmarks, which ask for it explicitly."""
        return self.config["Dataset"].get("device_aug", "pil")

    def _prefetch_partner(self, u_image):
        """With the identity in-step augmentation the image that reaches the mixing IS the input image: rank 0's batch can be on its way to
        the other ranks while the teacher runs (gpu_aug.prefetch_partner_image); any other augmentation rescales / crops it first."""
        if self._device_aug() != "identity":
            return None
        return prefetch_partner_image(u_image, self.config["Dataset"]["mix_mode"])

    def set_compute_dtype(self, dtype):
        self.model.set_compute_dtype(dtype)
        self.ema_model.set_compute_dtype(dtype)
        return self

    def _ensure_flat(self):
        p0 = next(self.model.parameters())
        e0 = next(self.ema_model.parameters())
        # (re-)flatten when never done or when .to()/.cuda()/load replaced the parameter storages
        if self._flat is None or p0.data_ptr() != self._flat[0].data_ptr() or e0.data_ptr() != self._flat[1].data_ptr():
            self._flat = (flatten_parameters(self.model), flatten_parameters(self.ema_model))
        return self._flat

    def ema_update(self):
        """decay = min(1 - 1/(step+1), alpha); ema = decay*ema + (1-decay)*param over PARAMETERS only (ddp_model.py:93-97)."""
        decay = min(1 - 1 / (self.step + 1), self.alpha)
        if next(self.model.parameters()).is_cuda:
            s, t = self._ensure_flat()
            dev, st = dev_stream(s)
            call("css_ema", t, s, s.numel(), float(decay), dev, st)
            self.refresh_weights()
        else:
            raise RuntimeError("css_amd has no CPU path: move the model to the MI355X first")
        self.step += 1

    def refresh_weights(self):
        """After the parameters changed (optimizer / EMA step): rebuild every compute-dtype weight copy in bulk."""
        ops.invalidate_weight_cache()
        if self._flat is not None and self.model.compute_dtype != torch.float32:
            ops.prepare_flat_weights(self.model, self.model.compute_dtype, dgrad=True)
            ops.prepare_flat_weights(self.ema_model, self.ema_model.compute_dtype, dgrad=False)

    # ---- shared pieces of the three forwards ------------------------------------------------------------------
    # The reference calls each network twice per step (labeled batch, unlabeled batch: ddp_model.py:102-103,140-143).  Here
    # both batches go through the network in ONE pass, concatenated along dim 0, with two batch-norm statistics groups
    # (css_amd.ops.bn_groups): identical arithmetic, half the kernel launches, twice the rows per launch.
    @staticmethod
    def _check_pair(xl, xu):
        # the two passes are batched as ONE tensor with two statistics groups split at half its rows: that is the reference's two
        # separate calls only when both batches have the same shape
        if xl.shape != xu.shape:
            raise ValueError(f"labeled and unlabeled batches must have the same shape (got {tuple(xl.shape)} and {tuple(xu.shape)}): "
                             "the batched two-group pass splits the rows at the half")

    def _teacher_pair(self, xl, xu):
        self._check_pair(xl, xu)
        with ops.bn_groups(2):
            pred, rep = self.ema_model.forward_nhwc(self.ema_model.stage([xl, xu]))
        b = xl.shape[0]
        return pred[b:], rep[b:]            # the labeled half only moves the teacher's BN running statistics (ddp_model.py:102)

    def _teacher(self, x):
        return self.ema_model.forward_nhwc(self.ema_model.stage([x]))

    def _student_pair(self, xl, xu, out_hw, small=False):
        """-> pred [2B,h,w,K], rep_all [2B,h,w,C] (labeled first), pred_l_large, pred_u_large (logical NCHW, fp32).
        ``small=True`` (fused trainer): the last two are the two halves of the LOW-resolution NHWC logits instead - the losses then
        up-sample on the fly (loss._PixelCESmall) and the full-resolution logits are never materialised."""
        self._check_pair(xl, xu)
        with ops.bn_groups(2):
            pred, rep = self.model.forward_nhwc(self.model.stage([xl, xu]))
        if small:
            sl, su = ops.split2(pred, xl.shape[0])
            return pred, rep, sl, su
        large = ops.bilinear(pred, out_hw[0], out_hw[1], torch.float32)     # align_corners=True, ddp_model.py:141,144
        ll, lu = ops.split2(large, xl.shape[0])
        return pred, rep, ll.permute(0, 3, 1, 2), lu.permute(0, 3, 1, 2)


class Model_mix(_StudentTeacher):
    def __init__(self, base_encoder, num_classes=21, output_dim=256, ema_alpha=0.99, config=None, temp=0.25) -> None:
        super().__init__()
        self._init_common(base_encoder, num_classes, output_dim, ema_alpha, config)
        self.temp = temp

    def forward(self, train_l_image, train_u_image, prototypes, _want_prob=True, _small_logits=False):
        hw = train_u_image.shape[2:]
        with torch.no_grad(), aug_mode(self._device_aug()):
            cfg = self.config["Dataset"]
            pre = self._prefetch_partner(train_u_image)
            pred_u, rep_u = self._teacher_pair(train_l_image, train_u_image)
            sim, _, _ = Fn.similarity(rep_u, prototypes, self.temp, want_sim=True)
            logits_rep, labels_rep, logits_cls, labels_cls, pseudo = Fn.pseudo_labels(sim, pred_u, self.temp, hw)
            u_img, u_lab, u_lc, u_lr = batch_transform_2(train_u_image, pseudo, logits_cls, logits_rep, crop_size=cfg["crop_size"],
                                                         scale_size=cfg["scale_size"], augmentation=False)
            u_img, u_lab, u_lc, u_lr = generate_cut_gather_2(u_img, u_lab, u_lc, u_lr, mode=cfg["mix_mode"], prefetched=pre)
            u_img, u_lab, u_lc, u_lr = batch_transform_2(u_img, u_lab, u_lc, u_lr, crop_size=cfg["crop_size"], scale_size=(1.0, 1.0),
                                                         augmentation=True)
        _, rep_all, pred_l_large, pred_u_large = self._student_pair(train_l_image, u_img, train_l_image.shape[2:], small=_small_logits)
        prob_all = None
        if _want_prob:                                                     # the trainer derives the hard flags directly instead
            with torch.no_grad():
                _, prob_all, _ = Fn.similarity(rep_all, prototypes, self.temp, want_prob=True)
            prob_all = prob_all.permute(0, 3, 1, 2)
        return (pred_l_large, pred_u_large, u_lab, u_lc, u_lr, rep_all.permute(0, 3, 1, 2), prob_all)


class Model_cross(_StudentTeacher):
    def __init__(self, base_encoder, num_classes=21, output_dim=256, ema_alpha=0.99, config=None, temp=0.1) -> None:
        super().__init__()
        self._init_common(base_encoder, num_classes, output_dim, ema_alpha, config)
        self.temp = temp

    def forward(self, train_l_image, train_u_image, prototypes, _want_prob=True, _small_logits=False):
        hw = train_u_image.shape[2:]
        with torch.no_grad(), aug_mode(self._device_aug()):
            cfg = self.config["Dataset"]
            pre = self._prefetch_partner(train_u_image)
            pred_u, rep_u = self._teacher_pair(train_l_image, train_u_image)
            sim, _, _ = Fn.similarity(rep_u, prototypes, self.temp, want_sim=True)
            logits_rep, labels_rep, logits_cls, labels_cls, _ = Fn.pseudo_labels(sim, pred_u, self.temp, hw)
            a = batch_transform_3(train_u_image, labels_cls, labels_rep, logits_cls, logits_rep, crop_size=cfg["crop_size"],
                                  scale_size=cfg["scale_size"], augmentation=False)
            a = generate_cut_gather_3(*a, mode=cfg["mix_mode"], prefetched=pre)
            u_img, u_lab_c, u_lab_r, u_lc, u_lr = batch_transform_3(*a, crop_size=cfg["crop_size"], scale_size=(1.0, 1.0),
                                                                    augmentation=True)
        _, rep_all, pred_l_large, pred_u_large = self._student_pair(train_l_image, u_img, train_l_image.shape[2:], small=_small_logits)
        prob_all = None
        if _want_prob:                                                     # the fused trainer derives the hard flags directly instead
            with torch.no_grad():
                _, prob_all, _ = Fn.similarity(rep_all, prototypes, self.temp, want_prob=True)
            prob_all = prob_all.permute(0, 3, 1, 2)
        return (pred_l_large, pred_u_large, u_lab_c, u_lab_r, u_lc, u_lr, rep_all.permute(0, 3, 1, 2), prob_all)


class Model_ori_pseudo(_StudentTeacher):
    def __init__(self, base_encoder, num_classes=21, output_dim=256, ema_alpha=0.99, config=None) -> None:
        super().__init__()
        self._init_common(base_encoder, num_classes, output_dim, ema_alpha, config)

    def forward(self, train_l_image, train_u_image, _small_logits=False):
        hw = train_u_image.shape[2:]
        with torch.no_grad(), aug_mode(self._device_aug()):
            pre = self._prefetch_partner(train_u_image)
            pred_u, _ = self._teacher(train_u_image)
            raw = None if _small_logits else ops.bilinear(pred_u, hw[0], hw[1], torch.float32)   # (7th output: unused by the train body)
            # softmax + max in class space only: the pseudo-label kernel with a constant similarity map
            zero_sim = torch.zeros((*pred_u.shape[:3], self.num_classes), dtype=torch.float32, device=pred_u.device)
            _, _, logits, labels, _ = Fn.pseudo_labels(zero_sim, pred_u, 1.0, hw)
            cfg = self.config["Dataset"]
            u_img, u_lab, u_lg = batch_transform(train_u_image, labels, logits, crop_size=cfg["crop_size"], scale_size=cfg["scale_size"],
                                                 augmentation=False)
            u_img, u_lab, u_lg = generate_cut_gather(u_img, u_lab, u_lg, mode=cfg["mix_mode"], prefetched=pre)
            u_img, u_lab, u_lg = batch_transform(u_img, u_lab, u_lg, crop_size=cfg["crop_size"], scale_size=(1.0, 1.0), augmentation=True)
        pred_all, rep_all, pred_l_large, pred_u_large = self._student_pair(train_l_image, u_img, train_l_image.shape[2:], small=_small_logits)
        return (pred_l_large, pred_u_large, u_lab, u_lg, rep_all.permute(0, 3, 1, 2), pred_all.permute(0, 3, 1, 2),
                None if raw is None else raw.permute(0, 3, 1, 2))


@torch.no_grad()
def concat_all_gather(tensor):
    """torch.distributed.all_gather + cat (ddp_model.py:241-251).  Kept for API compatibility; the HIP losses never need it
    (prototype statistics are all-reduced as K*(C+1) numbers instead)."""
    tensors_gather = [torch.ones_like(tensor) for _ in range(torch.distributed.get_world_size())]
    torch.distributed.all_gather(tensors_gather, tensor, async_op=False)
    return torch.cat(tensors_gather, dim=0)
