"""Checkpoint wire format of the reference (mix_label.py:104-112 resume, :137-147 save).

    {'epoch': int, 'model': student state_dict, 'ema_model': teacher state_dict, 'optimizer': torch.optim.SGD.state_dict(),
     'lr_scheduler': PolyLR.state_dict(), 'prototypes': numpy [K, C]}

The HIP modules keep the reference's parameter / buffer names and logical NCHW shapes (memory is channels_last, which
``state_dict`` / ``load_state_dict`` do not see), so the two model entries need no translation.  The fused trainer keeps
SGD's momentum in one flat fp32 buffer: ``optimizer_state_dict`` / ``load_optimizer_state_dict`` convert between that buffer
and the per-parameter ``momentum_buffer`` tensors of ``torch.optim.SGD`` (parameter order = ``model.parameters()`` =
the single param group the reference builds at mix_label.py:96-97)."""
from __future__ import annotations

import numpy as np
import torch

from .scheduler.my_lr_scheduler import poly_lr


def optimizer_state_dict(trainer) -> dict:
    """torch.optim.SGD(model.module.model.parameters(), lr, weight_decay, momentum, nesterov=True).state_dict() of a MixTrainer."""
    params = list(trainer.model.model.parameters())
    state = {}
    if trainer.it > 0:                      # SGD creates momentum buffers on the first step
        for i, (p, o) in enumerate(zip(params, trainer.model.model._css_flat_offsets)):
            n = p.numel()
            buf = trainer.flat_m[o:o + n]
            if p.dim() == 4:
                co, ci, r, s = p.shape
                buf = buf.view(co, r, s, ci).permute(0, 3, 1, 2)
            else:
                buf = buf.view(p.shape)
            state[i] = {"momentum_buffer": buf.detach().clone().contiguous()}
    group = {"lr": float(trainer.lr), "momentum": trainer.momentum, "dampening": 0, "weight_decay": trainer.wd, "nesterov": True,
             "maximize": False, "foreach": None, "differentiable": False, "fused": None, "initial_lr": trainer.base_lr,
             "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd: dict) -> None:
    params = list(trainer.model.model.parameters())
    groups = sd["param_groups"]
    if len(groups) != 1 or len(groups[0]["params"]) != len(params):
        raise ValueError("expected the reference's single SGD param group over all student parameters")
    g = groups[0]
    if not g.get("nesterov", False):
        raise ValueError("the fused trainer implements SGD with nesterov momentum (mix_label.py:96-97)")
    trainer.momentum, trainer.wd = float(g["momentum"]), float(g["weight_decay"])
    trainer.base_lr = float(g.get("initial_lr", trainer.base_lr))
    trainer.flat_m.zero_()
    for i, (p, o) in enumerate(zip(params, trainer.model.model._css_flat_offsets)):
        st = sd["state"].get(i, sd["state"].get(str(i)))
        if st is None or st.get("momentum_buffer") is None:
            continue
        buf = st["momentum_buffer"].to(trainer.flat_m.device, torch.float32)
        if tuple(buf.shape) != tuple(p.shape):
            raise ValueError(f"momentum buffer {i}: shape {tuple(buf.shape)} != parameter {tuple(p.shape)}")
        n = p.numel()
        dst = trainer.flat_m[o:o + n]
        if p.dim() == 4:
            co, ci, r, s = p.shape
            dst.view(co, r, s, ci).copy_(buf.permute(0, 2, 3, 1))
        else:
            dst.view(p.shape).copy_(buf)


def lr_scheduler_state_dict(trainer) -> dict:
    """PolyLR(optimizer, total_iter, min_lr=1e-4).state_dict() after ``trainer.it`` scheduler steps (mix_label.py:102,196)."""
    return {"power": 0.9, "max_iters": trainer.total_iter, "min_lr": trainer.min_lr, "base_lrs": [trainer.base_lr], "last_epoch": trainer.it,
            "verbose": False, "_step_count": trainer.it + 1, "_get_lr_called_within_step": False,
            "_last_lr": [poly_lr(trainer.base_lr, trainer.it, trainer.total_iter, 0.9, trainer.min_lr)]}


def load_lr_scheduler_state_dict(trainer, sd: dict) -> None:
    trainer.it = int(sd["last_epoch"])
    trainer.total_iter = int(sd.get("max_iters", trainer.total_iter))
    trainer.min_lr = float(sd.get("min_lr", trainer.min_lr))
    if sd.get("base_lrs"):
        trainer.base_lr = float(sd["base_lrs"][0])


def state_for_save(trainer, epoch: int, exact_resume: bool = False) -> dict:
    """The dict the reference passes to torch.save (mix_label.py:139-146).  ``exact_resume=True`` adds ONE key the reference's resume code
    never reads (it picks its six keys by name, mix_label.py:107-112): 'css_amd_resume' = the two counters the reference's format loses -
    the EMA warm-up counter ``Model_mix.step`` (ddp_model.py:93-97) and the position of the contrastive sampler's stream - so that
    ``load_checkpoint(..., exact_resume=True)`` continues a run bit for bit (tests/test_mini_training_gpu.py)."""
    trainer.finish()                         # a step whose gradients were invalid raises here instead of being saved
    m = trainer.model
    sd = {"epoch": epoch + 1, "model": m.model.state_dict(), "ema_model": m.ema_model.state_dict(),
          "optimizer": optimizer_state_dict(trainer), "lr_scheduler": lr_scheduler_state_dict(trainer),
          "prototypes": trainer.prototypes.data.cpu().numpy()}
    if exact_resume:
        sd["css_amd_resume"] = {"ema_step": int(m.step), "sampler_calls": int(trainer.crit_contrast._calls)}
    return sd


def save_checkpoint(path: str, trainer, epoch: int, exact_resume: bool = False) -> None:
    torch.save(state_for_save(trainer, epoch, exact_resume), path)


def load_checkpoint(path_or_dict, trainer, exact_resume: bool = False) -> int:
    """Resume like mix_label.py:104-112; returns start_epoch.  Accepts the reference's files (keys with or without the
    DistributedDataParallel 'module.' prefix).  ``exact_resume=True``: also restore the two counters of 'css_amd_resume' when the file has
    them (default False = the reference's behaviour, below)."""
    ck = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, str) else path_or_dict

    def strip(sd):
        return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    m = trainer.model
    m.model.load_state_dict(strip(ck["model"]))
    m.ema_model.load_state_dict(strip(ck["ema_model"]))
    load_optimizer_state_dict(trainer, ck["optimizer"])
    load_lr_scheduler_state_dict(trainer, ck["lr_scheduler"])
    # like the reference, Model_mix.step (the EMA warm-up counter of ddp_model.py:93-97) is NOT part of the checkpoint: a
    # resumed run starts it at 0 again, so its first ema_update copies the student into the teacher (decay = 0)
    proto = torch.as_tensor(np.asarray(ck["prototypes"]), dtype=torch.float32)
    trainer.prototypes.copy_(proto.to(trainer.prototypes.device))
    extra = ck.get("css_amd_resume") if exact_resume else None
    if extra is not None:
        m.step = int(extra["ema_step"])
        trainer.crit_contrast._calls = int(extra["sampler_calls"])
    m.refresh_weights()                      # parameters changed behind the weight cache
    return int(ck["epoch"])
