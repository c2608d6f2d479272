"""One-hot encoders with the reference's signatures (generalframeworks/utils.py:116-136).

They exist for callers that follow mix_label.py:175-183 verbatim; ``css_amd.train_step.MixTrainer`` does not build
the [B,K,H,W] one-hot tensors at all (it folds label_all / mask_all into a class-id map on the device,
``css_amd.functional.class_map``).  Pure index bookkeeping via torch scatter on whatever device ``inputs`` lives on.
"""
import torch


def label_onehot(inputs, num_class):
    """relu maps -1 -> class 0, then scatter (utils.py:116-125)."""
    b, h, w = inputs.shape
    inputs = torch.relu(inputs)
    out = torch.zeros([b, num_class, h, w], device=inputs.device)
    return out.scatter_(1, inputs.unsqueeze(1), 1.0)


def label_onehot_2(inputs, num_class):
    """+1 shift, K+1 channels; the caller drops channel 0 (utils.py:127-136, mix_label.py:182)."""
    b, h, w = inputs.shape
    inputs = inputs + 1
    out = torch.zeros([b, num_class + 1, h, w], device=inputs.device)
    return out.scatter_(1, inputs.unsqueeze(1), 1.0)
