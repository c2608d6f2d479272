"""``test()`` of the reference (mix_label.py:199-225) on the HIP path: EMA model in eval mode (running-statistics BN),
fused up-sample + argmax + confusion matrix per batch, K^2 int64 all-reduce, mean IoU."""
from __future__ import annotations

import torch

from .util import ConfMatrix, mean_intersection_over_union, torch_dist_sum


@torch.no_grad()
def test(test_loader, model, config):
    k = config["Network"]["num_class"]
    meter = ConfMatrix(num_classes=k, fmt=":6.4f", name="test_miou")
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    for test_image, test_label in test_loader:
        test_image, test_label = test_image.to(dev), test_label.to(dev)
        pred, _ = model(test_image)
        meter.update_from_logits(pred, test_label)
    if meter.mat is None:
        raise ValueError("empty test loader")
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    mat = torch_dist_sum(rank, meter.mat)
    if was_training:
        model.train()
    return mean_intersection_over_union(mat[0])
