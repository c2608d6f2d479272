"""mean_intersection_over_union (generalframeworks/util/miou.py:3-9): a handful of flops on a K x K matrix, plain torch."""
import torch


def mean_intersection_over_union(mat: torch.Tensor):
    h = mat.float()
    iu = torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))
    return torch.mean(iu).item()
