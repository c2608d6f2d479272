"""mean_intersection_over_union (generalframeworks/util/miou.py:3-9): a handful of flops on a K x K matrix, plain torch."""
import torch


def mean_intersection_over_union(mat: torch.Tensor):
    """Mean over classes of TP / (row sum + column sum - TP) of a K x K confusion matrix (rows = ground truth), as a Python float.
    A class that never occurs and is never predicted contributes 0/0 = NaN, like the reference's expression."""
    counts = mat.to(torch.float32)
    hits = counts.diagonal()
    union = counts.sum(dim=1) + counts.sum(dim=0) - hits
    return float((hits / union).mean())
