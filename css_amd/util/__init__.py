"""Mirror of generalframeworks/util (the evaluation helpers of mix_label.py:199-225) on the HIP path."""
from .meter import AverageMeter, ConfMatrix  # noqa: F401
from .miou import mean_intersection_over_union  # noqa: F401
from .torch_dist_sum import torch_dist_sum  # noqa: F401
