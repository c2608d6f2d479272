"""PolyLR with the reference's formula (generalframeworks/scheduler/my_lr_scheduler.py:4-13)."""
from torch.optim.lr_scheduler import _LRScheduler


def poly_lr(base_lr, it, max_iters, power=0.9, min_lr=1e-6):
    return max(base_lr * (1 - it / max_iters) ** power, min_lr)


class PolyLR(_LRScheduler):
    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power, self.max_iters, self.min_lr = power, max_iters, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [poly_lr(b, self.last_epoch, self.max_iters, self.power, self.min_lr) for b in self.base_lrs]
