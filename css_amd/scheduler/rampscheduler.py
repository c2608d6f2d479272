"""Epoch-wise ramp-down of the contrastive-loss weight (generalframeworks/scheduler/rampscheduler.py:27-54)."""
import math


class RampdownScheduler(object):
    def __init__(self, begin_epoch, max_epoch, current_epoch, max_value, min_value, ramp_mult):
        self.begin_epoch, self.max_epoch = int(begin_epoch), int(max_epoch)
        self.max_value, self.mult = float(max_value), float(ramp_mult)
        self.epoch, self.min_value = current_epoch, min_value

    def step(self):
        self.epoch += 1

    @property
    def value(self):
        return max(self.get_lr(self.epoch, self.begin_epoch, self.max_epoch, self.max_value, self.min_value, self.mult), self.min_value)

    @staticmethod
    def get_lr(epoch, begin_epoch, max_epochs, max_val, min_value, mult):
        if epoch < begin_epoch:
            return 0.0
        if epoch >= max_epochs:
            return min_value
        return max_val * math.exp(mult * (float(epoch - begin_epoch) / (max_epochs - begin_epoch)) ** 2)
