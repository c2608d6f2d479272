"""Drop-in aliasing: make ``import generalframeworks.networks...`` / ``generalframeworks.loss...`` resolve to css_amd.

    import css_amd.compat; css_amd.compat.install()

after which the reference's entry scripts' imports

    from generalframeworks.networks.ddp_model import Model_mix            (mix_label.py:18)
    from generalframeworks.loss.loss import Attention_Threshold_Loss, Contrast_Loss, ProbOhemCrossEntropy2d
    from generalframeworks.networks import resnet
    from generalframeworks.utils import label_onehot, label_onehot_2
    from generalframeworks.scheduler.my_lr_scheduler import PolyLR
    from generalframeworks.scheduler.rampscheduler import RampdownScheduler

pick up the HIP-backed implementations.  Modules of the reference that are out of scope here (augmentation, the rest of
util) are left alone: if the real ``generalframeworks`` package is importable it keeps serving those.
"""
import importlib
import sys
import types

_MAP = {
    "generalframeworks.networks.ddp_model": "css_amd.networks.ddp_model",
    "generalframeworks.networks.resnet": "css_amd.networks.resnet",
    "generalframeworks.networks.deeplabv3.deeplabv3": "css_amd.networks.deeplabv3.deeplabv3",
    "generalframeworks.networks.deeplabv3.aspp": "css_amd.networks.deeplabv3.aspp",
    "generalframeworks.loss.loss": "css_amd.loss.loss",
    "generalframeworks.scheduler.my_lr_scheduler": "css_amd.scheduler.my_lr_scheduler",
    "generalframeworks.scheduler.rampscheduler": "css_amd.scheduler.rampscheduler",
    # evaluation helpers of test() (mix_label.py:199-225)
    "generalframeworks.util.meter": "css_amd.util.meter",
    "generalframeworks.util.miou": "css_amd.util.miou",
    "generalframeworks.util.torch_dist_sum": "css_amd.util.torch_dist_sum",
    # data path (SURVEY 8f-4 loaders on CPU workers; 8f-1 in-step augmentation on the device)
    "generalframeworks.dataset_helpers.VOC": "css_amd.dataset_helpers.VOC",
    "generalframeworks.dataset_helpers.Cityscapes": "css_amd.dataset_helpers.Cityscapes",
}


def _ensure_pkg(name):
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m


def install():
    for alias, real in _MAP.items():
        parts = alias.split(".")
        for i in range(1, len(parts)):
            _ensure_pkg(".".join(parts[:i]))
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
    # generalframeworks.utils: only the two one-hot helpers are on the hot path
    try:
        u = importlib.import_module("generalframeworks.utils")
    except Exception:
        u = types.ModuleType("generalframeworks.utils")
        sys.modules["generalframeworks.utils"] = u
        setattr(_ensure_pkg("generalframeworks"), "utils", u)
    from . import utils as cu
    u.label_onehot, u.label_onehot_2 = cu.label_onehot, cu.label_onehot_2
