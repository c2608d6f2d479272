import torch


def dev():
    return torch.device("cuda:0")


def to_nhwc(x_nchw, dtype, pad_to=None):
    """CPU NCHW fp32 -> GPU NHWC ``dtype`` (optionally zero-padded channels)."""
    x = x_nchw.permute(0, 2, 3, 1).contiguous()
    if pad_to is not None and pad_to != x.shape[-1]:
        xp = torch.zeros(*x.shape[:3], pad_to)
        xp[..., : x.shape[-1]] = x
        x = xp
    return x.to(dev()).to(dtype).contiguous()


def to_nchw_cpu(x_nhwc):
    return x_nhwc.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def bf16_round(x):
    return x.to(torch.bfloat16).float()


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def robust_err(a, b, q=0.9):
    """(q-quantile, max) of |a-b| / max|b|.  Block-level backward checks use the quantile: a ReLU whose pre-activation
    sits within fp32 rounding of 0 can get a different mask in the two implementations (probability ~1e-7 x #elements
    per layer), which perturbs a few per cent of the gradient elements by O(1e-2) -- a wiring bug perturbs far more."""
    a, b = a.double().flatten(), b.double().flatten()
    e = (a - b).abs() / (b.abs().max() + 1e-30)
    if e.numel() > 2_000_000:
        e = e[:: e.numel() // 2_000_000 + 1]
    return torch.quantile(e, q).item(), e.max().item()


def assert_close_robust(a, b, tol, what=""):
    q, m = robust_err(a, b)
    assert q < tol and m < 0.3, (what, q, m)
