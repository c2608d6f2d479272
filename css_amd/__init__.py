"""css_amd: MI355X-native (gfx950) implementation of the CSS data-parallel hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, autograd bookkeeping, torch.distributed).
Compute: hand-written HIP kernels in ``css_amd/csrc`` behind the C ABI ``include/css_hip.h``.
There is no CPU or eager-PyTorch fallback: without ``libcss_hip.so`` every op raises.
"""
__version__ = "0.1.0"
