"""Backward on a partitioned chip (css_amd/partition.py): batch norm + data gradients on a CU-masked main stream, the
weight gradients on the side stream's CUs, joined before the optimizer.  The partition must change NOTHING but the time: same kernels on
smaller grids, the same order of every sum - five training steps (fp32 and bf16: losses, all 59.5 M weights, momentum, the EMA teacher,
prototypes) bit-identical with and without it; the persistent kernels size their grids by the partition's CUs (css_stream_cu_count).
Reference: /root/reference/mix_label.py:193 (``total_loss.backward()``: the chain that is re-scheduled).
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import dev  # noqa: E402
from test_determinism_gpu import _run, _same_bits  # noqa: E402


def _with_partition(n, fn):
    prev = os.environ.get("CSS_BWD_PARTITION")
    os.environ["CSS_BWD_PARTITION"] = str(n)
    try:
        return fn()
    finally:
        if prev is None:
            os.environ.pop("CSS_BWD_PARTITION", None)
        else:
            os.environ["CSS_BWD_PARTITION"] = prev


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_partitioned_backward_is_bit_identical(dtype):
    from css_amd import partition
    a = _with_partition(0, lambda: _run(dtype, 5))
    b = _with_partition(192, lambda: _run(dtype, 5))
    # (the first window walked the graph: no torch-native kernel node between two nodes of this package -> the light mode)
    assert _with_partition(192, lambda: partition.get(dev())).strict is False
    os.environ["CSS_BWD_PARTITION_STRICT"] = "1"        # another split, every node synchronised both ways
    try:
        c = _with_partition(160, lambda: _run(dtype, 2))
        assert _with_partition(160, lambda: partition.get(dev())).strict is True
    finally:
        os.environ.pop("CSS_BWD_PARTITION_STRICT", None)
    for i in range(a["hist"].shape[0]):
        assert _same_bits(a["hist"][i], b["hist"][i]), (i, a["hist"][i].tolist(), b["hist"][i].tolist())
    for key in ("p", "m", "ema", "proto"):
        assert _same_bits(a[key], b[key]), (key, int((a[key] != b[key]).sum()), float((a[key] - b[key]).abs().max()))
    for i in range(2):
        assert _same_bits(a["hist"][i], c["hist"][i]), ("160 CUs", i)


def test_masked_streams_carry_their_cu_count():
    import ctypes
    from css_amd import partition
    from css_amd._lib import query
    part = _with_partition(192, lambda: partition.get(dev()))
    total = query("css_device_cu_count", 0)
    assert part is not None and part.main_cus == 192 and part.side_cus == total - 192
    assert query("css_stream_cu_count", 0, ctypes.c_void_p(part.main.cuda_stream)) == 192
    assert query("css_stream_cu_count", 0, ctypes.c_void_p(part.side.cuda_stream)) == total - 192
    assert query("css_stream_cu_count", 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == total
    assert _with_partition(0, lambda: partition.get(dev())) is None
    # work queued on the masked streams completes and is ordered by ordinary events
    x = torch.zeros(1 << 20, device=dev())
    with torch.cuda.stream(part.main):
        x += 1
    part.side.wait_stream(part.main)
    with torch.cuda.stream(part.side):
        x *= 3
    torch.cuda.current_stream().wait_stream(part.side)
    assert float(x.sum()) == 3.0 * (1 << 20)
