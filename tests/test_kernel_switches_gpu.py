"""Every environment switch the launchers read (css_amd/csrc/conv.hip, conv_wgrad.hip, conv_pp.hip, conv_p8.hip: kernel selection for A/B timing)
is a shipped configuration: each one runs a bench-shape convolution (32 images of 65x65, 256 -> 256, 3x3 dilation 2: whole rounds of
the chip + leftover rows + the fused statistics) forward, data gradient and weight gradient against torch-CPU fp32, in a process of its
own (the switches are read once).  Shapes as in tests/test_conv_bench_scale_gpu.py; reference: generalframeworks/networks/resnet.py:119-139."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import torch, torch.nn.functional as F
from css_amd import ops
from gpu_util import bf16_round, dev, rel_err
n, h, w, cin, cout, k, pad, dil = 32, 65, 65, 256, 256, 3, 2, 2
g = torch.Generator().manual_seed(77)
x = bf16_round(torch.randn(n, h, w, cin, generator=g) + 0.25)
wt = bf16_round(torch.randn(cout, k, k, cin, generator=g) / (cin * k * k) ** 0.5)
gy = bf16_round(torch.randn(n, h, w, cout, generator=g))
xr = x.permute(0, 3, 1, 2).requires_grad_(True)
wr = wt.permute(0, 3, 1, 2).requires_grad_(True)
yr = F.conv2d(xr, wr, None, 1, pad, dil)
yr.backward(gy.permute(0, 3, 1, 2))
xg = x.to(dev(), torch.bfloat16).requires_grad_(True)
wg = wt.permute(0, 3, 1, 2).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
with ops.bn_groups(2):
    y = ops.conv2d(xg, wg, None, 1, pad, dil, bn_stats=True)
y.backward(gy.to(dev(), torch.bfloat16))
torch.cuda.synchronize()
e = (rel_err(y.detach().float().cpu(), yr.detach().permute(0, 2, 3, 1)), rel_err(xg.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1)),
     rel_err(wg.grad.cpu(), wr.grad))
# fused statistics (when this configuration emits them) against an fp64 reduction of the CPU output
if hasattr(y, "_css_bnstats"):
    from css_amd._lib import call, dev_stream
    part, mg, groups, c_, bm = y._css_bnstats
    sums = torch.empty(groups * 2 * cout + groups, dtype=torch.float64, device=dev())
    d, st = dev_stream(y)
    call("css_bn_reduce_finalize_slabs", part, mg * groups, mg, groups, float(mg), None, None, None, None, 0.0, 0.0, None, None, None, None,
         sums, cout, y, cout, bm, d, st)
    yy = yr.detach().permute(0, 2, 3, 1).double().reshape(groups, -1, cout)
    got = sums.cpu()[:groups * 2 * cout].reshape(groups, 2, cout)
    es = rel_err(got[:, 1], (yy * yy).sum(1))
else:
    es = 0.0
print("ERRS", *e, es)
assert max(e) < 2e-2 and es < 2e-3, (e, es)
'''

SWITCHES = [{}, {"CSS_NO_P8_CONV": "1"}, {"CSS_NO_SMALL_SPLITK": "1"}, {"CSS_NO_P8_CONV": "1", "CSS_NO_PP_CONV": "1"},
            {"CSS_PP_KORDER": "0"}, {"CSS_NO_DMA256_CONV": "1"}, {"CSS_NO_DMA_CONV": "1"}, {"CSS_WGRAD_ATOMICS": "1"},
            {"CSS_NO_DMA256_WGRAD": "1"}, {"CSS_REM_N64": "1"}, {"CSS_BN_RED_BLOCKS": "256"}, {"CSS_SMALL_NST2": "0", "CSS_SMALL64_NST2": "0"}, {"CSS_N128_SMALL_ONLY": "0"},
            # round 6 (conv_wgrad.hip): the 16x16x32 form of the 256x256 weight-gradient kernel, live-row compaction off, slice-major dealing
            {"CSS_WGRAD_MFMA": "16"}, {"CSS_WGRAD_NO_COMPACT": "1"}, {"CSS_WGRAD_NO_LONGEST_FIRST": "1"},
            # round 6: cache policies (non-temporal slab stores + loads on / everything plain)
            {"CSS_WGRAD_NT": "3"}, {"CSS_WGRAD_NT": "0", "CSS_CONV_NT": "2"}]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_conv_parity_under_every_launcher_switch(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", WORKER % (ROOT, os.path.join(ROOT, "tests"))], env=e, capture_output=True, text=True, timeout=600)
    print(r.stdout[-400:], r.stderr[-800:])
    assert r.returncode == 0, (env, r.stderr[-800:])


def test_bn_activation_mask_switch_is_a_shipped_configuration():
    """CSS_BN_NO_MASK=1 (residual layers re-read their activation for the ReLU mask in backward, css_amd/ops.py): the batch-norm parity tests
    against F.batch_norm under that switch, in a process of its own."""
    e = dict(os.environ)
    e["CSS_BN_NO_MASK"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ops_gpu.py"), "-q", "-x", "-k", "bn_act_train", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-600:], r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-800:]


@pytest.mark.parametrize("env", [{"CSS_BN_PASS_ORDER": "1"}, {"CSS_BN_PASS_ORDER": "2"}, {"CSS_BN_PASS_ORDER": "4"}, {"CSS_BN_PASS_ORDER": "7"}, {"CSS_BN_PASS_ORDER": "0"},
                                 # round 6: the non-temporal load / store switches of the three streaming batch-norm kernels, all off and all on
                                 {"CSS_BN_NT": "0", "CSS_BN_NT_BWDR": "0", "CSS_BN_NT_BWDA": "0"}, {"CSS_BN_NT": "7", "CSS_BN_NT_BWDR": "7", "CSS_BN_NT_BWDA": "7"},
                                 {"CSS_BN_NT": "15", "CSS_BN_NT_BWDA": "15"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_bn_pass_order_switch_is_a_shipped_configuration(env):
    """CSS_BN_PASS_ORDER (css_amd/csrc/bn.hip: which of the three streaming batch-norm passes walk the rows downwards) and CSS_BN_NT* (their cache
    policies): same results."""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ops_gpu.py"), "-q", "-x", "-k", "bn_act_train", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-600:], r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-800:]


@pytest.mark.parametrize("env", [{"CSS_NO_P8_CONV": "1"}, {"CSS_NO_WS_CONV": "1"}, {"CSS_NO_DMA256_CONV": "1"}, {"CSS_NO_DMA_CONV": "1"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_masked_residual_gradient_under_every_conv_fallback(env):
    """css_conv2d_dgrad_add_masked on each kernel family the switches route it to (conv_pp64 / conv_pp / the 256x256 and 128x128 LDS-DMA
    kernels / the register-staged kernel): tests/test_dgrad_add_masked_gpu.py in a process of its own."""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_dgrad_add_masked_gpu.py"), "-q", "-x", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-600:], r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-800:]


def test_ce_gather_kernel_switch_is_a_shipped_configuration():
    """CSS_CE_NO_TILE=1 (css_amd/csrc/losses.hip: the cross-entropy from low-resolution logits on the gather kernel instead of the tiled
    one): the fused-loss parity tests under that switch, in a process of its own."""
    e = dict(os.environ)
    e["CSS_CE_NO_TILE"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_losses_gpu.py"), "-q", "-x", "-k", "low_resolution", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-600:], r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-800:]


PSEUDO_WORKER = r'''
import sys
sys.path.insert(0, %r)
import torch
from css_amd import functional as Fn
torch.manual_seed(2)
dev = torch.device("cuda:0")
out = []
for (B, h, H, K, dt) in ((4, 129, 513, 21, torch.bfloat16), (2, 97, 385, 19, torch.float32), (3, 17, 65, 21, torch.bfloat16), (1, 33, 70, 5, torch.float32)):
    sim = (torch.rand(B, h, h, K, device=dev) * 2 - 1).contiguous()
    pred = (torch.randn(B, h, h, K, device=dev) * 3).to(dt).contiguous()
    out.append([t.cpu() for t in Fn.pseudo_labels(sim, pred, 0.5, (H, H))])
torch.save(out, sys.argv[1])
'''


def test_pseudo_label_tile_kernel_equals_the_gather_kernel(tmp_path):
    """css_pseudo_label on 32 x 32 tiles with the footprint staged in LDS (round 4) against the per-pixel gather kernel it replaces
    (CSS_PSEUDO_NO_TILE=1, a process of its own: the switch is read once): confidences, arg-max maps and the agreement map bit for bit, at the
    c2 / c4 launch geometries, a small one and a non-integer factor (ddp_model.py:111-118)."""
    import torch
    res = []
    for sw in ("0", "1"):
        e = dict(os.environ)
        if sw == "1":
            e["CSS_PSEUDO_NO_TILE"] = "1"
        out = str(tmp_path / f"p{sw}.pt")
        r = subprocess.run([sys.executable, "-c", PSEUDO_WORKER % ROOT, out], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        res.append(torch.load(out))
    for a, b in zip(*res):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and torch.equal(x, y), (x.shape, float((x.double() - y.double()).abs().max()))


def test_eager_residual_gradient_switch_is_a_shipped_configuration():
    """CSS_BN_EAGER_DRES=1 (css_amd/ops.py: bn_bwd_apply writes the masked residual gradient itself instead of leaving the mask to the tapped
    convolution's dgrad store): the block-level parity tests under that switch, in a process of its own."""
    e = dict(os.environ)
    e["CSS_BN_EAGER_DRES"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_blocks_gpu.py"), "-q", "-x", "-m", "gpu"],
                       env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-600:], r.stderr[-400:])
    assert r.returncode == 0, r.stdout[-800:]


@pytest.mark.parametrize("env,select", [({"CSS_MAXPOOL_BWD_GENERIC": "1"}, "maxpool"), ({"CSS_NO_BN_POOL": "1"}, "blocks"), ({"CSS_WGRAD_N64": "1"}, "stemconv"),
                                        ({"CSS_NO_C64_CONV": "1"}, "layer1")],
                         ids=["CSS_MAXPOOL_BWD_GENERIC=1", "CSS_NO_BN_POOL=1", "CSS_WGRAD_N64=1", "CSS_NO_C64_CONV=1"])
def test_round5_stem_region_switches_are_shipped_configurations(env, select):
    """Round 5: CSS_MAXPOOL_BWD_GENERIC=1 (the generic max-pool adjoint instead of the 3x3 s2 p1 form: the pooling tests against F.max_pool2d) and
    CSS_NO_BN_POOL=1 (the stem's max pool as a pass of its own behind bn_apply: the stem block tests against the reference's golden vectors), each
    in a process of its own.  (CSS_NO_STEM_S2D=1 has its own two-process test in tests/test_conv_stem_gpu.py.)"""
    e = dict(os.environ)
    e.update(env)
    # (CSS_WGRAD_N64=1: the Cout <= 64 weight gradients on 64 x 64 tiles - the stem's weight gradient against torch-CPU)
    target = {"maxpool": [os.path.join(ROOT, "tests", "test_ops_gpu.py"), "-k", "maxpool"],
              "blocks": [os.path.join(ROOT, "tests", "test_blocks_gpu.py"), "-k", "stem"],
              "stemconv": [os.path.join(ROOT, "tests", "test_conv_stem_gpu.py"), "-k", "forward_stats_and_wgrad"],
              # (CSS_NO_C64_CONV=1: layer 1's 3x3 convolutions and the deep stem back on the implicit-GEMM kernels - the Bottleneck and stem blocks
              # against the reference's golden vectors)
              "layer1": [os.path.join(ROOT, "tests", "test_blocks_gpu.py")]}[select]
    r = subprocess.run([sys.executable, "-m", "pytest"] + target + ["-q", "-x", "-m", "gpu"], env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout[-600:], r.stderr[-400:])
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-800:]
