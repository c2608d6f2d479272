"""bf16 stays on the fp32 trajectory over many steps: the proxy this repo can offer for north_star's "mIoU within +-0.3 of the
reference", which needs VOC and ImageNet weights that are not here.

* 3 steps at 65^2 (B=2+2): fp32 HIP path against the fp32 CPU oracle step by step with the oracle's sampler draws injected - SGD
  momentum, the EMA teacher and the prototype EMA carried across steps (/root/reference/mix_label.py:162-196,
  generalframeworks/networks/ddp_model.py:93-97) - then the bf16 HIP path on the same steps;
* 30 steps at 129^2, B=4+4, K=21 at the training lr (6.4e-3): once with ``set_compute_dtype(bfloat16)``, once in fp32, both on the HIP
  path with the same seeds (weights, crops, device sampler).  Measured on MI355X: both runs take the supervised loss from 7.03 to
  0.10-0.11; the two curves run up to one and a half steps apart in TIME (29 % apart at equal step index around step 5, where the loss
  halves every two steps; 17 % at step 28 of another run, where it falls 7 % per step) and which one leads changes from run to run
  (fp32 atomics in the loss backward: the order of their adds is not fixed).  Asserted: no NaN; both runs end below a tenth of the
  initial loss; the TIME lag between the curves (for every bf16 loss: distance to the nearest step at which the piecewise-linear fp32
  curve takes that value) stays within 8 steps (measured on three boxes: up to 2.3, 3.9 and 4.6 - one early event, a plateau left one
  step sooner or later, shifts the rest of a curve; a second fp32 run is printed against the first as the yardstick of that spread); the
  means of the last five steps agree within 50 % (measured 13-25 %: 4 steps of lag where the loss falls 7 % per step);
  the contrastive loss agrees within 1 % at every step; prototype cosine >= 0.98 and cosine of the centred weight vectors >= 0.98
  (measured 0.991-0.9997) at step 30.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import dev  # noqa: E402

K = 21


def _trainer(S, seed, gain, dtype, lr, Q, N, total_iter=1000):
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from css_amd.train_step import MixTrainer
    from oracle import css_oracle as O
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none", "device_aug": "identity"}}
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.5)
    sd = O.init_state("tv", K, 256, seed, gain)
    m.model.load_state_dict(sd)
    m.ema_model.load_state_dict(sd)
    m = m.to(dev()).train().set_compute_dtype(dtype)
    return MixTrainer(m, K, lr=lr, total_iter=total_iter, min_lr=0.0, num_queries=Q, num_negatives=N, strong_threshold=0.8,
                      weak_threshold=0.0, un_threshold=0.97)


def _batch(S, B, seed, block):
    g = torch.Generator().manual_seed(seed)
    l_img, u_img = torch.randn(B, 3, S, S, generator=g), torch.randn(B, 3, S, S, generator=g)
    nb = (S + block - 1) // block
    l_lab = torch.randint(0, K, (B, nb, nb), generator=g).repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :S, :S].clone()
    # make the label a function of the image so that there is something to learn: class c shifts the three channels by a fixed code
    code = torch.randn(K, 3, generator=g)
    l_img = l_img * 0.5 + code[l_lab].permute(0, 3, 1, 2)
    return l_img, l_lab, u_img


def test_three_steps_fp32_and_bf16_vs_oracle():
    from oracle import css_oracle as O
    S, B, seed, gain = 65, 2, 7, 0.25
    l_img, l_lab, u_img = _batch(S, B, 3, 13)
    # lr: on a random-init network |grad| ~ 1e4 |param|, so at the training lr one SGD step moves the logits chaotically (a handful of
    # ReLU-mask flips in the gradient change the next loss by per cents: tests/test_train_step_gpu.py compares its second step at
    # 0.35 for that reason; measured here at 1e-5: 5e-3 apart by the third step); 1e-6 keeps the three steps in the regime where losses can be compared tightly while momentum, the EMA
    # teacher and the prototype EMA still carry state from step to step
    lr3 = float(os.environ.get("CSS_TRAJ_LR3", "1e-6"))
    args = dict(lr=lr3, temp_model=0.5, strong_threshold=0.8, weak_threshold=0.0, un_threshold=0.97, num_queries=64, num_negatives=128)
    st = O.MixState("tv", K, 256, seed, gain)
    torch.manual_seed(0)
    np.random.seed(0)
    recs, refs = [], []
    for _ in range(3):
        rec = {}
        refs.append({k: float(v) for k, v in O.train_step_mix(st, l_img, l_lab, u_img, record=rec, **args).items() if k in ("sup", "unsup", "contrast")})
        recs.append(rec)
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 5e-2)):       # measured: 3e-4 / 5e-4 (fp32), 2.4e-2 (bf16)
        tr = _trainer(S, seed, gain, dtype, lr3, 64, 128, total_iter=10 ** 9)
        for i in range(3):
            r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()), _injected=dict(anchor=recs[i]["anchor"], negative=recs[i]["negative"]))
            for key in ("sup", "contrast"):
                a, b = float(r[key]), refs[i][key]
                print(f"{dtype} step {i} {key}: hip {a:.6f} oracle {b:.6f}")
                assert math.isfinite(a) and abs(a - b) <= tol * max(1.0, abs(b)), (dtype, i, key, a, b)
        pa, pb = tr.prototypes.cpu().double(), st.prototypes.double()
        present = pb.abs().sum(1) > 0
        cos = torch.nn.functional.cosine_similarity(pa[present], pb[present], dim=1)
        print(dtype, "prototype cosine after 3 steps: min", float(cos.min()))
        assert float(cos.min()) > (0.9999 if dtype == torch.float32 else 0.97)
        del tr
        torch.cuda.empty_cache()


def _lag(sf, i, b):
    """Distance in steps from index i to the nearest point where the piecewise-linear fp32 curve ``sf`` takes the value ``b`` (signed: + when
    the fp32 curve reached ``b`` earlier); beyond the last fp32 step the curve is continued with the decay rate of its last three steps."""
    best = None
    for j in range(len(sf) - 1):
        lo, hi = sorted((sf[j], sf[j + 1]))
        if lo <= b <= hi:
            t = j + (0.0 if hi == lo else (sf[j] - b) / (sf[j] - sf[j + 1]))
            if best is None or abs(t - i) < abs(best - i):
                best = t
    if best is None:
        if b >= max(sf):
            best = float(int(np.argmax(sf)))
        else:
            rate = (sf[-1] / sf[-4]) ** (1.0 / 3.0)
            best = len(sf) - 1 + math.log(b / sf[-1]) / math.log(min(rate, 0.999))
    return i - best


def test_thirty_steps_bf16_tracks_fp32():
    S, B, seed, gain, steps = 129, 4, 11, 0.25, 30
    l_img, l_lab, u_img = _batch(S, B, 5, 16)
    runs = {}
    # the fp32 run twice: its own run-to-run spread (fp32 atomics in the loss backward add in no fixed order) is the yardstick printed below
    for name, dtype in (("f32", torch.float32), ("f32_again", torch.float32), ("bf16", torch.bfloat16)):
        tr = _trainer(S, seed, gain, dtype, float(os.environ.get("CSS_TRAJ_LR", "6.4e-3")), 256, 512)
        np.random.seed(0)
        torch.manual_seed(0)
        hist = []
        for _ in range(steps):
            r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()))
            hist.append({k: float(r[k]) for k in ("sup", "unsup", "contrast", "total")})
        runs[name] = (hist, tr.prototypes.cpu().double(), tr.flat_p.detach().cpu().double())
        del tr
        torch.cuda.empty_cache()
    hf, pf, wf = runs["f32"]
    hb, pb, wb = runs["bf16"]
    for i in range(steps):
        print(f"step {i:2d}  fp32 sup {hf[i]['sup']:.4f} contrast {hf[i]['contrast']:.4f}   bf16 sup {hb[i]['sup']:.4f} contrast {hb[i]['contrast']:.4f}")
    for h in (hf, hb):
        assert all(math.isfinite(v) for d in h for k, v in d.items() if k != "unsup")        # (unsup is NaN by definition when no pixel is valid)
        assert np.mean([d["sup"] for d in h[-3:]]) < 0.1 * h[0]["sup"], (h[0]["sup"], h[-1]["sup"])      # the supervised loss goes down
    sf, sb = [d["sup"] for d in hf], [d["sup"] for d in hb]
    worst = max(abs(a - b) / a for a, b in zip(sf, sb))
    lags = [_lag(sf, i, b) for i, b in enumerate(sb)]
    lags_ref = [_lag(sf, i, b) for i, b in enumerate(d["sup"] for d in runs["f32_again"][0])]
    tail = abs(np.mean(sb[-5:]) - np.mean(sf[-5:])) / np.mean(sf[-5:])
    worst_c = max(abs(a["contrast"] - b["contrast"]) / max(abs(a["contrast"]), 1.0) for a, b in zip(hf, hb))
    present = pf.abs().sum(1) > 0
    cos = torch.nn.functional.cosine_similarity(pf[present], pb[present], dim=1)
    wcos = float(torch.nn.functional.cosine_similarity(wf - wf.mean(), wb - wb.mean(), dim=0))
    print(f"30 steps: worst |d sup| / sup at equal step index {worst:.4f}, last five steps {tail:.4f}, contrast {worst_c:.4f}; "
          f"prototype cosine min {float(cos.min()):.4f}; weights cosine {wcos:.6f}")
    print("lag of the bf16 curve behind (+) / ahead of (-) the fp32 curve, in steps:", " ".join(f"{v:+.1f}" for v in lags))
    print("the same for a second fp32 run against the first:                          ", " ".join(f"{v:+.1f}" for v in lags_ref))
    assert max(abs(v) for v in lags) <= 8.0, lags
    assert tail <= 0.5 and worst_c <= 0.01, (tail, worst_c)
    assert float(cos.min()) >= 0.98 and wcos >= 0.98
