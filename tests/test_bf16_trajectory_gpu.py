"""bf16 stays on the fp32 trajectory over many steps: the proxy this repo can offer for north_star's "mIoU within +-0.3 of the
reference", which needs VOC and ImageNet weights that are not here (re-posed after VERDICT r03).

* 3 steps at 65^2 (B=2+2): fp32 HIP path against the fp32 CPU oracle step by step with the oracle's sampler draws injected - SGD
  momentum, the EMA teacher and the prototype EMA carried across steps (/root/reference/mix_label.py:162-196,
  generalframeworks/networks/ddp_model.py:93-97) - then the bf16 HIP path on the same steps.  An oracle parity test: collected with them.
* 30 steps at 129^2, B=4+4, K=21 at the training lr (6.4e-3) - a regime that is CHAOTIC on a random-init network (|grad| ~ 1e4 |param|:
  a last-bit difference is amplified to per cents within a few steps).  Since round 4 every reduction of the step is ordered, so a run is a
  pure function of its inputs (tests/test_determinism_gpu.py: bit-identical twice) and the question "does bf16 differ, or does the run
  differ" can be put properly: an ENSEMBLE of fp32 runs whose input images are perturbed by one unit in the last place (x (1 +- 2^-23),
  the smallest perturbation there is) measures how far apart two legitimate fp32 trajectories of this problem end up; the bf16 run must
  lie inside a fixed multiple (ENV_C = 2) of that envelope - in TIME lag between the supervised-loss curves, in the mean of the last
  five steps, in the cosines of prototypes and weights.  Round 5: the envelope is the SECOND-LARGEST member value (robust
  against one member's excursion) and every bound is capped by an absolute figure (lag 15 steps, tail 3x, cosines) - see the end of the test.
* 20 steps of the same problem at 1/32 of the training lr, where two fp32 runs one ulp apart stay together: bf16 against fp32 under
  ABSOLUTE per-step bounds (test_bf16_tracks_fp32_in_a_calm_regime).
  Asserted absolutely: every run is finite, and the MEDIAN run of the ten (fp32, eight perturbed fp32, bf16) takes the supervised loss below a
  tenth of its start - a single run may not: in the second deterministic "universe" of round 4 (same test, weight gradients summed in
  another fixed order) the member fp32+ulp3 bounced back to 7.3 at step 9 and ended at 1.8, 21 steps behind the base run, which is exactly what
  the envelope is for.  Eight members: bf16 fails only if it lies outside TWICE the worst of eight legitimate fp32 runs; with the lags seen so
  far (2.4 - 21 steps over ten fp32 runs, bf16 2.6 and 2.55) that is a per-cent-level event under "bf16 behaves like fp32", not a coin flip.
  The full curves of all members are printed (the records of round 4: profiles/r04_determinism_and_trajectory_ensemble.txt, r04_trajectory_universe2.txt).
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


ENV_C = 2.0           # bf16 must lie inside ENV_C x the spread of the fp32 ensemble
N_MEMBERS = 8         # fp32 runs with 1-ulp input perturbations


def _perturb_ulp(x, seed):
    """x (1 +- 2^-23): every element moves by about one unit in the last place, sign from a seeded coin."""
    g = torch.Generator().manual_seed(1000 + seed)
    sign = torch.randint(0, 2, x.shape, generator=g).float() * 2.0 - 1.0
    return x * (1.0 + sign * 2.0 ** -23)


def _thirty(dtype, l_img, l_lab, u_img, steps, S, seed, gain):
    tr = _trainer(S, seed, gain, dtype, float(os.environ.get("CSS_TRAJ_LR", "6.4e-3")), 256, 512)
    np.random.seed(0)
    torch.manual_seed(0)
    hist = []
    for _ in range(steps):
        r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()))
        hist.append({k: float(r[k]) for k in ("sup", "unsup", "contrast", "total")})
    out = (hist, tr.prototypes.cpu().double(), tr.flat_p.detach().cpu().double())
    del tr
    torch.cuda.empty_cache()
    return out


def _metrics(base, run):
    """Distance of ``run`` from the fp32 baseline: max |time lag| of the supervised curve, relative gap of the last-five-steps mean, worst
    relative contrastive-loss gap, min prototype cosine, cosine of the centred weight vectors."""
    (hf, pf, wf), (hb, pb, wb) = base, run
    sf, sb = [d["sup"] for d in hf], [d["sup"] for d in hb]
    lags = [_lag(sf, i, b) for i, b in enumerate(sb)]
    tail = abs(np.mean(sb[-5:]) - np.mean(sf[-5:])) / np.mean(sf[-5:])
    worst_c = max(abs(a["contrast"] - b["contrast"]) / max(abs(a["contrast"]), 1.0) for a, b in zip(hf, hb))
    present = pf.abs().sum(1) > 0
    cos = float(torch.nn.functional.cosine_similarity(pf[present], pb[present], dim=1).min())
    wcos = float(torch.nn.functional.cosine_similarity(wf - wf.mean(), wb - wb.mean(), dim=0))
    return dict(lag=max(abs(v) for v in lags), tail=tail, contrast=worst_c, proto_cos=cos, w_cos=wcos, lags=lags)


def test_thirty_steps_bf16_inside_fp32_ensemble():
    S, B, seed, gain, steps = 129, 4, 11, 0.25, 30
    l_img, l_lab, u_img = _batch(S, B, 5, 16)
    base = _thirty(torch.float32, l_img, l_lab, u_img, steps, S, seed, gain)
    members = [_thirty(torch.float32, _perturb_ulp(l_img, k), l_lab, _perturb_ulp(u_img, 100 + k), steps, S, seed, gain) for k in range(N_MEMBERS)]
    bf = _thirty(torch.bfloat16, l_img, l_lab, u_img, steps, S, seed, gain)
    names = ["fp32"] + [f"fp32+ulp{k}" for k in range(N_MEMBERS)] + ["bf16"]
    print("supervised loss per step:", " ".join(f"{n:>10s}" for n in names))
    for i in range(steps):
        print(f"step {i:2d}                 ", " ".join(f"{r[0][i]['sup']:10.4f}" for r in [base] + members + [bf]))
    print("contrastive loss per step (fp32 / bf16):", " ".join(f"{a['contrast']:.4f}/{b['contrast']:.4f}" for a, b in zip(base[0], bf[0])))
    mm = [_metrics(base, m) for m in members]
    mb = _metrics(base, bf)
    for n, m in zip(names[1:], mm + [mb]):
        print(f"{n:>10s} vs fp32: max |lag| {m['lag']:.2f} steps, last-five-steps gap {m['tail']:.3f}, contrast {m['contrast']:.4f}, "
              f"prototype cosine {m['proto_cos']:.4f}, weights cosine {m['w_cos']:.5f}; lags " + " ".join(f"{v:+.1f}" for v in m["lags"]))
    # asserted absolutely: finite everywhere; the median run learns (a single run may take an excursion - that is what the envelope is for)
    runs = [base, bf] + members
    for h, _, _ in runs:
        assert all(math.isfinite(v) for d in h for k, v in d.items() if k != "unsup")        # (unsup is NaN by definition when no pixel is valid)
    ends = sorted(np.mean([d["sup"] for d in h[-3:]]) / h[0]["sup"] for h, _, _ in runs)
    print("mean of the last three supervised losses over the first, per run, sorted:", " ".join(f"{e:.4f}" for e in ends))
    assert ends[len(ends) // 2] < 0.1, ends
    # everything else relative to the envelope of the fp32 ensemble (floors: half a step of lag, the resolution of the lag measure;
    # 1e-3 on the cosines' distance from 1; 0.5 % on the contrastive loss, whose sampler is seeded and whose logits are normalised)
    # ROBUST envelope (the MAX over eight chaotic members gave bounds - 32 to 42 steps of lag, 8x to 44x on the tail - that could not
    # fail): the SECOND-LARGEST member value, so one member's excursion does not widen it, and every bound is CAPPED by an absolute figure that a
    # broken bf16 path would violate (a path that does not learn sits ~30 steps behind with a tail gap of ~60; cosines >= 0.97 / 0.98,
    # contrastive gap <= 2 %).  Round 3's fixed caps (lag <= 8 steps, tail <= 50 %) do NOT hold for fp32 itself here: in the first universe of
    # round 5 two of eight 1-ulp fp32 members sit 8.8 and 12.6 steps from the base run (tail gaps 0.80 / 1.53) and bf16 sits at 8.6 / 0.77 -
    # inside the fp32 spread, outside those caps (gpurun_out/r05_run1.txt).  The ABSOLUTE per-step bounds live in the calm-regime test below.
    # (A run is a pure function of its inputs since round 4, so this is not a coin flip per run: it is decided once per source tree.)
    def second(vals):
        v = sorted(vals)
        return v[-2] if len(v) > 1 else v[-1]
    env = dict(lag=max(second(m["lag"] for m in mm), 0.5), tail=second(m["tail"] for m in mm), contrast=max(second(m["contrast"] for m in mm), 5e-3),
               proto=max(second(1.0 - m["proto_cos"] for m in mm), 1e-3), w=max(second(1.0 - m["w_cos"] for m in mm), 1e-3))
    cap = dict(lag=15.0, tail=3.0, contrast=2e-2, proto=0.03, w=0.02)
    bound = {k: min(ENV_C * env[k], cap[k]) for k in env}
    print(f"fp32 ensemble envelope (second-largest of {N_MEMBERS}): lag {env['lag']:.2f} steps, tail {env['tail']:.3f}, contrast {env['contrast']:.4f}, "
          f"1 - prototype cosine {env['proto']:.4f}, 1 - weights cosine {env['w']:.5f};  bf16 must stay inside min({ENV_C} x these, {cap}) = {bound}")
    assert mb["lag"] <= bound["lag"], (mb["lag"], bound)
    assert mb["tail"] <= bound["tail"], (mb["tail"], bound)
    assert mb["contrast"] <= bound["contrast"], (mb["contrast"], bound)
    assert 1.0 - mb["proto_cos"] <= bound["proto"] and 1.0 - mb["w_cos"] <= bound["w"], (mb, bound)


def test_bf16_tracks_fp32_in_a_calm_regime():
    """The same question where the problem is NOT chaotic ("compare bf16 against the ensemble in a calmer regime, keep at least one
    absolute bound that a broken bf16 path would violate"): the same 129^2, B = 4 + 4 problem at lr = CSS_TRAJ_LR_CALM (default 2e-4, 1/32 of the
    training rate) for 20 steps.  There two fp32 runs that differ by 1 ulp of input stay together, so bf16 can be held to ABSOLUTE bounds
    step by step: supervised loss within 25 % of fp32 at every step (a path that does not learn is 49 % off at the end), contrastive loss within
    2 %, prototype cosine >= 0.98, centred weight cosine >= 0.99 - and the perturbed fp32 run, printed beside it, shows how much of that is the
    problem's own spread (12 % on the supervised loss: the problem is chaotic at ANY useful rate)."""
    S, B, seed, gain, steps = 129, 4, 11, 0.25, 20
    lr = os.environ.get("CSS_TRAJ_LR_CALM", "2e-4")
    l_img, l_lab, u_img = _batch(S, B, 5, 16)
    prev = os.environ.get("CSS_TRAJ_LR")
    os.environ["CSS_TRAJ_LR"] = lr
    try:
        base = _thirty(torch.float32, l_img, l_lab, u_img, steps, S, seed, gain)
        pert = _thirty(torch.float32, _perturb_ulp(l_img, 0), l_lab, _perturb_ulp(u_img, 100), steps, S, seed, gain)
        bf = _thirty(torch.bfloat16, l_img, l_lab, u_img, steps, S, seed, gain)
    finally:
        if prev is None:
            os.environ.pop("CSS_TRAJ_LR", None)
        else:
            os.environ["CSS_TRAJ_LR"] = prev
    for i in range(steps):
        print(f"step {i:2d} sup fp32 {base[0][i]['sup']:.4f} fp32+ulp {pert[0][i]['sup']:.4f} bf16 {bf[0][i]['sup']:.4f} | contrast "
              f"{base[0][i]['contrast']:.4f} {pert[0][i]['contrast']:.4f} {bf[0][i]['contrast']:.4f}")
    def gaps(run):
        g_sup = max(abs(a["sup"] - b["sup"]) / abs(a["sup"]) for a, b in zip(base[0], run[0]))
        g_con = max(abs(a["contrast"] - b["contrast"]) / max(abs(a["contrast"]), 1.0) for a, b in zip(base[0], run[0]))
        present = base[1].abs().sum(1) > 0
        pc = float(torch.nn.functional.cosine_similarity(base[1][present], run[1][present], dim=1).min())
        wc = float(torch.nn.functional.cosine_similarity(base[2] - base[2].mean(), run[2] - run[2].mean(), dim=0))
        dw = float(torch.nn.functional.cosine_similarity(run[2] - w0, base[2] - w0, dim=0))
        return g_sup, g_con, pc, wc, dw
    w0 = _thirty(torch.float32, l_img, l_lab, u_img, 0, S, seed, gain)[2]               # the initial flat weights (no step)
    gp, gb = gaps(pert), gaps(bf)
    print(f"calm regime (lr {lr}): fp32+ulp vs fp32: sup {gp[0]:.4f} contrast {gp[1]:.4f} proto cos {gp[2]:.5f} w cos {gp[3]:.6f} update cos {gp[4]:.5f}")
    print(f"calm regime (lr {lr}):     bf16 vs fp32: sup {gb[0]:.4f} contrast {gb[1]:.4f} proto cos {gb[2]:.5f} w cos {gb[3]:.6f} update cos {gb[4]:.5f}")
    assert all(math.isfinite(d["sup"]) and math.isfinite(d["contrast"]) for d in bf[0])
    # ABSOLUTE bounds (measured on MI355X, round 5: bf16 0.106 / 0.0018 / 0.9981 / 0.99974, the fp32 pair 0.121 / 0.0015 / 0.9982 / 0.99988).  Even
    # at 1/32 of the training rate two fp32 runs ONE ULP apart differ by 12 % in the supervised loss of some step, and their accumulated
    # 20-step updates are nearly orthogonal (cosine 0.03: the gradient of this random-init network is dominated by components that flip sign
    # with the ReLU masks) - the 5 % per-step bound first written here does not hold for fp32 itself, and the update cosine is printed, not
    # asserted.  What a BROKEN bf16 path would violate: a path that does not learn stays at 7.0 while fp32 reaches 4.7 (gap 0.49 > 0.25);
    # wrong statistics / wrong prototypes show in the contrastive loss (2 %) and the prototype cosine (0.98).
    assert base[0][-1]["sup"] < 0.85 * base[0][0]["sup"] and bf[0][-1]["sup"] < 0.85 * bf[0][0]["sup"], (base[0][-1], bf[0][-1])      # both learn
    assert gb[0] <= 0.25 and gb[1] <= 0.02 and gb[2] >= 0.98 and gb[3] >= 0.99, gb
    assert gb[0] <= 2.5 * max(gp[0], 0.04), (gb[0], gp[0])          # ... and relative to the fp32 pair of the same run
