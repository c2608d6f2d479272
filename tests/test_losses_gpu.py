"""Similarity / pseudo-label / cross-entropy / contrastive kernels through the C ABI against the reference's golden
vectors (tests/golden/*.npz) and against the CPU oracle on fresh seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import dev, rel_err  # noqa: E402


def T(a):
    return torch.from_numpy(np.asarray(a))


def cl(t):  # logical NCHW on GPU in channels_last memory
    return t.to(dev()).contiguous(memory_format=torch.channels_last)


# ---------------- similarity + pseudo labels -----------------------------------------------------------------
@pytest.mark.parametrize("name", ["rand", "zero"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_similarity_and_pseudo_labels_golden(golden, name, dtype):
    from css_amd import functional as Fn
    g = golden("pseudo_labels")
    pred_u, rep_u, protos = T(g[f"{name}::pred_u"]), T(g[f"{name}::rep_u"]), T(g[f"{name}::protos"])
    rep = Fn.nhwc(rep_u.to(dev()).to(dtype))
    pred = Fn.nhwc(pred_u.to(dev()).to(dtype))
    sim, prob, _ = Fn.similarity(rep, protos.to(dev()), 0.5, want_sim=True, want_prob=True)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(sim.cpu().permute(0, 3, 1, 2), T(g[f"{name}::sim"])) < tol or name == "zero"
    assert (sim.cpu().permute(0, 3, 1, 2) - T(g[f"{name}::sim"])).abs().max() < tol
    assert (prob.cpu().permute(0, 3, 1, 2) - T(g[f"{name}::prob_all"])).abs().max() < tol
    lr, ar, lc, ac, ps = Fn.pseudo_labels(sim, pred, 0.5, (65, 65))
    if dtype == torch.float32:
        assert (lr.cpu() - T(g[f"{name}::lg_rep"])).abs().max() < 2e-5
        assert (lc.cpu() - T(g[f"{name}::lg_cls"])).abs().max() < 2e-5
        for got, key in ((ar, "lb_rep"), (ac, "lb_cls")):
            mism = (got.cpu() != T(g[f"{name}::{key}"])).float().mean().item()
            assert mism < 1e-3, (key, mism)      # arg-max ties between near-equal classes only
        mism = (ps.cpu() != T(g[f"{name}::pseudo"])).float().mean().item()
        assert mism < 2e-3
    else:
        assert (lc.cpu() - T(g[f"{name}::lg_cls"])).abs().max() < 5e-2


def test_similarity_large_vs_oracle():
    from oracle import css_oracle as O
    from css_amd import functional as Fn
    g = torch.Generator().manual_seed(0)
    rep = torch.randn(3, 256, 33, 35, generator=g)
    protos = torch.randn(19, 256, generator=g)
    sim_ref = O.similarity(rep, protos)
    prob_ref = O.prob_all_from_rep(rep, protos, 0.25)
    sim, prob, _ = Fn.similarity(Fn.nhwc(rep.to(dev())), protos.to(dev()), 0.25, True, True)
    assert (sim.cpu().permute(0, 3, 1, 2) - sim_ref).abs().max() < 2e-6
    assert (prob.cpu().permute(0, 3, 1, 2) - prob_ref).abs().max() < 2e-6


@pytest.mark.parametrize("H,h", [(65, 17), (129, 33), (97, 25)])
def test_class_map_golden(golden, H, h):
    from css_amd import functional as Fn
    g = golden("label_mask")
    l_lab, u_lab = T(g[f"{H}::l_lab"]).long(), T(g[f"{H}::u_lab"]).long()
    u_logits = T(g[f"{H}::u_logits"])
    label_all, mask_all = T(g[f"{H}::label_all"]).float(), T(g[f"{H}::mask_all"]).float()
    valid = label_all * mask_all                      # [2B,K,h,h]
    ref = torch.where(valid.sum(1) > 0, valid.argmax(1), torch.full_like(valid.argmax(1), -1)).flatten()
    cls = Fn.class_map(l_lab.to(dev()), u_lab.to(dev()), u_logits.to(dev()), 0.7, (h, h))
    assert torch.equal(cls.cpu().long(), ref)


# ---------------- cross-entropy family ---------------------------------------------------------------------------
def _ce_case(g, prefix):
    pred = T(g[f"{prefix}::pred"])
    return pred, T(g[f"{prefix}::lab"]).long(), T(g[f"{prefix}::loss"]), T(g[f"{prefix}::grad"])


def test_ce_golden(golden):
    from css_amd.loss.loss import CrossEntropyLoss
    pred, lab, loss, grad = _ce_case(golden("losses"), "ce")
    p = cl(pred).requires_grad_(True)
    out = CrossEntropyLoss(-1)(p, lab.to(dev()))
    out.backward()
    assert abs(out.item() - loss.item()) < 2e-6 * abs(loss.item()) + 1e-6
    assert rel_err(p.grad.cpu(), grad) < 2e-5


@pytest.mark.parametrize("name", ["normal", "allignored"])
def test_attention_threshold_golden(golden, name):
    from css_amd.loss.loss import Attention_Threshold_Loss
    g = golden("losses")
    pred, lab, loss, grad = _ce_case(g, f"att_{name}")
    p = cl(pred).requires_grad_(True)
    out = Attention_Threshold_Loss(0.7)(p, lab.to(dev()), T(g[f"att_{name}::logits"]).to(dev()))
    out.backward()
    if name == "allignored":
        assert torch.isnan(out) and p.grad.abs().max() == 0
    else:
        assert abs(out.item() - loss.item()) < 2e-6 * abs(loss.item()) + 1e-6
        assert rel_err(p.grad.cpu(), grad) < 2e-5


@pytest.mark.parametrize("name", ["raise", "keep", "toofew"])
def test_ohem_golden(golden, name):
    from css_amd.loss.loss import ProbOhemCrossEntropy2d
    g = golden("losses")
    pred, lab, loss, grad = _ce_case(g, f"ohem_{name}")
    p = cl(pred).requires_grad_(True)
    out = ProbOhemCrossEntropy2d(-1, thresh=0.7, min_kept=int(g[f"ohem_{name}::min_kept"]))(p, lab.to(dev()))
    out.backward()
    assert abs(out.item() - loss.item()) < 1e-5 * abs(loss.item()) + 1e-6
    assert rel_err(p.grad.cpu(), grad) < 5e-5


def test_non_finite_logits_surface_as_nan_losses():
    """the loss statistics are 64-bit fixed-point integers (order-independent sums) and float -> integer conversion of NaN is 0 on
    AMDGPU - a diverged network reported a FINITE supervised loss.  A non-finite pixel loss now poisons its image's accumulator and the
    finalize kernel returns NaN, like the reference's fp32 mean (mix_label.py:169; loss.py:57-62), on the materialising path, on the fused
    low-resolution path and for the attention-threshold loss; a clean input next to it stays finite."""
    from css_amd.loss.loss import CrossEntropyLoss, Attention_Threshold_Loss, ProbOhemCrossEntropy2d
    g = torch.Generator().manual_seed(5)
    B, K, H, h = 2, 21, 129, 33
    lab = torch.randint(0, K, (B, H, H), generator=g).to(dev())
    conf = torch.rand(B, H, H, generator=g).to(dev())
    for bad in (float("nan"), float("inf")):
        big = torch.randn(B, K, H, H, generator=g)
        small = torch.randn(B, h, h, K, generator=g)
        clean = [CrossEntropyLoss(-1)(big.to(dev()), lab), CrossEntropyLoss(-1).forward_small(small.to(dev()), lab),
                 Attention_Threshold_Loss(0.5)(big.to(dev()), lab, conf), ProbOhemCrossEntropy2d(-1, thresh=0.7, min_kept=100).forward_small(small.to(dev()), lab)]
        assert all(torch.isfinite(c) for c in clean), clean
        big[1, 3, 40, 50] = bad                                     # one logit of one pixel of the second image
        small[1, 10, 12, 3] = bad
        outs = [CrossEntropyLoss(-1)(big.to(dev()), lab), CrossEntropyLoss(-1).forward_small(small.to(dev()), lab),
                Attention_Threshold_Loss(0.5)(big.to(dev()), lab, conf), Attention_Threshold_Loss(0.5).forward_small(small.to(dev()), lab, conf)]
        ref = torch.nn.functional.cross_entropy(big, lab.cpu(), ignore_index=-1)
        print(bad, [float(o) for o in outs], "torch:", float(ref))
        assert torch.isnan(ref) and all(torch.isnan(o) for o in outs), (bad, outs)


def test_ce_nchw_input_and_big():
    """Plain NCHW-contiguous input (what an unmodified caller passes) and a size that spans many tiles/images."""
    from oracle import css_oracle as O
    from css_amd.loss.loss import CrossEntropyLoss, Attention_Threshold_Loss
    g = torch.Generator().manual_seed(2)
    pred = torch.randn(3, 21, 129, 131, generator=g) * 3
    lab = torch.randint(-1, 21, (3, 129, 131), generator=g)
    conf = torch.rand(3, 129, 131, generator=g)
    for crit, ref in ((CrossEntropyLoss(-1), lambda p: O.ce_loss(p, lab)),
                      (Attention_Threshold_Loss(0.6), lambda p: O.attention_threshold_loss(p, lab, conf, 0.6))):
        pr = pred.clone().requires_grad_(True)
        lr = ref(pr)
        lr.backward()
        pg = pred.to(dev()).requires_grad_(True)
        args = (pg, lab.to(dev())) if isinstance(crit, CrossEntropyLoss) else (pg, lab.to(dev()), conf.to(dev()))
        lg = crit(*args)
        (lg * 2.0).backward()
        assert abs(lg.item() - lr.item()) < 1e-5 * abs(lr.item())
        assert rel_err(pg.grad.cpu(), 2.0 * pr.grad) < 2e-5


# ---------------- contrastive loss ---------------------------------------------------------------------------------
def _contrast_case(g, name):
    K = 21
    rep = T(g[f"{name}::rep"])
    label = F.one_hot(T(g[f"{name}::label_idx"]).long(), K).permute(0, 3, 1, 2).float()
    mask = T(g[f"{name}::mask"]).float()
    prob = T(g[f"{name}::prob"])
    protos = T(g[f"{name}::protos_in"]).clone()
    Q, N = [int(v) for v in g[f"{name}::QN"]]
    return rep, label, mask, prob, protos, Q, N


def _injection_from_golden(g, name, rep, label, mask, prob, protos, Q, N):
    from oracle import css_oracle as O
    na = int(g[f"{name}::n_anchor"])
    if not na:
        return None
    rec = {}
    O.contrast_loss(rep, label, mask, prob, protos.clone(), Q, N, 0.5, 0.8, 0.99, record=rec)
    anchors, negs, j = [], [], 0
    for hn in rec["hard_num"]:
        if hn > 0:
            anchors.append(g[f"{name}::anchor{j}"].astype(np.int64))
            negs.append(g[f"{name}::negative{j}"].astype(np.int64))
            j += 1
        else:
            anchors.append(None)
            negs.append(None)
    return dict(anchor=anchors, negative=negs)


@pytest.mark.parametrize("name", ["first", "ema", "nohard", "single"])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_contrast_loss_golden_injected(golden, name, layout):
    from css_amd.loss.loss import Contrast_Loss
    g = golden("contrast_loss")
    rep, label, mask, prob, protos, Q, N = _contrast_case(g, name)
    inj = _injection_from_golden(g, name, rep, label, mask, prob, protos, Q, N)
    to = (lambda t: t.to(dev())) if layout == "nchw" else cl
    rg = to(rep).requires_grad_(True)
    pg = protos.to(dev())
    crit = Contrast_Loss(Q, N, temp=0.5, strong_threshold=0.8, alpha=0.99)
    loss = crit(rg, to(label), mask.to(dev()), to(prob), pg, _injected=inj)
    loss.backward()
    ref = float(g[f"{name}::loss"])
    assert abs(loss.item() - ref) < 2e-5 * max(1.0, abs(ref)), (loss.item(), ref)
    assert rel_err(pg.cpu(), T(g[f"{name}::protos_out"])) < 1e-5
    C = rep.shape[1]
    gr = rg.grad.cpu().permute(0, 2, 3, 1).reshape(-1, C)
    rows = T(g[f"{name}::grad_rows"]).long()
    if len(rows):
        assert rel_err(gr[rows], T(g[f"{name}::grad_vals"])) < 5e-5
    other = torch.ones(gr.shape[0], dtype=torch.bool)
    other[rows] = False
    assert gr[other].abs().max().item() == 0 if other.any() else True


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Q,N,K,present", [(256, 512, 21, 15), (1024, 2048, 19, 6)])
def test_contrast_loss_vs_oracle_replay(dtype, Q, N, K, present):
    """Bigger pools: draws recorded from the oracle's own RNG replay of the reference sampler are injected."""
    from oracle import css_oracle as O
    from css_amd.loss.loss import Contrast_Loss
    gen = torch.Generator().manual_seed(Q + K)
    B2, C, h = 4, 256, 33
    cls_ids = torch.randperm(K, generator=gen)[:present]
    lab = cls_ids[torch.randint(present, (B2, 5, 5), generator=gen)].repeat_interleave(7, 1).repeat_interleave(7, 2)[:, :h, :h]
    label = F.one_hot(lab, K).permute(0, 3, 1, 2).float()
    mask = (torch.rand(B2, 1, h, h, generator=gen) > 0.1).float()
    rep = torch.randn(B2, C, h, h, generator=gen)
    if dtype == torch.bfloat16:
        rep = rep.to(torch.bfloat16).float()
    prob = torch.softmax(torch.randn(B2, K, h, h, generator=gen) * 2, 1)
    protos = torch.randn(K, C, generator=gen)
    protos[cls_ids[0]] = 0
    torch.manual_seed(1)
    np.random.seed(1)
    rr = rep.clone().requires_grad_(True)
    pr, rec = protos.clone(), {}
    lref = O.contrast_loss(rr, label, mask, prob, pr, Q, N, 0.5, 0.8, 0.99, record=rec)
    lref.backward()
    rg = cl(rep.to(dtype)).requires_grad_(True)
    pg = protos.to(dev())
    crit = Contrast_Loss(Q, N, temp=0.5, strong_threshold=0.8, alpha=0.99)
    loss = crit(rg, cl(label), mask.to(dev()), cl(prob), pg, _injected=dict(anchor=rec["anchor"], negative=rec["negative"]))
    loss.backward()
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    assert abs(loss.item() - lref.item()) < tol * abs(lref.item())
    assert rel_err(pg.cpu(), pr) < (1e-5 if dtype == torch.float32 else 1e-3)
    assert rel_err(rg.grad.float().cpu(), rr.grad) < (5e-5 if dtype == torch.float32 else 2e-2)


def test_contrast_sampler_distribution():
    """The device sampler follows the reference's distributions (loss.py:127,133-140,410-418): anchors uniform over the
    hard pixels of the class, negative class ~ softmax(cos(proto)/temp) over the other present classes, negative pixel
    uniform inside the drawn class."""
    from oracle import css_oracle as O
    from css_amd.loss.loss import _ContrastCore
    from css_amd._lib import call, dev_stream, query
    import ctypes
    K, C, P, Q, N = 8, 256, 4096, 256, 512
    gen = torch.Generator().manual_seed(5)
    cls = torch.randint(-1, 5, (P,), generator=gen).int()          # classes 0..4 present, 5..7 absent
    cls[cls == 2] = -1                                              # class 2 absent too
    hard = (torch.rand(P, generator=gen) < 0.5).to(torch.uint8)
    hard[cls == 3] = 0                                              # class 3 present but without hard pixels
    rep = torch.randn(P, C, generator=gen)
    protos = torch.randn(K, C, generator=gen)
    d = dev()
    rep_g, cls_g, hard_g, protos_g = rep.to(d), cls.to(d), hard.to(d), protos.to(d).contiguous()
    dv, st = dev_stream(rep_g)
    i32 = dict(dtype=torch.int32, device=d)
    meta = torch.zeros(query("css_contrast_meta_bytes"), dtype=torch.uint8, device=d)
    chunk = torch.empty(query("css_contrast_nchunks", P) * 64, **i32)
    listV, listH = torch.empty(P, **i32), torch.empty(P, **i32)
    call("css_contrast_compact", cls_g, hard_g, P, K, chunk, listV, listH, meta, dv, st)
    m = meta.cpu().numpy().view(np.int32)
    V, present = int(m[0]), m[1:33]
    cntV, cntH, baseV, baseH = m[33:65], m[65:97], m[97:129], m[129:161]
    pres = [0, 1, 3, 4]
    assert V == 4 and list(present[:4]) == pres
    for k in range(K):
        vp = torch.nonzero(cls == k).flatten()
        hp = torch.nonzero((cls == k) & (hard > 0)).flatten()
        assert cntV[k] == len(vp) and cntH[k] == len(hp)
        assert torch.equal(listV.cpu()[baseV[k]:baseV[k] + cntV[k]].long(), vp)     # stable order = reference order
        assert torch.equal(listH.cpu()[baseH[k]:baseH[k] + cntH[k]].long(), hp)
    anchor = torch.zeros(K * Q, **i32)
    neg = torch.zeros(K * Q * N, **i32)
    cdf = torch.zeros(1024, dtype=torch.float32, device=d)
    call("css_contrast_sample", protos_g, C, meta, 0.5, cdf, listV, listH, Q, N, 1234, 1, anchor, neg, dv, st)
    anchor, neg = anchor.cpu().view(K, Q), neg.cpu().view(K, Q, N)
    proto_rep = protos[pres]
    for v, cid in enumerate(pres):
        if cid == 3:
            continue
        a = anchor[v].long()
        assert ((cls[a] == cid) & (hard[a] > 0)).all()
        ncls = cls[neg[v].long().flatten()].long()
        p_ref, order = O.class_negative_probs(proto_rep, v, 0.5)
        counts = torch.stack([(ncls == pres[o]).sum() for o in order]).double()
        assert counts.sum() == Q * N
        expect = p_ref.double() * Q * N
        chi2 = ((counts - expect) ** 2 / expect).sum().item()
        assert chi2 < 30, (v, chi2, counts, expect)              # 3 dof, p < 1e-6
        # uniform inside a class: bucket the within-class rank of the drawn pixel
        o0 = pres[order[0]]
        pool = torch.nonzero(cls == o0).flatten()
        rank = torch.searchsorted(pool, neg[v].long().flatten()[ncls == o0])
        hist = torch.histc(rank.float(), bins=8, min=0, max=len(pool)).double()
        e = hist.sum() / 8
        assert ((hist - e) ** 2 / e).sum().item() < 40           # 7 dof
    # a second call with another offset must give other draws
    anchor2 = torch.zeros(K * Q, **i32)
    neg2 = torch.zeros(K * Q * N, **i32)
    call("css_contrast_sample", protos_g, C, meta, 0.5, cdf, listV, listH, Q, N, 1234, 2, anchor2, neg2, dv, st)
    assert (neg2.cpu().view(K, Q, N)[0] != neg[0]).float().mean() > 0.5


# ---- losses with the bilinear up-sampling folded in (css_ce_small_*) ----------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["ce", "attention", "ohem"])
@pytest.mark.parametrize("geom", [(2, 21, 17, 17, 65, 65), (3, 19, 9, 13, 33, 49), (1, 21, 33, 33, 129, 129),
                                  (1, 5, 9, 9, 65, 65),       # up-sampling factor 8: LDS-atomic fallback of the backward
                                  (1, 30, 9, 9, 33, 33)])     # K too large for the per-wave copies: fallback as well
def test_losses_from_low_resolution_logits(geom, kind, dtype):
    """loss(small) == loss(F.interpolate(small, align_corners=True)) and so are the gradients w.r.t. the small logits: the fused
    kernels against the materialising path (bilinear op + full-resolution loss), which is itself pinned to the reference."""
    from css_amd import ops
    from css_amd.loss.loss import Attention_Threshold_Loss, CrossEntropyLoss, ProbOhemCrossEntropy2d
    b, k, h, w, hh, ww = geom
    g = torch.Generator().manual_seed(b * 100 + k)
    small = (torch.randn(b, h, w, k, generator=g) * 2).to(dev(), dtype)
    lab = torch.randint(-1, k, (b, hh, ww), generator=g).to(dev())
    conf = torch.rand(b, hh, ww, generator=g).to(dev())
    if kind == "ce":
        crit = CrossEntropyLoss(-1)
        args = (lab,)
    elif kind == "attention":
        crit = Attention_Threshold_Loss(0.6)
        args = (lab, conf)
    else:
        crit = ProbOhemCrossEntropy2d(-1, thresh=0.7, min_kept=(b * hh * ww) // 5)
        args = (lab,)
    s1 = small.clone().requires_grad_(True)
    l1 = crit.forward_small(s1, *args)
    l1.backward()
    s0 = small.clone().requires_grad_(True)
    l0 = crit(ops.bilinear(s0, hh, ww, torch.float32).permute(0, 3, 1, 2), *args)
    l0.backward()
    assert abs(float(l1) - float(l0)) < 2e-6 * max(1.0, abs(float(l0)))
    tol = 2e-5 if dtype == torch.float32 else 1e-2          # bf16: the gradient itself is rounded to bf16 on both paths
    assert rel_err(s1.grad.float().cpu(), s0.grad.float().cpu()) < tol
    assert torch.isfinite(s1.grad.float()).all()


def test_fused_loss_rejects_small_upsampling_factors():
    from css_amd._lib import CssHipError
    from css_amd.loss.loss import CrossEntropyLoss, fused_upsample_ok
    assert fused_upsample_ok((129, 129), (513, 513)) and fused_upsample_ok((193, 193), (769, 769)) and not fused_upsample_ok((33, 33), (40, 40))
    small = torch.randn(1, 33, 33, 5, device=dev(), requires_grad=True)
    lab = torch.zeros(1, 40, 40, dtype=torch.int64, device=dev())
    loss = CrossEntropyLoss(-1).forward_small(small, lab)          # forward works for any factor
    with pytest.raises(CssHipError):
        loss.backward()
