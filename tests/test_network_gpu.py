"""DeepLabv3+/ResNet-101 forward+backward on the GPU (fp32 parity path) against golden vectors captured from the
reference (tests/golden/net_*.npz).

Two regimes (measured, see DESIGN.md "Parity"):
* ``*_damped`` fixtures (bn3 gains x0.25, i.e. the well-conditioned regime of a trained network): the 1e-3 relative
  bar on logits that BASELINE.json's north_star states, asserted strictly.
* undamped random-init fixtures with batch 2: the network itself amplifies a 1-ulp perturbation (input, weights, or
  the summation order of the oracle's own GEMMs) to 0.5 ... 1.8e-3 and the reference's own fp32 CPU output is only
  within 0.5 ... 3.8e-3 of an fp64 evaluation.  The HIP-vs-reference error is printed and asserted against
  max(1e-3, 2 x that measured floor) - the gap to north_star's 1e-3 is explicit in the log, not hidden in a flat
  5e-3 - plus: as close to the fp64 truth as the reference's fp32 path, and a per-stage guard.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev, rel_err  # noqa: E402


def build(backbone, K, seed, residual_gain=1.0):
    from css_amd.networks import resnet
    from css_amd.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep
    from oracle import css_oracle as O
    bb = resnet.resnet101_tv() if backbone == "tv" else resnet.resnet101()
    net = DeepLabv3Plus_with_rep(bb, dilate_scale=8, num_classes=K, output_dim=256)
    net.load_state_dict(O.init_state(backbone, K, 256, seed, residual_gain), strict=True)
    return net.to(dev())


def probe_slice(t):
    return t.detach().flatten()[:: max(1, t.numel() // 2048)][:2048]


def fp64_truth(backbone, K, seed, gain, x):
    from oracle import css_oracle as O
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in O.init_state(backbone, K, 256, seed, gain).items()}
    with torch.no_grad():
        p, r = O.deeplab_forward(sd, x.double(), backbone, True, K, 256)
    return p, r


CASES = [("net_tv_65_damped", "tv"), ("net_stem_65_damped", "stem"), ("net_tv_65", "tv"), ("net_stem_65", "stem"),
         ("net_tv_97", "tv")]


@pytest.mark.parametrize("tag,backbone", CASES)
def test_network_fp32_vs_reference_golden(golden, tag, backbone):
    g = golden(tag)
    K, seed, gain = int(g["K"]), int(g["seed"]), float(g["residual_gain"])
    damped = gain < 1.0
    net = build(backbone, K, seed, gain)
    net.train()
    x = torch.from_numpy(g["x"]).to(dev())
    pred, rep = net(x)
    assert pred.shape == g["pred"].shape and rep.shape == g["rep"].shape
    gp, gr = torch.from_numpy(g["pred"]), torch.from_numpy(g["rep"])
    e_pred, e_rep = rel_err(pred.detach().cpu(), gp), rel_err(rep.detach().cpu(), gr)
    print(f"{tag}: rel err vs reference fp32: pred {e_pred:.2e} rep {e_rep:.2e}")
    if damped:
        assert e_pred < 1e-3 and e_rep < 1e-3
    else:
        tp, tr = fp64_truth(backbone, K, seed, gain, torch.from_numpy(g["x"]))
        ref_p, ref_r = rel_err(gp, tp), rel_err(gr, tr)
        hip_p, hip_r = rel_err(pred.detach().cpu(), tp), rel_err(rep.detach().cpu(), tr)
        print(f"{tag}: vs fp64 truth: reference fp32 {ref_p:.2e}/{ref_r:.2e}, HIP fp32 {hip_p:.2e}/{hip_r:.2e}")
        # noise floor of the problem itself (round 5: the bound is DERIVED from it, no flat 5e-3 any more): what legitimate fp32
        # evaluations of this very network differ by - a 1-ulp relative perturbation of the input (3 draws), of every weight (3 draws),
        # and the same oracle on one thread (another summation order inside its GEMMs).  Each of these stands for "another correct
        # fp32 implementation"; the HIP path is one more.  Measured in the build container (max norm): 0.4 ... 0.9e-3 (tv 65), 0.5 ... 1.7e-3
        # (stem 65), 0.7 ... 1.9e-3 (tv 97) - the reference's fp32 output itself sits 1.4e-3 / 0.5e-3 / 3.9e-3 from its fp64 evaluation.
        from oracle import css_oracle as O
        xn0 = torch.from_numpy(g["x"])
        sd0 = O.init_state(backbone, K, 256, seed, gain)
        floors = []
        with torch.no_grad():
            for s_ in range(3):
                gen = torch.Generator().manual_seed(s_)
                pn, rn = O.deeplab_forward(sd0, xn0 * (1 + 1e-7 * torch.randn(xn0.shape, generator=gen)), backbone, True, K, 256)
                floors.append((rel_err(pn, gp), rel_err(rn, gr)))
            for s_ in range(3):
                gen = torch.Generator().manual_seed(100 + s_)
                sdn = {k: (v * (1 + 6e-8 * torch.randn(v.shape, generator=gen)) if v.is_floating_point() else v) for k, v in sd0.items()}
                pn, rn = O.deeplab_forward(sdn, xn0, backbone, True, K, 256)
                floors.append((rel_err(pn, gp), rel_err(rn, gr)))
            nt = torch.get_num_threads()
            torch.set_num_threads(1)
            try:
                pn, rn = O.deeplab_forward(sd0, xn0, backbone, True, K, 256)
            finally:
                torch.set_num_threads(nt)
            floors.append((rel_err(pn, gp), rel_err(rn, gr)))
        ulp_p, ulp_r = max(f[0] for f in floors), max(f[1] for f in floors)
        bound_p, bound_r = max(1e-3, 2 * ulp_p), max(1e-3, 2 * ulp_r)
        print(f"{tag}: 1-ulp floor of the fp32 oracle (7 legitimate re-evaluations, max): {ulp_p:.2e}/{ulp_r:.2e}; "
              f"HIP vs reference {e_pred:.2e}/{e_rep:.2e} = {e_pred / 1e-3:.2f}x/{e_rep / 1e-3:.2f}x north_star's 1e-3, "
              f"{e_pred / ulp_p:.2f}x/{e_rep / ulp_r:.2f}x the floor; asserted against max(1e-3, 2 x floor) = {bound_p:.2e}/{bound_r:.2e}")
        assert hip_p < 4 * max(ref_p, ulp_p) and hip_r < 4 * max(ref_r, ulp_r)
        assert e_pred < bound_p and e_rep < bound_r, (e_pred, bound_p, e_rep, bound_r)
    loss = (pred * torch.from_numpy(g["wp"]).to(dev())).sum() + (rep * torch.from_numpy(g["wr"]).to(dev())).sum()
    loss.backward()
    # Gradients: the two fp32 forwards differ by ~1e-5..1e-3, so a few dozen of ~1e5 pre-activations per layer sit on
    # opposite sides of 0 and their ReLU masks differ; n flips of N active elements move a gradient by ~sqrt(n/N) in
    # relative L2 (measured 1.3 % damped / 6 % undamped, cosine 0.9999 / 0.998).  Exact gradient parity is asserted
    # block-wise in test_blocks_gpu.py (5e-5); here the flip-robust metrics must hold.
    named = dict(net.named_parameters())
    a = torch.cat([probe_slice(named[k[6:]].grad).cpu().double() for k in g if k.startswith("grad::")])
    b = torch.cat([torch.from_numpy(g[k]).double() for k in g if k.startswith("grad::")])
    l2 = ((a - b).norm() / b.norm()).item()
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    print(f"{tag}: probe grads rel-L2 {l2:.2e} cosine {cos:.6f}")
    assert (l2 < 3e-2 and cos > 0.9995) if damped else (l2 < 0.15 and cos > 0.99)
    bufs = dict(net.named_buffers())
    for key in g:
        if key.startswith("rm::"):
            assert rel_err(bufs[key[4:] + ".running_mean"].cpu(), torch.from_numpy(g[key])) < 2e-3
        if key.startswith("rv::"):
            assert rel_err(bufs[key[4:] + ".running_var"].cpu(), torch.from_numpy(g[key])) < 2e-3
    net.eval()
    with torch.no_grad():
        pe, re_ = net(x)
    assert rel_err(pe.cpu(), torch.from_numpy(g["pred_eval"])) < 1e-3
    assert rel_err(re_[:, ::16].cpu(), torch.from_numpy(g["rep_eval_sub"])) < 1e-3


def test_network_bf16_close_to_fp32(golden):
    g = golden("net_tv_65_damped")
    net = build("tv", int(g["K"]), int(g["seed"]), float(g["residual_gain"])).set_compute_dtype(torch.bfloat16)
    net.train()
    pred, rep = net(torch.from_numpy(g["x"]).to(dev()))
    assert pred.dtype == torch.bfloat16
    # bf16 activations through 100+ batch-stat BN layers: judged loosely (throughput path; parity is the fp32 path)
    e = rel_err(pred.float().cpu(), torch.from_numpy(g["pred"]))
    print("bf16 pred rel err vs reference fp32:", e)
    assert e < 0.3
    cos = torch.nn.functional.cosine_similarity(pred.float().cpu().flatten(), torch.from_numpy(g["pred"]).flatten(), dim=0)
    print("bf16 pred cosine vs reference fp32:", float(cos))
    assert cos > 0.97
    (pred.float().sum() + rep.float().sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_network_stagewise_vs_oracle(golden):
    """Per-stage activations against the CPU oracle: the error must grow no faster than a 1-ulp perturbation does."""
    from oracle import css_oracle as O
    from css_amd import ops
    g = golden("net_tv_65")
    K, seed = int(g["K"]), int(g["seed"])
    net = build("tv", K, seed)
    net.train()
    sd = O.init_state("tv", K, 256, seed)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        _, _, inter = O.deeplab_forward(sd, x, "tv", True, K, 256, return_intermediates=True)
        h = ops.stage_input(x.to(dev()), torch.float32)
        h = net.resnet_maxpool(net.resnet_bn1(net.resnet_conv1(h), relu=True))
        errs = {"stem": rel_err(h.cpu().permute(0, 3, 1, 2), inter["stem"])}
        for i, layer in enumerate([net.resnet_layer1, net.resnet_layer2, net.resnet_layer3, net.resnet_layer4], 1):
            h = layer(h)
            errs[f"layer{i}"] = rel_err(h.cpu().permute(0, 3, 1, 2), inter[f"layer{i}"])
        a = net.ASPP(h)
        errs["aspp"] = rel_err(a.cpu().permute(0, 3, 1, 2), inter["aspp"])
    print("stagewise rel err:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert errs["stem"] < 2e-6 and errs["layer1"] < 1e-5 and errs["layer2"] < 5e-5 and errs["layer4"] < 2e-3
