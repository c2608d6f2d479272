"""Two full training iterations (teacher fwd x2, student fwd/bwd x2, three losses, SGD+EMA) on the GPU against the trace
captured from the reference's Model_mix / Contrast_Loss / SGD (tests/golden/train_trace.npz), sampler draws injected."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev, rel_err  # noqa: E402


def T(a):
    return torch.from_numpy(np.asarray(a))


def probe_slice(t):
    return t.detach().flatten()[:: max(1, t.numel() // 2048)][:2048]


def make(seed, K=21, S=65, gain=1.0):
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from oracle import css_oracle as O
    cfg = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none", "device_aug": "identity"}}
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.5)
    sd = O.init_state("tv", K, 256, seed, gain)
    m.model.load_state_dict(sd, strict=True)
    m.ema_model.load_state_dict(sd, strict=True)
    return m.to(dev()).train()


@pytest.mark.parametrize("tag", ["train_trace_damped", "train_trace"])
def test_two_steps_vs_reference_trace(golden, tag):
    from oracle import css_oracle as O
    from css_amd.train_step import MixTrainer
    g = golden(tag)
    seed, gain = int(g["seed"]), float(g["residual_gain"])
    damped = gain < 1.0
    m = make(seed, gain=gain)
    tr = MixTrainer(m, 21, lr=6.4e-3, total_iter=100, min_lr=1e-4, num_queries=64, num_negatives=128, strong_threshold=0.8,
                    weak_threshold=0.0, un_threshold=0.97)
    st = O.MixState("tv", 21, 256, seed, gain)     # CPU oracle run side by side to line the recorded draws up with present classes
    probes = ["resnet_conv1.weight", "resnet_layer3.10.conv2.weight", "classifier.3.weight", "representation.3.bias"]
    for it in range(2):
        l_img, l_lab, u_img = T(g[f"{it}::l_img"]), T(g[f"{it}::l_lab"]).long(), T(g[f"{it}::u_img"])
        args = dict(lr=O.poly_lr(6.4e-3, it, 100), temp_model=0.5, strong_threshold=0.8, weak_threshold=0.0, un_threshold=0.97,
                    num_queries=64, num_negatives=128)
        rec = {}
        torch.manual_seed(it)
        np.random.seed(it)
        O.train_step_mix(copy.deepcopy(st), l_img, l_lab, u_img, record=rec, **args)
        if it == 0:
            # step 0: the draws recorded from the REFERENCE run (golden), lined up with the classes that have hard pixels
            anchors, negs, j = [], [], 0
            for hn in rec["hard_num"]:
                if hn > 0:
                    anchors.append(g[f"{it}::anchor{j}"].astype(np.int64))
                    negs.append(g[f"{it}::negative{j}"].astype(np.int64))
                    j += 1
                else:
                    anchors.append(None)
                    negs.append(None)
            assert j == int(g[f"{it}::n_anchor"])
            inj = dict(anchor=anchors, negative=negs)
        else:
            # step 1 sits behind one SGD step whose gradient carries ReLU-flip noise; the host CPU of the GPU box is also
            # not the CPU the golden trace was made on, so class presence / list lengths may differ from the golden run.
            # Use the oracle's own draws (made on this box) for both sides and compare statistically.
            inj = dict(anchor=rec["anchor"], negative=rec["negative"])
        ro = O.train_step_mix(st, l_img, l_lab, u_img, injected=inj, **args)
        r = tr.step(l_img.to(dev()), l_lab.to(dev()), u_img.to(dev()), ramp=1.0, _injected=inj)
        fails = []
        for key, gk, tol in (("sup", "sup", 2e-3), ("unsup", "unsup", 3e-2), ("contrast", "con", 2e-3)):
            ref = float(g[f"{it}::{gk}"]) if it == 0 else ro[key if key != "contrast" else "contrast"]
            tol = tol if it == 0 else 0.35   # sanity only: lr*|grad| >> |param| on a random-init net, see module docstring
            print(f"{tag} step {it} {key}: hip {r[key].item():.6f} reference {ref:.6f}")
            if not abs(r[key].item() - ref) < tol * max(1.0, abs(ref)):
                fails.append((key, r[key].item(), ref))
        assert not fails, fails
        ref_lab = T(g[f"{it}::ulab"]).long() if it == 0 else ro["pseudo"]
        mism = (r["pseudo"].cpu() != ref_lab).float().mean().item()
        print(f"{tag} step {it}: pseudo-label mismatch fraction {mism:.2e}")
        assert mism < (1e-3 if it == 0 else 0.2)
        ref_proto = T(g[f"{it}::protos"]) if it == 0 else st.prototypes
        e = rel_err(tr.prototypes.cpu(), ref_proto)
        print(f"{tag} step {it}: prototypes rel err {e:.2e}")
        assert e < (2e-3 if it == 0 else 0.2)
        sdm, sde = m.model.state_dict(), m.ema_model.state_dict()
        for p in probes:
            rs = T(g[f"{it}::student::{p}"]) if it == 0 else probe_slice(st.student[p])
            rt = T(g[f"{it}::teacher::{p}"]) if it == 0 else probe_slice(st.teacher[p])
            es, et = rel_err(probe_slice(sdm[p]).cpu(), rs), rel_err(probe_slice(sde[p]).cpu(), rt)
            print(f"{tag} step {it} {p}: student {es:.2e} teacher {et:.2e}")
            # parameters after the update are dominated by lr*grad (random init: |grad| ~ 1e4); the grad carries the ReLU-flip
            # noise quantified in test_network_gpu.py (~1 % damped, ~6 % undamped)
            lim = (3e-2 if damped else 0.12) * (1 if it == 0 else 3)
            if it == 0:       # step 1 parameters are printed for information only (see module docstring)
                assert es < lim and et < lim
        if it == 0:
            assert rel_err(sde["resnet_bn1.running_mean"].cpu(), T(g[f"{it}::teacher_rm::resnet_bn1"])) < 1e-3
    assert all(torch.isfinite(p).all() for p in m.model.parameters())


def test_sgd_ema_kernel_matches_reference_optimizer(golden):
    """css_sgd_ema against the trajectory of torch.optim.SGD(nesterov) + Model_mix.ema_update captured from the reference."""
    from css_amd._lib import call, dev_stream
    g = golden("schedules")
    p = torch.tensor([1.0, -2.0, 0.5, 0.0], device=dev())
    buf, ema = torch.zeros(4, device=dev()), torch.zeros(4, device=dev())
    d, st = dev_stream(p)
    grads = T(g["sgd_grads"])
    for i in range(5):
        gr = torch.cat([grads[i], torch.zeros(1)]).to(dev())
        call("css_sgd_ema", p, gr, buf, ema, 4, 0.01, 0.9, 5e-4, int(i == 0), 0.5, 1.0, None, d, st)
        assert rel_err(p[:3].cpu(), T(g["sgd_traj"])[i]) < 1e-6
    e = torch.zeros(5, device=dev())
    one = torch.ones(5, device=dev())
    for step in range(150):
        decay = min(1 - 1 / (step + 1), 0.99)
        call("css_ema", e, one, 5, float(decay), d, st)
        assert abs(e[4].item() - float(g["ema"][step])) < 1e-6
