"""Checkpoint wire format (mix_label.py:104-112,137-147): the dict MixTrainer saves is what the reference's
torch.optim.SGD / PolyLR / nn.Module.load_state_dict consume, and resuming from it continues the run."""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import dev, rel_err  # noqa: E402


def _trainer(seed, K=21, S=65):
    from css_amd.networks import resnet
    from css_amd.networks.ddp_model import Model_mix
    from css_amd.train_step import MixTrainer
    torch.manual_seed(seed)
    cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [1.0, 1.0], "mix_mode": "none", "device_aug": "identity"}}
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.25).to(dev())
    m.model.train()
    m.ema_model.train()
    return MixTrainer(m, num_classes=K, lr=6.4e-3, total_iter=1000, num_queries=32, num_negatives=64)


def _batch(seed, K=21, S=65, B=2):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 3, S, S, generator=g).to(dev()), torch.randint(-1, K, (B, S, S), generator=g).to(dev()),
            torch.randn(B, 3, S, S, generator=g).to(dev()))


def test_checkpoint_roundtrip_and_reference_format():
    from css_amd import checkpoint as ck
    from css_amd.scheduler.my_lr_scheduler import PolyLR
    a = _trainer(1)
    for i in range(2):
        a.step(*_batch(10 + i))
    sd = ck.state_for_save(a, epoch=4)
    assert set(sd) == {"epoch", "model", "ema_model", "optimizer", "lr_scheduler", "prototypes"} and sd["epoch"] == 5
    # through torch.save / torch.load like the reference
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu", weights_only=False)

    # (1) the optimizer / scheduler entries are real torch state dicts
    cpu_params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in a.model.model.parameters()]
    opt = torch.optim.SGD(cpu_params, lr=6.4e-3, weight_decay=5e-4, momentum=0.9, nesterov=True)
    opt.load_state_dict(sd["optimizer"])
    assert len(opt.state) == len(cpu_params)
    i_big = max(range(len(cpu_params)), key=lambda i: cpu_params[i].numel())
    mb = opt.state[cpu_params[i_big]]["momentum_buffer"]
    assert mb.shape == cpu_params[i_big].shape and float(mb.abs().max()) > 0
    sch = PolyLR(opt, 1000, min_lr=1e-4)
    sch.load_state_dict(sd["lr_scheduler"])
    assert sch.last_epoch == 2 and abs(sch.get_last_lr()[0] - a.lr) < 1e-12
    # model entries: the reference's keys (no 'module.' prefix), logical NCHW shapes
    assert "resnet_layer3.5.conv2.weight" in sd["model"] and tuple(sd["model"]["resnet_conv1.weight"].shape) == (64, 3, 7, 7)
    assert sd["prototypes"].shape == (21, 256)

    # (2) resume into a differently initialised trainer (DDP-style 'module.' keys accepted), state equal
    sd["model"] = {"module." + k: v for k, v in sd["model"].items()}
    b = _trainer(2)
    assert ck.load_checkpoint(sd, b) == 5
    assert b.it == a.it and abs(b.lr - a.lr) < 1e-15
    assert torch.equal(b.flat_p, a.flat_p) and torch.equal(b.flat_ema, a.flat_ema) and torch.equal(b.flat_m, a.flat_m)
    assert torch.equal(b.prototypes, a.prototypes)
    for (k1, v1), (k2, v2) in zip(a.model.ema_model.state_dict().items(), b.model.ema_model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1

    # (3) and the run continues identically (same batch, same sampler stream); the reference restarts Model_mix.step at 0 on resume
    a.model.step = 0
    b.crit_contrast._calls = a.crit_contrast._calls     # the sampler's stream position is not checkpointed (nor is the reference's RNG)
    x = _batch(20)
    torch.manual_seed(77)
    oa = a.step(*x)
    torch.manual_seed(77)
    ob = b.step(*x)
    for k in ("sup", "unsup"):
        assert abs(float(oa[k]) - float(ob[k])) <= 1e-6 * max(1.0, abs(float(oa[k]))), k
    assert torch.equal(oa["pseudo"], ob["pseudo"])
    assert abs(float(oa["contrast"]) - float(ob["contrast"])) < 1e-5 * abs(float(oa["contrast"]))
    assert rel_err(b.flat_p, a.flat_p) < 1e-4      # fp32 atomics in wgrad: order-dependent last bits
