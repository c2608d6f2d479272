"""Two ranks on ONE MI355X (gloo backend moving device tensors through the host): the data-parallel code paths --
SyncBN statistics forward/backward, prototype-sum all-reduce, flat-gradient all-reduce + fused SGD/EMA -- against a
single-process run on the concatenated batch.  With SyncBN, 2 ranks x B images == 1 rank x 2B images for the network
itself (same batch statistics), so student logits, the supervised loss (mean of equal-sized means) and the gradient
(mean over ranks) must agree."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_util import dev  # noqa: E402

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from css_amd import ops
from css_amd.networks import resnet
from css_amd.networks.deeplabv3.deeplabv3 import DeepLabv3Plus_with_rep
from css_amd.loss.loss import CrossEntropyLoss
from oracle import css_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
K, S, seed = 21, 65, 5
sd = O.init_state("tv", K, 256, seed, 0.25)


def run(mode):                               # mode: "fp32" | "bf16" | "bf16same" (one process serves all three: start-up and imports paid once)
    net = DeepLabv3Plus_with_rep(resnet.resnet101_tv(), dilate_scale=8, num_classes=K)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    if mode != "fp32":
        net.set_compute_dtype(torch.bfloat16)    # bf16: batch-norm statistics come from the conv epilogue (slab rows) before the all-reduce
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, S, S, generator=g)
    lab = torch.randint(0, K, (4, S, S), generator=g)
    if mode == "bf16same":
        x, lab = x[:2], lab[:2]                  # every rank (and the single process) holds the SAME two images: a deterministic comparison
    elif world > 1:
        x, lab = x[2 * rank: 2 * rank + 2], lab[2 * rank: 2 * rank + 2]
    pred, rep = net(x.to(dev))
    large = ops.bilinear(pred.permute(0, 2, 3, 1).contiguous(), S, S, torch.float32).permute(0, 3, 1, 2)
    loss = CrossEntropyLoss(-1)(large, lab.to(dev)) + rep.float().pow(2).mean()
    loss.backward()
    grads = torch.cat([p.grad.flatten() for p in net.parameters()])
    if world > 1:
        dist.all_reduce(grads)
        grads /= world
        l = loss.detach().clone()
        dist.all_reduce(l)
        loss = l / world
    probe = grads[:: grads.numel() // 4096][:4096].cpu()
    # per-parameter samples along the backward chain (16 parameters, evenly spread; <= 512 elements each)
    layers, o = {}, 0
    plist = list(net.named_parameters())
    for i, (n_, p_) in enumerate(plist):
        if i %% max(len(plist) // 16, 1) == 0:
            gsl = grads[o:o + p_.numel()]
            layers[n_] = gsl[:: max(gsl.numel() // 509, 1) | 1][:512].cpu().tolist()      # (odd stride: no aliasing with the [Cout][R][S][Cin] layout)
        o += p_.numel()
    tail = grads[-(grads.numel() // 4):]                       # ASPP + decoder heads: the layers closest to the loss
    tail = tail[:: tail.numel() // 4096][:4096].cpu()
    out = dict(loss=float(loss), pred=pred.detach().float().cpu().flatten()[::97].tolist(), grad=probe.tolist(), grad_tail=tail.tolist(),
               rm=net.resnet_bn1.running_mean.cpu().tolist(), layers=layers)
    if rank == 0:
        json.dump(out, open(sys.argv[1] + mode + ".json", "w"))
    del net
    torch.cuda.empty_cache()


for mode in os.environ["CSS_TEST_MODES"].split(","):
    run(mode)
if world > 1:
    dist.destroy_process_group()
'''

_W12 = {}      # mode -> (single-process result, two-rank result): ONE single process and ONE pair of ranks serve the three tests below (round 6)


def _world_1_and_2(tmp_path_factory, mode):
    import json
    if not _W12:
        d = tmp_path_factory.mktemp("w12")
        modes = "fp32,bf16,bf16same"
        for world in (1, 2):
            procs = []
            for r in range(world):
                env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", CSS_TEST_MODES=modes)
                procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT, str(d / f"w{world}_")], env=env))
            for p in procs:
                assert p.wait(timeout=900) == 0
        for m in modes.split(","):
            _W12[m] = (json.load(open(d / f"w1_{m}.json")), json.load(open(d / f"w2_{m}.json")))
    return _W12[mode]


def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path_factory):
    import torch
    r1, r2 = _world_1_and_2(tmp_path_factory, "fp32")
    assert abs(r1["loss"] - r2["loss"]) < 1e-4 * abs(r1["loss"])
    p1, p2 = torch.tensor(r1["pred"]), torch.tensor(r2["pred"][: len(r1["pred"])])
    # rank 0 of the 2-rank run holds the first two images: compare against the first half of the single-process output
    n = len(r2["pred"])
    # (rank 0 of the 2-rank run holds the first two images; reported, the assertions are on loss, gradient and running statistics)
    print("world1 vs world2: max logit difference", ((p1[:n] - torch.tensor(r2["pred"])).abs().max() / p1.abs().max()).item())
    g1, g2 = torch.tensor(r1["grad"]), torch.tensor(r2["grad"])
    cos = torch.nn.functional.cosine_similarity(g1, g2, dim=0).item()
    rel = ((g1 - g2).norm() / g1.norm()).item()
    print("world1 vs world2: grad cosine", cos, "rel-L2", rel)
    assert cos > 0.999 and rel < 3e-2          # ReLU-flip noise level, see test_network_gpu.py
    rm1, rm2 = torch.tensor(r1["rm"]), torch.tensor(r2["rm"])
    assert ((rm1 - rm2).abs().max() / rm1.abs().max()).item() < 1e-4     # SyncBN running statistics = global batch statistics


def test_two_ranks_bf16_fused_statistics_path(tmp_path_factory):
    """Same experiment on the bf16 throughput path: SyncBN there all-reduces the (sum, sum of squares) that stage 2 extracts from
    the convolution epilogue's slab rows (css_bn_reduce_finalize_slabs with sums_out).  bf16 tolerance."""
    import torch
    r1, r2 = _world_1_and_2(tmp_path_factory, "bf16")
    assert abs(r1["loss"] - r2["loss"]) < 2e-2 * abs(r1["loss"])
    n = len(r2["pred"])
    p1, p2 = torch.tensor(r1["pred"][:n]), torch.tensor(r2["pred"])
    assert torch.nn.functional.cosine_similarity(p1, p2, dim=0) > 0.99
    g1, g2 = torch.tensor(r1["grad"]), torch.tensor(r2["grad"])
    cos_all = float(torch.nn.functional.cosine_similarity(g1, g2, dim=0))
    t1, t2 = torch.tensor(r1["grad_tail"]), torch.tensor(r2["grad_tail"])
    cos_tail = float(torch.nn.functional.cosine_similarity(t1, t2, dim=0))
    print("bf16 world1 vs world2 gradient cosine: all parameters", cos_all, "ASPP + heads", cos_tail)
    # bf16 through 100+ batch-stat layers on 4 random images: the backward chain is chaotic at its far end.  Measured per parameter
    # (scripts/bf16_grad_layers.py, round 2): the layers next to the loss agree to 0.999 between the two runs and with fp32; the backbone
    # gradients of EITHER run have cosine ~0.6 with the fp32 gradients, 0.65-0.82 with each other, and norms within 0.74-1.0 of each other.
    # The old all-parameter probe (one element in 14500) holds a single stem-weight element that outweighs everything else in it - it read
    # 0.36, 0.76 and 0.89 on three builds that pass every parity test - and is reported only.  Asserted: the layers next to the loss, and per
    # parameter along the chain a positive correlation and comparable size (a wrong count or a missed all-reduce in SyncBN's backward shows as
    # a factor of 2 or as no correlation at all).  The exact form of this check is test_syncbn_with_unequal_pixel_counts_per_rank (op level,
    # against torch-CPU) and the fp32 test above (cosine > 0.999).
    assert torch.isfinite(g2).all() and cos_tail > 0.9
    cs = []
    for name in r1["layers"]:
        a_, b_ = torch.tensor(r1["layers"][name]).double(), torch.tensor(r2["layers"][name]).double()
        if float(a_.norm()) == 0.0 and float(b_.norm()) == 0.0:
            continue                                    # (e.g. the padding-only taps of a dilated ASPP convolution on a 9x9 map)
        c_, ratio = float(torch.nn.functional.cosine_similarity(a_, b_, dim=0)), float(b_.norm() / a_.norm())
        print(f"  {name:50s} cosine {c_:.3f} norm ratio {ratio:.3f}")
        assert c_ > 0.3 and 0.5 < ratio < 2.0, (name, c_, ratio)
        cs.append(c_)
    assert sum(cs) / len(cs) > 0.55, cs
    rm1, rm2 = torch.tensor(r1["rm"]), torch.tensor(r2["rm"])
    assert ((rm1 - rm2).abs().max() / rm1.abs().max()).item() < 2e-2


def test_two_ranks_bf16_identical_halves_equal_one_rank(tmp_path_factory):
    """The tight gradient gate of the bf16 path (advisor, round 2): when both ranks hold the SAME two images, SyncBN's global statistics are
    exactly the single process's (sums and counts double), every rank computes the same gradient and their mean is that gradient - so
    world 2 must reproduce world 1 along the WHOLE backward chain up to the order of fp32 atomic adds, although the chain is chaotic
    between different data (test above).  A wrong count, a missed all-reduce or a scaling error anywhere in SyncBN's forward or backward
    shows here as a factor, not as noise."""
    import torch
    r1, r2 = _world_1_and_2(tmp_path_factory, "bf16same")
    assert abs(r1["loss"] - r2["loss"]) < 1e-5 * abs(r1["loss"])
    p1, p2 = torch.tensor(r1["pred"]), torch.tensor(r2["pred"])
    assert float((p1 - p2).abs().max()) <= 1e-3 * float(p1.abs().max())
    g1, g2 = torch.tensor(r1["grad"]).double(), torch.tensor(r2["grad"]).double()
    cos, rel = float(torch.nn.functional.cosine_similarity(g1, g2, dim=0)), float((g1 - g2).norm() / g1.norm())
    print("bf16, identical halves: world1 vs world2 gradient cosine", cos, "rel-L2", rel)
    assert cos > 0.9999 and rel < 1e-2
    for name in r1["layers"]:
        a_, b_ = torch.tensor(r1["layers"][name]).double(), torch.tensor(r2["layers"][name]).double()
        if float(a_.norm()) == 0.0 and float(b_.norm()) == 0.0:
            continue
        c_, ratio = float(torch.nn.functional.cosine_similarity(a_, b_, dim=0)), float(b_.norm() / a_.norm())
        print(f"  {name:50s} cosine {c_:.6f} norm ratio {ratio:.6f}")
        assert c_ > 0.999 and 0.99 < ratio < 1.01, (name, c_, ratio)
    rm1, rm2 = torch.tensor(r1["rm"]), torch.tensor(r2["rm"])
    assert ((rm1 - rm2).abs().max() / rm1.abs().max()).item() < 1e-5


TRAINER_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import numpy as np
from css_amd.networks import resnet
from css_amd.networks.ddp_model import Model_mix
from css_amd.train_step import MixTrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=rank, world_size=world)
K, S = 21, 65
cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [1.0, 1.0], "mix_mode": "cutmix", "device_aug": "identity"}}
for conf in os.environ["CSS_TEST_CONFIGS"].split(","):       # "bucket MB:steps" - one pair of processes serves every configuration (round 6)
    bucket_mb, steps = conf.split(":")
    os.environ["CSS_GRAD_BUCKET_MB"] = bucket_mb            # (read by MixTrainer.__init__)
    torch.manual_seed(11)                                   # same initial weights on every rank (DDP broadcasts rank 0's)
    m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.25).to(dev)
    m.model.train(); m.ema_model.train()
    m.set_compute_dtype(torch.bfloat16)
    tr = MixTrainer(m, num_classes=K, lr=6.4e-3, total_iter=100, num_queries=32, num_negatives=64)
    g = torch.Generator().manual_seed(100 + rank)           # different data per rank
    losses = []
    np.random.seed(5)                                        # cutmix boxes (host draws)
    for it in range(int(steps)):
        l = torch.randn(2, 3, S, S, generator=g).to(dev); y = torch.randint(-1, K, (2, S, S), generator=g).to(dev)
        u = torch.randn(2, 3, S, S, generator=g).to(dev)
        out = tr.step(l, y, u)
        losses.append([float(out["sup"]), float(out["contrast"])])
    tr.finish()
    probe = tr.flat_p[:: tr.flat_p.numel() // 4096][:4096].double().cpu()
    ema = tr.flat_ema[:: tr.flat_ema.numel() // 4096][:4096].double().cpu()
    proto = tr.prototypes.double().cpu()
    rm = m.model.resnet_bn1.running_mean.double().cpu()
    nb = [len(tr._buckets), sum(len(r) for r, _ in tr._buckets), len(tr._ready_order)] if tr._buckets else [0, 0, 0]
    json.dump(dict(losses=losses, p=probe.tolist(), ema=ema.tolist(), proto=proto.flatten().tolist(), rm=rm.tolist(), buckets=nb),
              open(sys.argv[1] + bucket_mb + "_" + str(rank), "w"))
    del tr, m
    torch.cuda.empty_cache()
dist.destroy_process_group()
'''

_PAIRS = {}      # bucket MB -> (rank 0 result, rank 1 result): one pair of processes runs the three configurations of the two tests below


def _trainer_pairs(tmp_path_factory):
    import json
    if not _PAIRS:
        d = tmp_path_factory.mktemp("trainer_pairs")
        out = str(d / "r")
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", CSS_TEST_CONFIGS="8:2,0:2,48:3")
            procs.append(subprocess.Popen([sys.executable, "-c", TRAINER_WORKER % ROOT, out], env=env))
        for p in procs:
            assert p.wait(timeout=900) == 0
        for k in ("8", "0", "48"):
            _PAIRS[k] = (json.load(open(out + k + "_0")), json.load(open(out + k + "_1")))
    return _PAIRS


def test_bucketed_gradient_all_reduce_equals_the_single_collective(tmp_path_factory):
    """The gradient all-reduce overlapped with backward in buckets (train_step.MixTrainer._backward_and_reduce; DDP's buckets at
    mix_label.py:77) against ONE all-reduce after backward (CSS_GRAD_BUCKET_MB=0): two steps (the first records the readiness order,
    the second runs bucketed), replicas bit-identical in both modes, and the two modes equal up to run-to-run noise (the order of
    the remaining fp32 atomic adds - class sums, loss footprints - which a random-init bf16 network amplifies step by step: 3e-3 to
    8e-3 after two steps on different boxes, 2e-2 after three - hence two steps and a 3e-2 bound here; a bucket that missed a range
    or reduced one twice breaks the replica equality (the gradient of that range is then no longer the same sum on both ranks) or
    shows up as O(1))."""
    import torch
    pairs = _trainer_pairs(tmp_path_factory)
    (a8, b8), (a0, b0) = pairs["8"], pairs["0"]
    print("buckets / runs / parameters reported:", a8["buckets"])
    assert a8["buckets"][0] >= 10 and a8["buckets"][2] > 300 and a0["buckets"] == [0, 0, 0]
    assert a8["buckets"][1] <= a8["buckets"][0] + 4              # backward runs the layers in reverse: a bucket is one or two runs
    for a, b in ((a8, b8), (a0, b0)):
        assert torch.equal(torch.tensor(a["p"]), torch.tensor(b["p"])) and torch.equal(torch.tensor(a["ema"]), torch.tensor(b["ema"]))
    p8, p0 = torch.tensor(a8["p"]), torch.tensor(a0["p"])
    err = ((p8 - p0).norm() / p0.norm()).item()
    print("bucketed vs single all-reduce after 2 steps: rel-L2 of the parameters", err, a8["losses"], a0["losses"])
    assert err < 3e-2


def test_two_rank_trainer_keeps_replicas_in_sync(tmp_path_factory):
    """MixTrainer.step on two ranks (bf16, different data per rank): SyncBN statistics, the prototype class sums and the flat
    gradient are all-reduced, so after three steps both replicas hold the same parameters, EMA teacher, BN running statistics and
    prototypes (for the classes both ranks see) - the data-parallel contract of mix_label.py:76-77."""
    import torch
    a, b = _trainer_pairs(tmp_path_factory)["48"]
    # (the unsupervised term is NaN-valued with zero gradient when no pseudo-label is confident, like the reference: SURVEY L2)
    assert all(v == v and abs(v) < 1e3 for l in a["losses"] + b["losses"] for v in l), (a["losses"], b["losses"])
    pa, pb = torch.tensor(a["p"]), torch.tensor(b["p"])
    assert torch.equal(pa, pb), float((pa - pb).abs().max())                         # same summed gradient -> same update
    assert torch.equal(torch.tensor(a["ema"]), torch.tensor(b["ema"]))
    assert torch.equal(torch.tensor(a["rm"]), torch.tensor(b["rm"]))                 # SyncBN: global statistics on both ranks


VIOLATE_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from css_amd.networks import resnet
from css_amd.networks.ddp_model import Model_mix
from css_amd.train_step import MixTrainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=rank, world_size=world)
K, S = 21, 65
torch.manual_seed(11)
cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [1.0, 1.0], "mix_mode": "cutmix", "device_aug": "identity"}}
m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.25).to(dev)
m.model.train(); m.ema_model.train()
m.set_compute_dtype(torch.bfloat16)
tr = MixTrainer(m, num_classes=K, lr=6.4e-3, total_iter=100, num_queries=32, num_negatives=64)
g = torch.Generator().manual_seed(100 + rank)
import numpy as np
np.random.seed(5)
raised, its, call_no = [], [], 0
while call_no < 7:
    l = torch.randn(2, 3, S, S, generator=g).to(dev); y = torch.randint(-1, K, (2, S, S), generator=g).to(dev)
    u = torch.randn(2, 3, S, S, generator=g).to(dev)
    poked = None
    if call_no == 2 and rank == 1:           # ONE rank's gradient reports stop fitting the recorded plan: a span waits for a report that never comes
        poked = next(iter(tr._span_reports))
        tr._span_reports[poked] += 1
    try:
        out = tr.step(l, y, u)
        if rank == 0:
            float(out["sup"])                # rank 0's host waits for the device every step; rank 1's host runs ahead as far as it can
    except RuntimeError as e:
        raised.append([call_no, tr.it, m.step, "skipped on every rank" in str(e), "1 step(s)" in str(e)])
    if poked is not None:
        tr._span_reports[poked] -= 1
    its.append(tr.it)
    call_no += 1
tr.finish()
torch.cuda.synchronize()
probe = tr.flat_p[:: tr.flat_p.numel() // 4096][:4096].double().cpu()
ema = tr.flat_ema[:: tr.flat_ema.numel() // 4096][:4096].double().cpu()
json.dump(dict(raised=raised, its=its, p=probe.tolist(), ema=ema.tolist(), it=tr.it, step=m.step, pool=len(tr._pinned_pool),
               buckets=len(tr._buckets) if tr._buckets else 0), open(sys.argv[1] + str(rank), "w"))
dist.destroy_process_group()
'''


def test_invalid_step_is_discovered_at_the_same_step_on_both_ranks(tmp_path):
    """the agreed verdict of step k is read at the start of step k + MixTrainer.VERDICT_LAG on EVERY rank - not whenever a
    rank's host happens to see the copy.  Two ranks on one GPU (gloo), rank 0's host blocked on the device every step and rank 1's running
    ahead; rank 1's plan is violated at call 2.  Both ranks must raise at call 4 with the same counters, the callers continue (as SKIP_WORKER
    does): both re-record their bucket plan in the same call, issue the same collective sequence (no hang) and end with bit-identical weights
    and teacher; exactly one step is rolled back; every pinned verdict word is back in the pool."""
    import json
    import torch
    out = str(tmp_path / "v_")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", CSS_GRAD_BUCKET_MB="8")
        procs.append(subprocess.Popen([sys.executable, "-c", VIOLATE_WORKER % ROOT, out], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    a, b = json.load(open(out + "0")), json.load(open(out + "1"))
    print(a["raised"], b["raised"], a["its"], b["its"])
    assert a["raised"] == b["raised"] and len(a["raised"]) == 1
    call_no, it_after, step_after, msg_ok, one_step = a["raised"][0]
    assert call_no == 2 + 2 and msg_ok and one_step               # VERDICT_LAG = 2; step 3 (queued behind the invalid one) was valid
    assert it_after == 3 and step_after == 3                      # calls 0, 1, 3 were applied; 2 was skipped on the device
    assert a["its"] == b["its"] and a["it"] == b["it"] == 5 and a["step"] == b["step"] == 5
    assert a["buckets"] == b["buckets"] and a["buckets"] >= 10    # re-recorded (call 5) and bucketed again (call 6) on both ranks
    assert torch.equal(torch.tensor(a["p"]), torch.tensor(b["p"])) and torch.equal(torch.tensor(a["ema"]), torch.tensor(b["ema"]))
    assert a["pool"] >= 1 and b["pool"] >= 1


RCCL_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from css_amd.networks import resnet
from css_amd.networks.ddp_model import Model_mix
from css_amd.train_step import MixTrainer
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
if os.environ.get("CSS_FORCE_COLLECTIVES") == "1":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # "nccl" IS RCCL on ROCm
K, S = 21, 65
torch.manual_seed(11)
cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [1.0, 1.0], "mix_mode": "cutmix", "device_aug": "identity"}}
m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.25).to(dev)
m.model.train(); m.ema_model.train()
m.set_compute_dtype(torch.bfloat16)
tr = MixTrainer(m, num_classes=K, lr=6.4e-3, total_iter=100, num_queries=32, num_negatives=64)
g = torch.Generator().manual_seed(100)
import numpy as np
np.random.seed(3)
losses = []
for it in range(2):
    l = torch.randn(2, 3, S, S, generator=g).to(dev); y = torch.randint(-1, K, (2, S, S), generator=g).to(dev)
    u = torch.randn(2, 3, S, S, generator=g).to(dev)
    out = tr.step(l, y, u)
    losses.append([float(out["sup"]), float(out["contrast"])])
torch.cuda.synchronize()
probe = tr.flat_p[:: tr.flat_p.numel() // 4096][:4096].double().cpu()
json.dump(dict(losses=losses, p=probe.tolist(), proto=tr.prototypes.double().cpu().flatten().tolist(),
               rm=m.model.resnet_bn1.running_mean.double().cpu().tolist()), open(sys.argv[1], "w"))
if dist.is_initialized():
    dist.destroy_process_group()
'''


def test_rccl_collectives_on_one_rank_change_nothing(tmp_path):
    """The data-parallel exchanges through RCCL itself (backend "nccl") on the one GPU a test box has: a 1-rank group with
    CSS_FORCE_COLLECTIVES=1 sends every SyncBN statistics tensor (fp64, forward and backward), the prototype sums (fp64) and the
    flat gradient (fp32, 238 MB) through ncclAllReduce on RCCL's stream and back.  Sum over one rank is the identity and every reduction of the
    step is ordered (round 4), so the two training steps must reproduce the no-group run to rounding - which also pins the stream hand-over
    between the compute stream and RCCL's (a missing wait shows up as garbage statistics).  Third run: the same with CSS_SYNCBN=peer - the
    SyncBN statistics through the peer-exchange kernel in loop-back (css_amd/peer.py, world 1: publish, wait, sum and finalise through the very
    launch a multi-GPU node would use) instead of RCCL."""
    import json
    import torch
    outs = []
    for force, mode in (("0", "rccl"), ("1", "rccl"), ("1", "peer")):
        out = str(tmp_path / f"r{force}{mode}.json")
        env = dict(os.environ, CSS_FORCE_COLLECTIVES=force, CSS_SYNCBN=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK="0", WORLD_SIZE="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.Popen([sys.executable, "-c", RCCL_WORKER % ROOT, out], env=env)
        assert p.wait(timeout=900) == 0
        outs.append(json.load(open(out)))
    a = outs[0]
    for b in outs[1:]:
        for step in (0, 1):
            for x, y in zip(a["losses"][step], b["losses"][step]):
                assert abs(x - y) <= 1e-5 * max(1.0, abs(x)), (a["losses"], b["losses"])
        pa, pb = torch.tensor(a["p"]), torch.tensor(b["p"])
        assert ((pa - pb).norm() / pa.norm()).item() < 1e-5
        ra, rb = torch.tensor(a["rm"]), torch.tensor(b["rm"])
        assert ((ra - rb).abs().max() / ra.abs().max()).item() < 1e-5
        qa, qb = torch.tensor(a["proto"]), torch.tensor(b["proto"])
        assert ((qa - qb).norm() / qa.norm().clamp_min(1e-12)).item() < 1e-4


SKIP_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from css_amd import peer
from css_amd.networks import resnet
from css_amd.networks.ddp_model import Model_mix
from css_amd.train_step import MixTrainer
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
K, S = 21, 65
torch.manual_seed(11)
cfg = {"Dataset": {"crop_size": [S, S], "scale_size": [1.0, 1.0], "mix_mode": "cutmix", "device_aug": "identity"}}
m = Model_mix(resnet.resnet101_tv(), num_classes=K, output_dim=256, config=cfg, temp=0.25).to(dev)
m.model.train(); m.ema_model.train()
m.set_compute_dtype(torch.bfloat16)
tr = MixTrainer(m, num_classes=K, lr=6.4e-3, total_iter=100, num_queries=32, num_negatives=64)
g = torch.Generator().manual_seed(100)
import numpy as np
np.random.seed(3)
def batch():
    return (torch.randn(2, 3, S, S, generator=g).to(dev), torch.randint(-1, K, (2, S, S), generator=g).to(dev), torch.randn(2, 3, S, S, generator=g).to(dev))
tr.step(*batch()); tr.step(*batch())
tr.finish()
ex = peer.exchange(dev)
assert ex is not None and ex.mem_kind in ("fine-grained", "uncached")
p0, m0, e0 = tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_ema.clone()
it0, step0 = tr.it, m.step
ex.status.fill_(77)                     # what the exchange kernel leaves behind when it gave up waiting for a peer
tr.step(*batch())                       # this step must not reach weights, momentum or teacher
torch.cuda.synchronize()
same = bool(torch.equal(tr.flat_p, p0) and torch.equal(tr.flat_m, m0) and torch.equal(tr.flat_ema, e0))
raised = ""
try:
    tr.finish()
except RuntimeError as e:
    raised = str(e)
res = dict(same=same, raised=raised, it_back=(tr.it == it0), step_back=(m.step == step0), status=int(ex.status.item()), kind=ex.mem_kind)
tr.step(*batch())                       # the trainer goes on (the caller decided to): a valid step moves the weights again
tr.finish()
res["moved"] = not bool(torch.equal(tr.flat_p, p0))
json.dump(res, open(sys.argv[1], "w"))
dist.destroy_process_group()
'''


def test_peer_timeout_verdict_skips_the_step_on_the_device(tmp_path):
    """a SyncBN peer exchange that gave up waiting leaves its number in the status word; the trainer ORs it into the
    verdict every rank agrees on (one MAX all-reduce with the bucket flag), css_sgd_ema skips that step on the device - weights, momentum and
    the EMA teacher bit-identical to before - and the host learns about it late (finish(): blocking; step(): a poll) with the iteration and
    EMA-step counters rolled back.  Also: the exchange buffer is fine-grained memory of the library's own (css_peer_alloc)."""
    import json
    out = str(tmp_path / "skip.json")
    env = dict(os.environ, CSS_FORCE_COLLECTIVES="1", CSS_SYNCBN="peer", MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, "-c", SKIP_WORKER % ROOT, out], env=env)
    assert p.wait(timeout=900) == 0
    r = json.load(open(out))
    print(r)
    assert r["same"] and r["it_back"] and r["step_back"] and r["moved"] and r["status"] == 0
    assert "peer exchange" in r["raised"] and "skipped on every rank" in r["raised"]


def test_peer_exchange_kernel_plays_three_ranks_in_one_process():
    """css_amd/csrc/peer.hip with W = 3 exchange buffers in ONE process: every exchange is played as publish (phase 1) for each rank, then
    wait + sum (phase 2) for each rank - the slot ring, the flags, the rank-ordered sum and the fused train-mode finalize are the multi-GPU
    code path minus the wire.  Ten exchanges (the ring of four slots wraps twice); then a peer that never publishes: the waiting rank gives up
    after its timeout, records the exchange number in its status word and completes - it never hangs the GPU.
    Reference: nn.SyncBatchNorm's exchange at /root/reference/mix_label.py:76."""
    import torch
    from css_amd._lib import call, dev_stream, query
    W, G, C = 3, 2, 256
    n = G * 2 * C + G
    slot = n + 6                                        # (any slot size >= n)
    nbytes = query("css_peer_buffer_bytes", slot)
    bufs = [torch.zeros(nbytes, dtype=torch.uint8, device=dev()) for _ in range(W)]
    bases = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device=dev())
    status = [torch.zeros(1, dtype=torch.int32, device=dev()) for _ in range(W)]
    d, st = dev_stream(bufs[0])
    g = torch.Generator().manual_seed(4)
    gamma, beta = torch.rand(C, generator=g).to(dev()) + 0.5, torch.randn(C, generator=g).to(dev())
    BIG = 10 ** 9
    for seq in range(1, 11):
        local = []
        for r in range(W):
            x = torch.randn(G, 2, C, generator=g, dtype=torch.float64)
            x[:, 1] = x[:, 1].abs() * 50 + 30.0                                       # sums of squares: positive, var > 0
            cnt = torch.tensor([100.0 + r, 90.0 + 2 * r], dtype=torch.float64)
            local.append(torch.cat([x.flatten(), cnt]).to(dev()))
        want = local[0].clone()
        for r in range(1, W):
            want = want + local[r]                                                  # rank order, fp64
        if seq % 2:       # backward form: plain sums
            outs = [torch.empty(n, dtype=torch.float64, device=dev()) for _ in range(W)]
            for r in range(W):
                call("css_bn_peer_gather", bases, W, r, seq, slot, local[r], n, outs[r], status[r], BIG, 1, d, st)
            for r in range(W):
                call("css_bn_peer_gather", bases, W, r, seq, slot, local[r], n, outs[r], status[r], BIG, 2, d, st)
            for r in range(W):
                assert torch.equal(outs[r], want), (seq, r)
        else:             # forward form: finalize fused, against css_bn_finalize on the summed statistics
            res = []
            for r in range(W):
                call("css_bn_peer_finalize", bases, W, r, seq, slot, local[r], G, C, None, None, None, None, 0.0, 0.0, None, None, None, None, None,
                     status[r], BIG, 1, d, st)
            for r in range(W):
                o = [torch.empty(G * C, device=dev()) for _ in range(4)]
                rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
                cnt = torch.empty(G, dtype=torch.float64, device=dev())
                call("css_bn_peer_finalize", bases, W, r, seq, slot, local[r], G, C, gamma, beta, rm, rv, 0.1, 1e-5, o[0], o[1], o[2], o[3], cnt,
                     status[r], BIG, 2, d, st)
                res.append(o + [rm, rv, cnt])
            ref = [torch.empty(G * C, device=dev()) for _ in range(4)]
            rm, rv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
            call("css_bn_finalize", want, G, 0.0, want[G * 2 * C:], gamma, beta, rm, rv, 0.1, 1e-5, ref[0], ref[1], ref[2], ref[3], C, d, st)
            for r in range(W):
                # same formulas in another kernel: equal up to the compiler's choice of fused multiply-adds (last bit) ...
                for a, b in zip(res[r][:6], ref + [rm, rv]):
                    assert torch.allclose(a, b, rtol=2e-6, atol=1e-9), (seq, r, float((a - b).abs().max()))
                assert torch.equal(res[r][6], want[G * 2 * C:])
                # ... and bit-identical on every rank (same kernel, same numbers, same order)
                for a, b in zip(res[r], res[0]):
                    assert torch.equal(a, b), (seq, r)
    torch.cuda.synchronize()
    assert all(int(s.item()) == 0 for s in status)
    # a dead peer: rank 0 publishes exchange 11, ranks 1 and 2 never do; rank 0 waits 2 ms, gives up, says so, and the stream goes on
    loc = torch.ones(n, dtype=torch.float64, device=dev())
    out = torch.empty(n, dtype=torch.float64, device=dev())
    call("css_bn_peer_gather", bases, W, 0, 11, slot, loc, n, out, status[0], 200_000, 0, d, st)
    torch.cuda.synchronize()
    assert int(status[0].item()) == 11 and bool(torch.isfinite(out).all())


PEER2_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
from css_amd import ops
torch.manual_seed(5)
x_all = torch.randn(8, 17, 17, 64)
x = x_all[rank * 4:(rank + 1) * 4].to(dev, torch.bfloat16).requires_grad_(True)
gamma, beta = (torch.rand(64) + 0.5).to(dev).requires_grad_(True), torch.randn(64).to(dev).requires_grad_(True)
rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
res = []
for it in range(3):
    out = ops.bn_act(x, gamma, beta, rm, rv, relu=True, training=True, sync=True, groups=1)
    (out.float() * torch.linspace(0, 1, 64, device=dev)).sum().backward()
    res.append([out.float().sum().item(), x.grad.float().abs().sum().item(), gamma.grad.sum().item()])
    x.grad = None; gamma.grad = None; beta.grad = None
torch.cuda.synchronize()
if os.environ.get("CSS_SYNCBN") == "peer":
    from css_amd import peer
    peer.exchange(dev).check()
json.dump(dict(res=res, rm=rm.tolist(), rv=rv.tolist()), open(sys.argv[1] + str(rank), "w"))
dist.destroy_process_group()
'''


def test_peer_syncbn_two_processes(tmp_path):
    """The peer exchange across TWO processes on two GPUs (IPC-mapped buffers over xGMI) against the RCCL all-reduce path: same batch-norm
    outputs, gradients and running statistics on both ranks.  Needs a node with >= 2 GPUs (the single-GPU boxes of the test pool skip it;
    everything up to the wire is covered by the two tests above)."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on one node")
    got = {}
    for mode in ("rccl", "peer"):
        procs = []
        for r in range(2):
            env = dict(os.environ, CSS_SYNCBN=mode, MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", RANK=str(r), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, "-c", PEER2_WORKER % ROOT, str(tmp_path / mode)], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
        got[mode] = [json.load(open(str(tmp_path / mode) + str(r))) for r in range(2)]
    for r in range(2):
        a, b = got["rccl"][r], got["peer"][r]
        for x, y in zip(sum(a["res"], []), sum(b["res"], [])):
            assert abs(x - y) <= 1e-5 * max(1.0, abs(x)), (a, b)
        assert max(abs(x - y) for x, y in zip(a["rm"] + a["rv"], b["rm"] + b["rv"])) < 1e-6
    assert got["peer"][0]["rm"] == got["peer"][1]["rm"]             # rank-ordered sums: bit-identical statistics on both ranks


DDP_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parallel import DistributedDataParallel
import css_amd.compat
css_amd.compat.install()
# ---- the imports of mix_label.py:9-21, resolved by the aliases ----
from generalframeworks.networks import resnet
from generalframeworks.networks.ddp_model import Model_mix
from generalframeworks.scheduler.my_lr_scheduler import PolyLR
from generalframeworks.scheduler.rampscheduler import RampdownScheduler
from generalframeworks.utils import label_onehot, label_onehot_2
from generalframeworks.loss.loss import ProbOhemCrossEntropy2d, Attention_Threshold_Loss, Contrast_Loss
from css_amd.loss.loss import CrossEntropyLoss
from oracle import css_oracle as O

rank = 0
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", rank))
g = dict(np.load(os.path.join(%r, "tests", "golden", "train_trace_damped.npz")))
seed, gain = int(g["seed"]), float(g["residual_gain"])
K, S = 21, 65
config = {"Dataset": {"crop_size": (S, S), "scale_size": (1.0, 1.0), "mix_mode": "none", "device_aug": "identity"}, "Network": {"num_class": K}}
backbone = resnet.resnet101_tv()
model = Model_mix(backbone, num_classes=K, output_dim=256, config=config, temp=0.5)
sd = O.init_state("tv", K, 256, seed, gain)
model.model.load_state_dict(sd, strict=True); model.ema_model.load_state_dict(sd, strict=True)
model = model.cuda()                                                                   # mix_label.py:75
model = nn.SyncBatchNorm.convert_sync_batchnorm(model)                                 # :76
model = DistributedDataParallel(model, device_ids=[rank], find_unused_parameters=True)  # :77
criterion = {"ce_loss": CrossEntropyLoss(ignore_index=-1).cuda(), "unsup_loss": Attention_Threshold_Loss(0.97).cuda(),
             "contrast_loss": Contrast_Loss(strong_threshold=0.8, num_queries=64, num_negatives=128, temp=0.5, alpha=0.99).cuda()}
optimizer = torch.optim.SGD(model.module.model.parameters(), lr=6.4e-3, weight_decay=5e-4, momentum=0.9, nesterov=True)
scheduler = PolyLR(optimizer, 100, min_lr=1e-4)
sche_d = RampdownScheduler(begin_epoch=0, max_epoch=200, current_epoch=0, max_value=1.0, min_value=0.1, ramp_mult=-5.0)
prototypes = torch.zeros(K, 256).cuda()
num_class, weak = K, 0.0
model.module.model.train(); model.module.ema_model.train()
T = lambda a: torch.from_numpy(np.asarray(a))
train_l_image, train_l_label = T(g["0::l_img"]).cuda(), T(g["0::l_lab"]).long().cuda()
train_u_image = T(g["0::u_img"]).cuda()
# ---- mix_label.py:166-196 ----
pred_l_large, pred_u_large, train_u_aug_label, train_u_aug_logits_cls, train_u_aug_logits_rep, rep_all, pred_all = model(train_l_image, train_u_image, prototypes)
sup_loss = criterion["ce_loss"](pred_l_large, train_l_label)
unsup_loss = criterion["unsup_loss"](pred_u_large, train_u_aug_label, train_u_aug_logits_cls)
with torch.no_grad():
    train_u_aug_mask = train_u_aug_logits_cls.ge(weak).float()
    mask_all = torch.cat(((train_l_label.unsqueeze(1) >= 0).float(), train_u_aug_mask.unsqueeze(1)))
    mask_all = F.interpolate(mask_all, size=pred_all.shape[2:], mode="nearest")
    label_l = F.interpolate(label_onehot(train_l_label, num_class), size=pred_all.shape[2:], mode="nearest")
    label_u = F.interpolate(label_onehot_2(train_u_aug_label, num_class), size=pred_all.shape[2:], mode="nearest")
    label_u = label_u[:, 1:, :, :]
    label_all = torch.cat((label_l, label_u))
rec = {}
O.contrast_loss(rep_all.detach().float().cpu(), label_all.cpu(), mask_all.cpu(), pred_all.detach().float().cpu(), prototypes.cpu().clone(),
                64, 128, 0.5, 0.8, 0.99, record=rec)
anchors, negs, j = [], [], 0
for hn in rec["hard_num"]:
    if hn > 0:
        anchors.append(g[f"0::anchor{j}"].astype(np.int64)); negs.append(g[f"0::negative{j}"].astype(np.int64)); j += 1
    else:
        anchors.append(None); negs.append(None)
assert j == int(g["0::n_anchor"])
contrast_loss = criterion["contrast_loss"](rep_all, label_all, mask_all, pred_all, prototypes, _injected=dict(anchor=anchors, negative=negs))
total_loss = sup_loss + unsup_loss + contrast_loss * sche_d.value
optimizer.zero_grad()
total_loss.backward()
optimizer.step()
model.module.ema_update()
scheduler.step()
# ----
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
def probe(t):
    return t.detach().flatten()[:: max(1, t.numel() // 2048)][:2048]
sdm, sde = model.module.model.state_dict(), model.module.ema_model.state_dict()
out = dict(sup=[sup_loss.item(), float(g["0::sup"])], unsup=[unsup_loss.item(), float(g["0::unsup"])], con=[contrast_loss.item(), float(g["0::con"])],
           ramp=float(sche_d.value), mism=float((train_u_aug_label.cpu() != T(g["0::ulab"]).long()).float().mean()),
           protos=rel(prototypes.cpu(), T(g["0::protos"])),
           w={p: [rel(probe(sdm[p]).cpu(), T(g[f"0::student::{p}"])), rel(probe(sde[p]).cpu(), T(g[f"0::teacher::{p}"]))]
              for p in ["resnet_conv1.weight", "resnet_layer3.10.conv2.weight", "classifier.3.weight", "representation.3.bias"]},
           grads_none=[n for n, p in model.module.model.named_parameters() if p.grad is None])
json.dump(out, open(sys.argv[1], "w"))
dist.destroy_process_group()
'''


def test_mix_label_body_under_ddp_matches_reference_trace(tmp_path):
    """mix_label.py:75-77 + :166-196 verbatim: the reference's imports resolved by css_amd.compat, `.cuda()`,
    `SyncBatchNorm.convert_sync_batchnorm`, `DistributedDataParallel(find_unused_parameters=True)` over RCCL, the loop body with the
    one-hot / nearest label assembly, `torch.optim.SGD`, `model.module.ema_update()`, PolyLR, RampdownScheduler - against the first
    iteration of the trace captured from the reference (train_trace_damped.npz, recorded sampler draws injected)."""
    import json
    out = str(tmp_path / "ddp.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29580", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, "-c", DDP_WORKER % (ROOT, ROOT, ROOT), out], env=env)
    assert p.wait(timeout=900) == 0
    r = json.load(open(out))
    print(r)
    assert abs(r["ramp"] - 1.0) < 1e-12                      # RampdownScheduler at epoch 0
    for k, tol in (("sup", 2e-3), ("unsup", 3e-2), ("con", 2e-3)):
        hip, ref = r[k]
        assert abs(hip - ref) < tol * max(1.0, abs(ref)), (k, hip, ref)
    assert r["mism"] < 1e-3 and r["protos"] < 2e-3
    for name, (es, et) in r["w"].items():
        assert es < 3e-2 and et < 3e-2, (name, es, et)
    assert r["grads_none"] == [], r["grads_none"][:5]        # every student parameter received a gradient through DDP


UNEQUAL_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import torch.nn.functional as F
from css_amd import ops
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=rank, world_size=world)
bf16 = bool(os.environ.get("CSS_TEST_BF16"))
dt = torch.bfloat16 if bf16 else torch.float32
g = torch.Generator().manual_seed(3)
n_all, c, h, w = 8, int(os.environ.get("CSS_TEST_COUT", "64")), 17, 17
cin = 256 if c == 256 else 64                 # (256 input channels: enough K steps for the persistent kernels)
x = torch.randn(n_all, cin, h, w, generator=g) + 0.3
wt = torch.randn(c, cin, 1, 1, generator=g) / cin ** 0.5
go = torch.randn(n_all, c, h, w, generator=g)
gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
if bf16:
    x, wt, go = x.bfloat16().float(), wt.bfloat16().float(), go.bfloat16().float()
# reference: ONE process over all 8 images (nn.SyncBatchNorm semantics = batch statistics of the global batch)
xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
rm_r, rv_r = torch.zeros(c), torch.ones(c)
yr = F.conv2d(xr, wt)
if bf16:
    yr = yr + (yr.detach().bfloat16().float() - yr.detach())      # the HIP path normalises the bf16-rounded conv output
o = F.relu(F.batch_norm(yr, rm_r, rv_r, gr, br, True, 0.1, 1e-5))
o.backward(go)
# ranks own 6 and 2 images: unequal pixel counts per rank
lo, hi = (0, 6) if rank == 0 else (6, 8)
xg = x[lo:hi].permute(0, 2, 3, 1).contiguous().to(dev, dt).requires_grad_(True)
wg = wt.to(dev).contiguous(memory_format=torch.channels_last)
gg, bg = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
with ops.bn_groups(1):
    y = ops.conv2d(xg, wg, None, 1, 0, 1, bn_stats=bf16)
    assert hasattr(y, "_css_bnstats") == bf16
    a = ops.bn_act(y, gg, bg, rm, rv, None, True, True, 0.1, 1e-5, True)
a.backward(go[lo:hi].permute(0, 2, 3, 1).contiguous().to(dev, dt))
def rel(p, q):
    return float((p.double() - q.double()).abs().max() / q.double().abs().max())
dgam, dbet = gg.grad.clone(), bg.grad.clone()
dist.all_reduce(dgam); dist.all_reduce(dbet)                      # parameter gradients are local sums (DDP reduces them)
out = dict(a=rel(a.detach().float().cpu().permute(0, 3, 1, 2), o.detach()[lo:hi]), dx=rel(xg.grad.float().cpu().permute(0, 3, 1, 2), xr.grad[lo:hi]),
           rm=rel(rm.cpu(), rm_r), rv=rel(rv.cpu(), rv_r), dgamma=rel(dgam.cpu(), gr.grad), dbeta=rel(dbet.cpu(), br.grad))
json.dump(out, open(sys.argv[1] + str(rank), "w"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("bf16,cout", [(False, 64), (True, 64), (True, 256)])
def test_syncbn_with_unequal_pixel_counts_per_rank(tmp_path, bf16, cout):
    """nn.SyncBatchNorm (mix_label.py:76) exchanges per-rank counts; here the counts ride behind the (sum, sum of squares) payload
    of the one all-reduce.  Ranks with 6 and 2 images == one process with 8 (outputs, running statistics incl. the unbiased
    variance factor, input and parameter gradients) - both statistics paths (bn_stats pass / conv-epilogue slab rows; 256 output
    channels: the persistent 256x256 convolution kernel and its slab layout)."""
    import json
    out = str(tmp_path / "u.json")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", CSS_TEST_COUT=str(cout))
        if bf16:
            env["CSS_TEST_BF16"] = "1"
        procs.append(subprocess.Popen([sys.executable, "-c", UNEQUAL_WORKER % ROOT, out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    tol = 2e-2 if bf16 else 1e-4
    for r in range(2):
        res = json.load(open(out + str(r)))
        print(r, res)
        assert res["a"] < tol and res["dx"] < 5 * tol and res["dgamma"] < 5 * tol and res["dbeta"] < 5 * tol
        assert res["rm"] < (1e-2 if bf16 else 1e-5) and res["rv"] < (1e-2 if bf16 else 1e-5)


def test_bench_self_launcher_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the launcher starts two fresh ranks before anything touches the GPU; here they
    share cuda:0 and exchange through gloo (CSS_BENCH_SHARE_GPU=1; a one-GPU box cannot host two RCCL ranks), tiny crops."""
    import json
    env = dict(os.environ, CSS_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "65", "--batch", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0 and d["scaling"] == "weak"
    assert all(v == v for v in d["losses"].values())
