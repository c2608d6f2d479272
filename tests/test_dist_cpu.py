"""world_size-2 gloo tests (CPU): the two exchanges the data-parallel path relies on are algebraically the reference's.

1. Prototype statistics: all-reduce of per-class (sum of embeddings, count) == the reference's all_gather of all
   embeddings followed by a masked mean (loss.py:77,81,102), including the local-presence rule (loss.py:96).
2. SyncBN statistics: all-reduce of per-channel (sum, sum of squares) gives the batch statistics of the concatenated
   batch (nn.SyncBatchNorm, mix_label.py:76).
3. Flat gradient buffer: one SUM all-reduce scaled by 1/world == DDP's averaged gradients.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import css_oracle as O
    K, C, h = 6, 16, 7
    g = torch.Generator().manual_seed(100 + rank)
    rep = torch.randn(2, C, h, h, generator=g)
    lab = torch.randint(0, 4 if rank == 0 else 6, (2, h, h), generator=g)        # classes 4,5 exist on rank 1 only
    label = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float()
    mask = (torch.rand(2, 1, h, h, generator=g) > 0.2).float()
    prob = torch.softmax(torch.randn(2, K, h, h, generator=g), 1)
    protos0 = torch.randn(K, C, generator=torch.Generator().manual_seed(7))
    protos0[1] = 0

    # (a) the reference way: gather everything, masked mean   [oracle restatement of loss.py:75-109]
    def gather(t):
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t.contiguous())
        return torch.cat(out, 0)
    p_ref = protos0.clone()
    torch.manual_seed(0)
    O.contrast_loss(rep, label, mask, prob, p_ref, 8, 16, 0.5, 0.8, 0.99, gather=gather)

    # (b) the css_amd way: local per-class sums + counts, one all-reduce, then the same update rule
    valid = (label * mask)                                            # [2,K,h,h]
    rows = rep.permute(0, 2, 3, 1).reshape(-1, C).double()
    vm = valid.permute(0, 2, 3, 1).reshape(-1, K).double()
    stats = torch.cat([(vm.t() @ rows).flatten(), vm.sum(0)])         # K*C sums, K counts
    local_cnt = vm.sum(0).clone()
    dist.all_reduce(stats)
    sums, cnt = stats[:K * C].view(K, C), stats[K * C:]
    p_new = protos0.clone()
    for k in range(K):
        if local_cnt[k] == 0:                                         # local-presence rule (loss.py:96)
            continue
        mean = (sums[k] / cnt[k]).float()
        p_new[k] = mean if p_new[k].sum() == 0 else 0.99 * p_new[k] + 0.01 * mean
    err_proto = (p_new - p_ref).abs().max().item()

    # (c) SyncBN statistics
    x = torch.randn(3, 5, 4, 4, generator=g) * (rank + 1) + rank
    st = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))]).double()
    dist.all_reduce(st)
    n = world * x.numel() // 5
    mean, var = st[0] / n, st[1] / n - (st[0] / n) ** 2
    xs = gather(x)
    err_bn = max((mean - xs.mean((0, 2, 3))).abs().max().item(), (var - xs.var((0, 2, 3), unbiased=False)).abs().max().item())

    # (d) flat gradient all-reduce == DDP mean
    grad = torch.randn(1000, generator=g)
    flat = grad.clone()
    dist.all_reduce(flat)
    flat /= world
    allg = gather(grad.view(1, -1)).mean(0)
    err_grad = (flat - allg).abs().max().item()
    q.put((rank, err_proto, err_bn, err_grad, float(p_ref[4].abs().sum()), float(p_new[4].abs().sum())))
    dist.destroy_process_group()


def test_world2_exchanges_equal_reference_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, e_proto, e_bn, e_grad, ref4, new4 in res:
        assert e_proto < 1e-5, (rank, e_proto)
        assert e_bn < 1e-5 and e_grad < 1e-6
    # class 4 lives on rank 1 only: rank 0 must leave its prototype row untouched, rank 1 updates it (reference behaviour)
    assert abs(res[0][4] - res[0][5]) < 1e-5 and abs(res[1][4] - res[1][5]) < 1e-5


def _bucket_worker(rank, world, port, q, mode="plain"):
    """MixTrainer._backward_and_reduce on a toy graph (CPU tensors, gloo): layers whose backward ADDS the parameter gradient into the
    flat buffer and reports the parameter (as the HIP conv / batch-norm backward do), one parameter whose gradient goes through
    autograd's own accumulation (never reported), three steps: record, bucketed, bucketed."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from css_amd import ops
    from css_amd.train_step import MixTrainer

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            ctx.w = w
            return x * w.sum()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            ctx.w.grad += (g * x).sum() * torch.ones_like(w)         # "kernel adds into the flat buffer"
            ops._grad_ready(ctx.w)
            return g * w.sum(), None

    sizes = [5, 300, 8, 1000, 16, 700, 24]                           # 7 layers + 1 late parameter
    offs, tot = [], 0
    for n in sizes + [40]:
        offs.append(tot)
        tot += (n + 7) // 8 * 8
    flat_p = torch.randn(tot, generator=torch.Generator().manual_seed(1))
    flat_g = torch.zeros(tot)
    params = []
    for n, o in zip(sizes + [40], offs):
        p = torch.nn.Parameter(flat_p[o:o + n].clone())
        p.grad = flat_g[o:o + n]
        params.append(p)
    tr = MixTrainer.__new__(MixTrainer)
    tr.flat_g = flat_g
    tr._span = {id(p): (o, e) for p, o, e in zip(params, offs, offs[1:] + [tot])}
    tr.bucket_mb = 4096 / 2 ** 20                                   # 1024 floats per bucket
    tr._grad_pg = tr._ready_order = tr._buckets = tr._span_reports = tr._span_bucket = tr._skip_flag = None
    tr._flags = [torch.zeros(2), torch.zeros(2)]
    tr._verdicts, tr._pinned_pool = [], []
    tr.model = type("M", (), dict(step=0))()                    # (the EMA step counter a skipped step rolls back)
    tr.it = 0
    errs = []
    raised = []
    for step in range(3 if mode == "plain" else 5):
        tr.it = step
        flat_g.zero_()
        x = torch.full((), float(rank + 1 + step), requires_grad=True)
        h = x
        for i, p in enumerate(params[:-1]):
            if mode == "violate" and step == 2 and rank == 1 and i == 3:
                continue                                             # the graph changes on ONE rank: that layer never reports
            h = Lin.apply(h, p)
        h = Lin.apply(h, params[1])                                  # a parameter used by two nodes reports twice (its bucket waits for both)
        total = h + (params[-1] * (rank + 2)).sum()                  # last parameter: plain autograd accumulation, never reported
        # reference: the same backward without any collective, then one all-reduce
        ref_g = torch.zeros(tot)
        for p, o in zip(params, offs):
            p.grad = ref_g[o:o + p.numel()]
        total.backward(retain_graph=True)
        dist.all_reduce(ref_g)
        for p, o in zip(params, offs):
            p.grad = flat_g[o:o + p.numel()]
        try:
            tr._backward_and_reduce(total)
        except RuntimeError as e:
            raised.append((step, "readiness changed" in str(e)))
            continue
        errs.append((flat_g - ref_g).abs().max().item())
    try:
        tr.finish()
    except RuntimeError as e:
        raised.append((-1, "readiness changed" in str(e)))
    q.put((rank, errs, len(tr._buckets) if tr._buckets else 0, [r for r, _ in tr._buckets] if tr._buckets else [], len(tr._ready_order), raised))
    dist.destroy_process_group()


def test_world2_bucketed_gradient_all_reduce_equals_one_all_reduce():
    os.environ["CSS_FORCE_COLLECTIVES"] = "0"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, errs, nb, runs, nrep, raised in res:
        assert max(errs) == 0.0, (rank, errs)                        # two ranks: a + b in any order is the same float
        assert nrep == 7 and nb >= 2, (nb, nrep)
        assert not raised, raised
        # backward visits the layers in reverse; the twice-used parameter completes last: at most two runs per bucket
        assert all(len(r) <= 2 for r in runs), runs
    assert res[0][3] == res[1][3]                                    # same plan on both ranks


def test_world2_bucket_plan_violation_raises_on_every_rank_without_hanging():
    """when the graph of ONE rank changes (a layer stops reporting), that rank must not stop issuing the
    collectives the other rank is waiting in, and both ranks must learn about it at the same point of the protocol: the start of
    step k + MixTrainer.VERDICT_LAG (a fixed lag - ADVICE r05: never a poll, whose outcome depends on each host's run-ahead), from the
    agreed flag.  Step 3 (queued behind the invalid step 2, its graph intact again on both ranks but recorded plan unchanged) is valid:
    exactly one step is rolled back."""
    os.environ["CSS_FORCE_COLLECTIVES"] = "0"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q, "violate")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, errs0, _, _, _, raised0), (r1, errs1, _, _, _, raised1) = res
    assert raised1 == [(4, True)], raised1                           # the rank whose graph changed: step 2 + the fixed lag
    assert raised0 == [(4, True)], raised0                           # the other rank: same step, from the agreed flag
    assert len(errs0) == 4 and len(errs1) == 4                       # steps 0-3 ran to completion on both ranks (2 skipped on the device only)
    assert errs0[:2] == [0.0, 0.0] and errs1[:2] == [0.0, 0.0]       # the steps before the change were exact


def _mix_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import numpy as np
    from css_amd.dataset_helpers import gpu_aug
    z = np.load(os.path.join(ROOT, "tests", "golden", "cut_gather_w2.npz"))
    ins = lambda: [torch.from_numpy(z[f"in_r{rank}_{k}"]).clone() for k in range(4)]
    res = {}
    for mode in ("none", "cutmix", "cutout"):
        res[mode] = [o.numpy() for o in gpu_aug.generate_cut_gather_2(*ins(), mode=mode, rng=np.random.RandomState(50 + rank))]
    # the same with rank 0's images already on their way before the "teacher" (gpu_aug.prefetch_partner_image: identity in-step augmentation)
    img, lab, l1, l2 = ins()
    pre = gpu_aug.prefetch_partner_image(img, "cutmix")
    assert pre is not None and gpu_aug.prefetch_partner_image(img, "cutout") is None
    res["cutmix_prefetched"] = [o.numpy() for o in gpu_aug.generate_cut_gather_2(img, lab, l1, l2, mode="cutmix", rng=np.random.RandomState(50 + rank),
                                                                                 prefetched=pre)]
    img, lab, l1, l2 = ins()
    res["cutmix3"] = [o.numpy() for o in gpu_aug.generate_cut_gather_3(img, lab, lab + 1, l1, l2, mode="cutmix", rng=np.random.RandomState(50 + rank))]
    # the packed buffer itself: ignore labels (-1) travel as byte 255 and come back as -1; float payloads bit for bit; odd sizes (alignment)
    g = torch.Generator().manual_seed(9)
    t_img = torch.randn(2, 3, 5, 7, generator=g) + rank
    t_lab = torch.randint(-1, 21, (2, 5, 7), generator=g) + 0 * rank
    t_map = torch.rand(2, 5, 7, generator=g) + rank
    unpack, _ = gpu_aug._broadcast_packed([t_img, t_lab, t_map])
    got = unpack()
    g0 = torch.Generator().manual_seed(9)
    w_img, w_lab, w_map = torch.randn(2, 3, 5, 7, generator=g0), torch.randint(-1, 21, (2, 5, 7), generator=g0), torch.rand(2, 5, 7, generator=g0)
    assert torch.equal(got[0], w_img) and torch.equal(got[1], w_lab) and torch.equal(got[2], w_map) and got[1].dtype == torch.int64
    assert bool((w_lab == -1).any())
    q.put((rank, res))
    dist.destroy_process_group()


def test_world2_cut_gather_partner_and_draws_follow_the_reference():
    """generate_cut_gather_2/3 under two ranks against vectors captured from the reference run by two gloo processes
    (tests/golden/make_cut_gather_w2.py; VOC.py:393-477): the partner of every image is RANK 0's image (i+1) % B (the reference
    indexes the all-gathered batch with the local batch size) and every rank draws one box per GATHERED image and uses its own
    block.  The oracle's restatement (oracle/aug_oracle.py: cut_gather_ranks) is pinned on the same vectors."""
    import numpy as np
    os.environ["CSS_FORCE_COLLECTIVES"] = "0"
    os.environ.pop("CSS_CUTMIX_LOCAL", None)
    z = np.load(os.path.join(ROOT, "tests", "golden", "cut_gather_w2.npz"))
    modes = ("none", "cutmix", "cutout", "cutmix3")
    n_out = {"none": 4, "cutmix": 4, "cutout": 4, "cutmix3": 5}
    # (1) the oracle against the reference's vectors
    sys.path.insert(0, ROOT)
    from oracle import aug_oracle as A
    ins = [[torch.from_numpy(z[f"in_r{r}_{k}"]) for k in range(4)] for r in range(2)]
    for mode in modes:
        per_rank = [tuple(i) for i in ins] if mode != "cutmix3" else [(i[0], i[1], i[1] + 1, i[2], i[3]) for i in ins]
        want = A.cut_gather_ranks(per_rank, mode.rstrip("3"), [np.random.RandomState(50 + r) for r in range(2)])
        for r in range(2):
            for k in range(n_out[mode]):
                assert np.array_equal(want[r][k].numpy(), z[f"out_{mode}_r{r}_{k}"]), ("oracle", mode, r, k)
    # (2) css_amd under two gloo ranks against the same vectors
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_mix_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for mode in modes:
        for r in range(2):
            for k in range(n_out[mode]):
                assert np.array_equal(res[r][mode][k], z[f"out_{mode}_r{r}_{k}"]), ("css_amd", mode, r, k)
    for r in range(2):                                               # packed broadcast + prefetched images: the same bits
        for k in range(4):
            assert np.array_equal(res[r]["cutmix_prefetched"][k], z[f"out_cutmix_r{r}_{k}"]), ("prefetched", r, k)
            assert res[r]["cutmix_prefetched"][k].dtype == z[f"out_cutmix_r{r}_{k}"].dtype
    # rank 1's cutmix result contains rank 0's pixels: it differs from mixing inside its own batch
    from css_amd.dataset_helpers import gpu_aug
    local = gpu_aug.generate_cut_gather_2(*[t.clone() for t in ins[1]], mode="cutmix", rng=np.random.RandomState(51))
    assert not np.array_equal(local[0].numpy(), z["out_cutmix_r1_0"])
