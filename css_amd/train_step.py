"""One training iteration of the CSS "mix_label" method on MI355X.

``MixTrainer.step`` reproduces the body of ``train()`` in the reference's mix_label.py:162-196:

    model(l, u, prototypes)                                   ddp_model.py:99-156
    sup_loss   = CE | OHEM (pred_l_large, l_label)            mix_label.py:168-171
    unsup_loss = Attention_Threshold_Loss(...)                mix_label.py:172
    mask_all / label_all                                      mix_label.py:175-183  (as a class-id map)
    contrast   = Contrast_Loss(rep_all, ...)                  mix_label.py:185
    total = sup + unsup + contrast * ramp                     mix_label.py:187-190
    zero_grad; backward; SGD(nesterov) step; ema_update; lr   mix_label.py:192-196

Data parallelism (mix_label.py:76-77): one process per GPU; SyncBN statistics and the contrastive prototype sums are
all-reduced inside the ops; the gradient of every student parameter lives in ONE flat fp32 buffer.  Like DDP's buckets
(mix_label.py:77) the buffer is all-reduced in a few pieces WHILE backward is still running: the kernels report each parameter
whose gradient has been enqueued (ops.set_grad_ready_callback); the first step records that order and cuts it into buckets of
CSS_GRAD_BUCKET_MB (default 48) MB, every later step launches a bucket's all-reduce (async, on a process group of its own so
that the latency-critical SyncBN collectives never queue behind 50 MB of gradients) as soon as its last parameter reported.
Whatever is left (parameters whose gradient goes through autograd's own accumulation) is reduced after backward; then ONE fused
SGD+EMA kernel over 59.5 M elements averages (1 / world) and applies.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import functional as Fn
from . import ops, peer
from ._lib import call, dev_stream, dtype_code
from .loss.loss import Attention_Threshold_Loss, Contrast_Loss, CrossEntropyLoss, ProbOhemCrossEntropy2d, fused_upsample_ok
from .scheduler.my_lr_scheduler import poly_lr


class MixTrainer:
    def __init__(self, model, num_classes=21, lr=6.4e-3, weight_decay=5e-4, momentum=0.9, total_iter=80000, min_lr=1e-4,
                 num_queries=256, num_negatives=512, temp_loss=0.5, alpha_proto=0.99, strong_threshold=0.8, weak_threshold=0.7,
                 un_threshold=0.97, sup="ce", ohem_min_kept=None, output_dim=256):
        self.model = model
        self.K = num_classes
        self.base_lr, self.wd, self.momentum, self.total_iter, self.min_lr = lr, weight_decay, momentum, total_iter, min_lr
        self.weak_threshold = weak_threshold
        self.crit_ce = CrossEntropyLoss(-1)
        self.crit_ohem = ProbOhemCrossEntropy2d(-1, thresh=0.7, min_kept=ohem_min_kept or 0) if sup == "ohem" else None
        self.crit_unsup = Attention_Threshold_Loss(un_threshold)
        self.crit_contrast = Contrast_Loss(num_queries, num_negatives, temp=temp_loss, strong_threshold=strong_threshold, alpha=alpha_proto)
        dev = next(model.parameters()).device
        self.prototypes = torch.zeros(num_classes, output_dim, device=dev)     # mix_label.py:93
        self.fused_loss = True     # CE / unsup / OHEM with the bilinear up-sampling folded in (no [B,K,H,W] logits)
        self.it = 0
        self.flat_p, self.flat_ema = model._ensure_flat()
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self._span = {}                        # id(param) -> (start, end) of its slice of the flat buffers (16-byte aligned slots)
        offs = list(model.model._css_flat_offsets) + [self.flat_p.numel()]
        for i, (p, o) in enumerate(zip(model.model.parameters(), model.model._css_flat_offsets)):
            n = p.numel()
            if p.dim() == 4:
                co, ci, r, s = p.shape
                p.grad = self.flat_g[o:o + n].view(co, r, s, ci).permute(0, 3, 1, 2)
            else:
                p.grad = self.flat_g[o:o + n].view(p.shape)
            self._span[id(p)] = (o, offs[i + 1])
        self.bucket_mb = float(os.environ.get("CSS_GRAD_BUCKET_MB", "48"))
        self._grad_pg = None                   # process group of the gradient buckets (created on first use, on every rank)
        self._ready_order = None               # parameter spans in the order their gradients were reported (recorded on the first step)
        self._buckets = None                   # [(runs = [(start, end), ...], frozenset of the bucket's spans)]
        self._span_reports = None              # span -> number of gradient reports per step (recorded on the first step)
        self._span_bucket = None               # span -> index of its bucket
        # The verdict of a step, agreed across ranks with one MAX all-reduce: non-zero when some rank's gradient reports did not fit the bucket
        # plan OR (CSS_SYNCBN=peer) an exchange of some rank gave up waiting for a peer.  It (a) stops the step's optimizer on the device
        # (css_sgd_ema's skip_flag: weights, momentum and teacher stay those of the last valid step - on EVERY rank, so replicas stay
        # equal) and (b) travels to a pinned host word behind an event that step k + VERDICT_LAG reads on every rank - see _check_verdicts.
        self._flags = [torch.zeros(2, device=self.flat_g.device) for _ in range(2)]   # [agreed verdict, this rank's own reason code]
        self._verdicts = []                    # pending (pinned [verdict, local reason], event, iteration) in step order
        self._pinned_pool = []
        self._skip_flag = None                 # device flag of THIS step for css_sgd_ema (non-zero: leave weights / momentum / teacher alone)

    # ---- what differs between the three entry scripts (overridden by CrossTrainer / OriTrainer below) -------------------------
    def _student_outputs(self, l_img, u_img):
        """-> low-resolution logits of both halves, (label, confidence) of the unsupervised loss, (label, confidence) behind the
        contrastive label / mask assembly, rep_all [2B,C,h,w] (NHWC memory), low-resolution logits of all 2B images or None."""
        pred_l, pred_u, u_lab, u_lc, u_lr, rep_all, _ = self.model.forward(l_img, u_img, self.prototypes, _want_prob=False, _small_logits=True)
        return pred_l, pred_u, (u_lab, u_lc), (u_lab, u_lc), rep_all, None

    def _contrast_labels(self, u_lab):
        """mix_label.py:181-182: label_onehot_2 + dropped channel 0 - an ignored (-1) pseudo label belongs to no class."""
        return u_lab

    def _hard_flags(self, rep_nhwc, cls, pred_small):
        """A valid pixel is hard when the probability of its own class (prototype similarity soft-max, ddp_model.py:147-154) is
        below the strong threshold (loss.py:90-91)."""
        _, _, hard = Fn.similarity(rep_nhwc, self.prototypes, self.model.temp, cls=cls, strong_threshold=self.crit_contrast.strong_threshold)
        return hard

    # ---- backward with the gradient all-reduce overlapped (DDP's bucketing, mix_label.py:77) ---------------------------------
    def _plan_buckets(self, order):
        """Cut the recorded readiness order (first report of every span) into buckets of ~bucket_mb; a bucket = (the contiguous runs
        of the flat buffer its parameters cover, the set of its spans) - backward runs the layers in reverse, so a bucket is one or
        two runs."""
        limit = self.bucket_mb * 2 ** 20 / 4
        buckets, cur, size = [], [], 0
        for sp in order:
            cur.append(sp)
            size += sp[1] - sp[0]
            if size >= limit:
                buckets.append((self._runs(cur), frozenset(cur)))
                cur, size = [], 0
        if cur:
            buckets.append((self._runs(cur), frozenset(cur)))
        return buckets

    @staticmethod
    def _runs(spans):
        runs = []
        for a, b in sorted(spans):
            if runs and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b])
        return [tuple(r) for r in runs]

    def _backward_and_reduce(self, total):
        """Protocol: the first step records, per span of the flat buffer, HOW OFTEN its gradient is reported (a parameter
        used by several graph nodes reports once per node) and the order in which the spans complete.  Later steps launch bucket b
        when every span of buckets 0..b has received ALL its recorded reports - never on a count of reports alone - and always in
        bucket order, so every rank issues the same sequence of collectives whatever the timing.  A report that does not fit the
        record (a span completes twice, an unknown span, a span that stays incomplete) is a violation: the remaining buckets are still
        launched in order, the ranks agree on the flag with one small all-reduce, and every rank raises at the same point (the start of
        step k + VERDICT_LAG, or finish() / state_dict())."""
        sync = ops.collectives_on()
        overlap = sync and self.bucket_mb > 0
        self._skip_flag = None
        if not overlap:
            with ops.direct_param_grads():       # parameter gradients are added in place into the flat buffer by the kernels
                total.backward()
            ops.assert_no_lazy_res_grads()
            if sync:
                dist.all_reduce(self.flat_g)     # one bucket, after backward
                if peer.enabled():               # (no bucket plan here; the peer exchange still needs an agreed verdict)
                    self._agree_on_verdict(0, None, None)
            return
        if self._grad_pg is None:
            self._grad_pg = dist.new_group()     # same ranks, own communicator / stream
        works, done = [], []
        counts = {}                              # span -> reports so far in this step
        state = dict(b=0, bad=0)
        recording = self._buckets is None
        order = []

        def launch(runs):
            for a, b in runs:
                works.append(dist.all_reduce(self.flat_g[a:b], group=self._grad_pg, async_op=True))
                done.append((a, b))

        def ready(p):
            sp = self._span.get(id(p))
            if sp is None:
                return
            n = counts.get(sp, 0) + 1
            counts[sp] = n
            if recording:
                if n == 1:
                    order.append(sp)
                else:                            # completion order = order of the LAST report
                    order.remove(sp)
                    order.append(sp)
                return
            want = self._span_reports.get(sp)
            if want is None or n > want:         # not in the record, or a contribution behind a bucket that may already be on the wire
                state["bad"] = 1
                return
            if n < want:
                return
            self._pending[self._span_bucket[sp]] -= 1
            while state["b"] < len(self._buckets) and self._pending[state["b"]] == 0:
                launch(self._buckets[state["b"]][0])
                state["b"] += 1

        self._check_bucket_flag()                # (direct callers; MixTrainer.step has read it already, before queueing its forward)
        if not recording:
            self._pending = [len(spans) for _, spans in self._buckets]
        prev = ops.set_grad_ready_callback(ready)
        try:
            with ops.direct_param_grads():
                total.backward()
        finally:
            ops.set_grad_ready_callback(prev)
        ops.assert_no_lazy_res_grads()
        if recording:                            # first step: learn the order and the report counts, reduce in one piece
            self._ready_order = list(order)
            self._span_reports = dict(counts)
            self._buckets = self._plan_buckets(order)
            self._span_bucket = {sp: b for b, (_, spans) in enumerate(self._buckets) for sp in spans}
            dist.all_reduce(self.flat_g, group=self._grad_pg)
            if peer.enabled():
                self._agree_on_verdict(0, self._grad_pg, None)
            return
        if state["b"] < len(self._buckets):      # incomplete spans: the graph changed; keep the collective sequence identical on every rank
            state["bad"] = 1
            while state["b"] < len(self._buckets):
                launch(self._buckets[state["b"]][0])
                state["b"] += 1
        # the rest of the buffer: parameters that never report (gradient through autograd's own accumulation) and alignment gaps
        pos, rest = 0, []
        for a, b in sorted(done):
            if a > pos:
                rest.append((pos, a))
            pos = max(pos, b)
        if pos < self.flat_g.numel():
            rest.append((pos, self.flat_g.numel()))
        launch(rest)
        self._agree_on_verdict(state["bad"], self._grad_pg, works)
        for w in works:
            w.wait()                             # the compute stream waits for the buckets (host does not block on RCCL)

    def _agree_on_verdict(self, bad, group, works):
        """One small MAX all-reduce behind the gradient buckets: [bucket-plan violation | peer-exchange timeout] of ANY rank -> every rank.
        ``works``: a list to append the asynchronous collective to (overlap path), or None for a blocking call on ``group``."""
        flag = self._flags[self.it & 1]
        code = 1.0 if bad else 0.0
        flag.fill_(code)
        ex = peer.exchange(self.flat_p.device) if (peer.enabled() and ops.collectives_on()) else None
        if ex is not None:                       # this step's exchanges (forward and backward) are all queued ahead of this point
            timed_out = ex.status.ne(0).to(flag.dtype)
            flag[0:1].copy_(torch.maximum(flag[0:1], timed_out))
            flag[1:2].add_(2.0 * timed_out)
            ex.status.zero_()                    # (stream-ordered behind the read: the next step starts from a clean word)
        if works is not None:
            works.append(dist.all_reduce(flag[0:1], op=dist.ReduceOp.MAX, group=group, async_op=True))
            works[-1].wait()                     # (stream-level wait: the host copy below is ordered behind the collective)
        else:
            dist.all_reduce(flag[0:1], op=dist.ReduceOp.MAX, group=group)
        self._skip_flag = flag
        host = self._pinned_pool.pop() if self._pinned_pool else (torch.zeros(2).pin_memory() if flag.is_cuda else torch.zeros(2))
        host.copy_(flag, non_blocking=True)
        ev = None
        if flag.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._verdicts.append((host, ev, self.it))

    VERDICT_LAG = 2                            # the verdict of step k is read at the start of step k + VERDICT_LAG - on every rank

    def _check_verdicts(self, block=False):
        """Raise - on every rank, AT THE SAME STEP - when the agreed verdict of an earlier step was non-zero.  Inside a run the verdict of
        step k is read at a FIXED lag: at the start of step k + VERDICT_LAG, after ``event.synchronize()`` (a poll made the discovery
        point depend on how far each rank's host had run ahead, so ranks raised at different steps, re-recorded their bucket plans at different
        steps and issued different collective sequences).  The copy of step k has long landed by then (the device is at most in step k + 1), so
        the host still never waits for the previous backward; it only cannot run more than VERDICT_LAG steps ahead.  ``finish()``
        and ``state_dict()`` read everything that is pending (``block=True``) - callers reach them at the same step on every rank.  Late is
        safe: an invalid step was already skipped on the device.  On a violation every pending verdict is drained (their pinned words go back
        to the pool), and the iteration counter and the EMA step counter go back by the number of steps that were skipped on the device - the
        lr schedule and the EMA decay continue from the last APPLIED step - before the exception leaves."""
        while self._verdicts:
            host, ev, it = self._verdicts[0]
            if not block and it > self.it - self.VERDICT_LAG:
                return
            if ev is not None:
                ev.synchronize()
            self._verdicts.pop(0)
            verdict, mine = float(host[0]), int(host[1])
            self._pinned_pool.append(host)
            if verdict != 0.0:
                skipped = 1
                for h2, ev2, _ in self._verdicts:        # the steps queued behind it: were they applied or skipped as well?
                    if ev2 is not None:
                        ev2.synchronize()
                    skipped += int(float(h2[0]) != 0.0)
                    self._pinned_pool.append(h2)
                self._verdicts.clear()
                self._buckets = None
                self.it -= skipped                   # those steps changed nothing: lr schedule and EMA decay continue from the last applied step
                self.model.step -= skipped
                why = {0: "another rank reported it", 1: "this rank's gradient reports did not fit the recorded bucket plan",
                       2: "a SyncBN peer exchange of this rank timed out waiting for a peer (CSS_PEER_TIMEOUT_S)",
                       3: "bucket-plan violation and peer-exchange timeout on this rank"}.get(mine, str(mine))
                raise RuntimeError(f"training step {it} was invalid and has been skipped on every rank ({why}): gradient readiness changed "
                                   "between steps on some rank (a span reported more or less often than on the first step) or SyncBN "
                                   f"statistics were incomplete; {skipped} step(s) were skipped on the device, buckets were reset on every rank, "
                                   "weights / momentum / teacher are those of the last valid step (batch-norm running statistics of a skipped "
                                   "step are not rolled back) - rebuild the trainer or resume from the last checkpoint")

    def _check_bucket_flag(self):
        """(kept for direct callers of _backward_and_reduce: the fixed-lag read)"""
        self._check_verdicts(False)

    def finish(self):
        """Call after the last step of a run and before saving a checkpoint: surfaces a pending invalid-step verdict (waits for it)."""
        self._check_verdicts(True)

    def state_dict(self):
        """Trainer state for a checkpoint (checked first: a step whose gradients were invalid raises here instead of being saved)."""
        self.finish()
        return dict(it=self.it, step=self.model.step, prototypes=self.prototypes.detach().clone(), momentum=self.flat_m.detach().clone())

    @property
    def lr(self):
        return poly_lr(self.base_lr, self.it, self.total_iter, 0.9, self.min_lr)

    def step(self, l_img, l_lab, u_img, ramp=1.0, _injected=None):
        m = self.model
        self._check_verdicts(False)                                          # (fixed lag: the verdict of step it - 2; the device is past it)
        self.flat_g.zero_()                                                  # optimizer.zero_grad()
        # student logits come back at LOW resolution (NHWC): the losses fold the bilinear up-sampling in whenever its factor
        # allows (>= 2: 513/129, 769/193 in the reference's configs), else they are up-sampled here like ddp_model.py:141,144
        pred_l, pred_u, (un_lab, un_conf), (u_lab, u_lc), rep_all, pred_small = self._student_outputs(l_img, u_img)
        if self.fused_loss and fused_upsample_ok(pred_l.shape[1:3], l_img.shape[2:], self.K):
            sup = (self.crit_ohem or self.crit_ce).forward_small(pred_l, l_lab)
            unsup = self.crit_unsup.forward_small(pred_u, un_lab, un_conf)
        else:
            hh, ww = l_img.shape[2:]
            sup = (self.crit_ohem or self.crit_ce)(ops.bilinear(pred_l, hh, ww, torch.float32).permute(0, 3, 1, 2), l_lab)
            unsup = self.crit_unsup(ops.bilinear(pred_u, hh, ww, torch.float32).permute(0, 3, 1, 2), un_lab, un_conf)
        b2, c, h, w = rep_all.shape
        rep_rows = rep_all.permute(0, 2, 3, 1).reshape(b2 * h * w, c)         # zero-copy: rep_all is NHWC memory
        with torch.no_grad():
            cls = Fn.class_map(l_lab, self._contrast_labels(u_lab), u_lc, self.weak_threshold, (h, w))
            hard = self._hard_flags(rep_rows.view(b2, h, w, c), cls, pred_small)
        con = self.crit_contrast.forward_fused(rep_rows, cls, hard, self.prototypes, self.K, _injected)
        total = sup + unsup + con * ramp
        self._backward_and_reduce(total)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        decay = min(1 - 1 / (m.step + 1), m.alpha)
        dev, st = dev_stream(self.flat_p)
        call("css_sgd_ema", self.flat_p, self.flat_g, self.flat_m, self.flat_ema, self.flat_p.numel(), float(self.lr), float(self.momentum),
             float(self.wd), int(self.it == 0), float(decay), 1.0 / world, self._skip_flag, dev, st)
        m.refresh_weights()
        m.step += 1
        self.it += 1
        return dict(sup=sup.detach(), unsup=unsup.detach(), contrast=con.detach(), total=total.detach(), pseudo=u_lab)


class CrossTrainer(MixTrainer):
    """The train body of cross_label.py:162-198 for ``Model_cross``: the unsupervised loss follows the class-predictor pseudo labels
    while ``warmup`` is set (epoch < args.warmup, :174-175) and the representation-space ones afterwards (:176-177); the contrastive
    label / mask assembly always uses the class-predictor maps through ``label_onehot`` (:186-187), whose ReLU sends an ignored (-1)
    label to class 0."""

    def __init__(self, *a, warmup=True, **k):
        super().__init__(*a, **k)
        self.warmup = warmup

    def _student_outputs(self, l_img, u_img):
        pred_l, pred_u, u_lab_c, u_lab_r, u_lc, u_lr, rep_all, _ = self.model.forward(l_img, u_img, self.prototypes, _want_prob=False,
                                                                                    _small_logits=True)
        un = (u_lab_c, u_lc) if self.warmup else (u_lab_r, u_lr)
        return pred_l, pred_u, un, (u_lab_c, u_lc), rep_all, None

    def _contrast_labels(self, u_lab):
        return torch.relu(u_lab)


class OriTrainer(MixTrainer):
    """The train body of ori_pseudo.py:158-187 for ``Model_ori_pseudo``: no prototype similarity in the model; ``prob_all`` of the
    contrastive loss is the soft-max of the student's own low-resolution logits (:178), labels through ``label_onehot`` (ReLU)."""

    def _student_outputs(self, l_img, u_img):
        pred_l, pred_u, u_lab, u_lg, rep_all, pred_all, _ = self.model.forward(l_img, u_img, _small_logits=True)
        return pred_l, pred_u, (u_lab, u_lg), (u_lab, u_lg), rep_all, pred_all

    def _contrast_labels(self, u_lab):
        return torch.relu(u_lab)

    def _hard_flags(self, rep_nhwc, cls, pred_small):
        b2, k, h, w = pred_small.shape                                    # logical NCHW view of NHWC memory
        rows = pred_small.detach().permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        hard = torch.empty(b2 * h * w, dtype=torch.uint8, device=rows.device)
        dev, st = dev_stream(rows)
        call("css_softmax_hard_flags", rows, k, cls, b2 * h * w, k, float(self.crit_contrast.strong_threshold), hard, dtype_code(rows.dtype), dev, st)
        return hard
