"""torch.autograd.Function wrappers over the C ABI (``include/css_hip.h``).

Internal activation format: contiguous ``[N, H, W, C]`` tensors (NHWC), fp32 (parity path) or
bf16 (throughput path).  PyTorch is used only for device memory, streams and autograd
bookkeeping; every arithmetic operation below is a HIP kernel in ``css_amd/csrc``.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib, peer
from ._lib import call, dev_stream, dtype_code

BN_EPS = 1e-5


def vec_of(dt: torch.dtype) -> int:
    return 8 if dt == torch.bfloat16 else 4


def pad_to(c: int, v: int) -> int:
    return (c + v - 1) // v * v


# --------------------------------------------------------------------------
# weight preparation cache (fp32 master [Cout,Cin,R,S] channels_last -> compute layouts)
# --------------------------------------------------------------------------
_epoch = 0


def invalidate_weight_cache():
    """Call after parameters were modified behind autograd's back (fused SGD / EMA kernels)."""
    global _epoch
    _epoch += 1


def _phys(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,R,S,Cin] contiguous fp32 view (or copy) of a conv weight."""
    w = weight.detach().permute(0, 2, 3, 1)
    if not w.is_contiguous():
        w = w.contiguous()
    return w


def prepared_weight(weight: torch.Tensor, dtype: torch.dtype, cin_pad: int, dgrad: bool) -> torch.Tensor:
    # the cache lives ON the parameter object (a global dict keyed by id() can hand a new model the
    # prepared weights of a dead one whose id and recycled device address coincide)
    cache = weight.__dict__.setdefault("_css_wcache", {})
    key = (dtype, cin_pad, dgrad)
    stamp = (weight.data_ptr(), weight._version, _epoch)
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    cout, cin, r, s = weight.shape
    w = _phys(weight)
    dev, st = dev_stream(w)
    v = vec_of(dtype)
    if not dgrad:
        if dtype == torch.float32 and cin_pad == cin:
            out = w
        else:
            out = torch.empty((cout, r, s, cin_pad), dtype=dtype, device=w.device)
            call("css_weight_layout", w, out, cout, r * s, cin, cin_pad, 0, dtype_code(dtype), dev, st)
    else:
        cout_pad = pad_to(cout, v)
        if cout_pad != cout:
            wp = torch.zeros((cout_pad, r, s, cin), dtype=torch.float32, device=w.device)
            wp[:cout] = w
            w = wp
        out = torch.empty((cin, r, s, cout_pad), dtype=dtype, device=w.device)
        call("css_weight_layout", w, out, cout_pad, r * s, cin, cin, 1, dtype_code(dtype), dev, st)
    cache[key] = (stamp, out)
    return out


def prepare_flat_weights(module, dtype: torch.dtype, dgrad: bool) -> None:
    """Refresh the compute-dtype copies of EVERY conv weight of a module whose parameters live in one flat buffer
    (networks.ddp_model.flatten_parameters) in two launches instead of one or two per layer: a cast of the whole buffer
    (the flat layout already is [Cout][R][S][Cin] per layer) and, with ``dgrad``, one batched transpose kernel.
    The per-layer entries of ``prepared_weight``'s cache are re-pointed at views of the two persistent buffers."""
    flat = getattr(module, "_css_flat", None)
    if flat is None or dtype == torch.float32:
        return
    v = vec_of(dtype)
    plan = module.__dict__.get("_css_lp_plan")
    if plan is None or plan["src"] is not flat or plan["dtype"] != dtype:
        lp = torch.empty(flat.numel(), dtype=dtype, device=flat.device)
        fwd, dg, desc, tiles, doff = [], [], [], 0, 0
        for p, o in zip(module.parameters(), module._css_flat_offsets):
            if p.dim() != 4:
                continue
            co, ci, r, s_ = p.shape
            n = p.numel()
            if ci % v == 0:
                fwd.append((p, lp[o:o + n].view(co, r, s_, ci), ci))
            if co % v == 0 and ci % v == 0 and p.requires_grad:
                desc.append([o, doff, co, r * s_, ci, tiles])
                dg.append((p, doff, (ci, r, s_, co)))
                tiles += -(-ci // 32) * -(-co // 32) * r * s_
                doff += (n + 7) // 8 * 8
        lpt = torch.empty(max(doff, 8), dtype=dtype, device=flat.device)
        plan = dict(src=flat, dtype=dtype, lp=lp, lpt=lpt, fwd=fwd, tiles=tiles,
                    dg=[(p, lpt[o:o + sh[0] * sh[1] * sh[2] * sh[3]].view(sh)) for p, o, sh in dg],
                    desc=torch.tensor(desc, dtype=torch.int64, device=flat.device) if desc else None)
        module.__dict__["_css_lp_plan"] = plan
    dev, st = dev_stream(flat)
    dc = dtype_code(dtype)
    call("css_cast", flat, plan["lp"], flat.numel(), dtype_code(torch.float32), dc, dev, st)
    for p, view, ci in plan["fwd"]:
        p.__dict__.setdefault("_css_wcache", {})[(dtype, ci, False)] = ((p.data_ptr(), p._version, _epoch), view)
    if dgrad and plan["desc"] is not None:
        call("css_weight_dgrad_layout_batched", flat, plan["lpt"], plan["desc"], len(plan["dg"]), plan["tiles"], dc, dev, st)
        for p, view in plan["dg"]:
            p.__dict__.setdefault("_css_wcache", {})[(dtype, p.shape[1], True)] = ((p.data_ptr(), p._version, _epoch), view)


# --------------------------------------------------------------------------
# direct gradient accumulation (trainer mode)
# --------------------------------------------------------------------------
_direct_grads = False


class direct_param_grads:
    """Inside this context the conv / batch-norm backward kernels ADD parameter gradients straight into the existing
    ``param.grad`` buffers (the trainer's flat gradient buffer) and report ``None`` to autograd: no per-layer zero-fill,
    no AccumulateGrad add.  Only for callers that own ``param.grad`` (``MixTrainer``); DDP needs the ordinary path."""

    def __enter__(self):
        global _direct_grads
        self.prev, _direct_grads = _direct_grads, True

    def __exit__(self, *a):
        global _direct_grads
        _direct_grads = self.prev


# Called with a parameter right after the kernels that ADD its gradient into the flat buffer were enqueued (direct mode only):
# lets the trainer start the all-reduce of a gradient bucket while the rest of backward is still running (train_step.py)
_grad_ready_cb = None


def set_grad_ready_callback(cb):
    global _grad_ready_cb
    prev, _grad_ready_cb = _grad_ready_cb, cb
    return prev


def _grad_ready(*params):
    if _grad_ready_cb is not None:
        for p in params:
            if p is not None:
                _grad_ready_cb(p)


def _grad_sink(param, phys_shape):
    """param.grad as a contiguous fp32 buffer in the kernel's physical layout, or None if it cannot be used in place."""
    g = getattr(param, "grad", None)
    if not _direct_grads or g is None or g.dtype != torch.float32:
        return None
    if g.dim() == 4:
        g = g.permute(0, 2, 3, 1)
    if not g.is_contiguous() or tuple(g.shape) != tuple(phys_shape):
        return None
    return g


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


# --------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, stat_groups, tap):
        global _conv_stats_out
        n, h, w_, cp = x.shape
        cout, cin, r, s = weight.shape
        dt = x.dtype
        assert x.is_contiguous() and cp == pad_to(cin, vec_of(dt)), (x.shape, weight.shape)
        ho, wo = conv_out_size(h, r, stride, pad, dil), conv_out_size(w_, s, stride, pad, dil)
        wf = prepared_weight(weight, dt, cp, False)
        y = torch.empty((n, ho, wo, cout), dtype=dt, device=x.device)
        dev, st = dev_stream(x)
        flops = 2.0 * n * ho * wo * cout * r * s * cin
        m = n * ho * wo
        _conv_stats_out = None
        if (stat_groups and bias is None and dt == torch.bfloat16 and m % stat_groups == 0 and m // stat_groups >= 128
                and cout % 8 == 0):
            # batch-norm statistics of the output from the convolution's own epilogue (no bn_stats pass)
            mg = m // stat_groups
            stats = torch.empty((2 * ((m + 255) // 256), 2, cout), dtype=torch.float32, device=x.device)
            call("css_conv2d_forward_bnstats", x, wf, y, stats, mg, n, h, w_, cp, cp, ho, wo, cout, cout, r, s, stride, pad, dil,
                 flops, dtype_code(dt), dev, st)
            # rows per convolution tile (= two statistics slabs), asked of the library - never hard-coded here
            bm = _lib.query("css_conv2d_forward_bnstats_tile_rows", x, wf, y, n, h, w_, cp, cp, ho, wo, cout, cout, r, s, stride, pad, dil,
                            dtype_code(dt), dev)
            _conv_stats_out = (stats, mg, stat_groups, cout, bm)
        else:
            call("css_conv2d_forward", x, wf, bias, y, n, h, w_, cp, cp, ho, wo, cout, cout, r, s, stride, pad, dil, flops,
                 dtype_code(dt), dev, st)
        ctx.save_for_backward(x, weight)
        ctx.bias_ref = bias
        ctx.cfg = (stride, pad, dil, bias is not None, flops)
        if tap:
            return y, x      # x comes back as a second output: its gradient is folded into this op's dgrad store
        return y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias, flops = ctx.cfg
        n, h, w_, cp = x.shape
        cout, cin, r, s = weight.shape
        dt = x.dtype
        v = vec_of(dt)
        dy = dy.contiguous()
        ho, wo = dy.shape[1], dy.shape[2]
        dev, st = dev_stream(dy)
        dc = dtype_code(dt)
        cout_pad = pad_to(cout, v)
        dyp = dy
        if cout_pad != cout:   # heads with K classes: pad the reduced / vectorised dimension
            dyp = torch.zeros((n, ho, wo, cout_pad), dtype=dt, device=dy.device)
            call("css_copy_channels", dy, cout, dyp, cout_pad, n * ho * wo, cout, dc, dc, dev, st)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = prepared_weight(weight, dt, cp, True)
            dx = torch.empty_like(x)
            lazy = _take_lazy_res_grad(ctx, dtap)
            if lazy is not None:
                # the residual gradient arrives as (gradient at the ReLU output, ReLU bit mask): masked inside the store (_BNAct.backward)
                call("css_conv2d_dgrad_add_masked", dyp, wt, dx, lazy[0], cp, lazy[1], n, h, w_, cp, cp, ho, wo, cout_pad, cout_pad, r, s,
                     stride, pad, dil, flops, dc, dev, st)
            elif dtap is not None:
                dtap = dtap.contiguous()
                call("css_conv2d_dgrad_add", dyp, wt, dx, dtap, cp, n, h, w_, cp, cp, ho, wo, cout_pad, cout_pad, r, s, stride,
                     pad, dil, flops, dc, dev, st)
            else:
                call("css_conv2d_dgrad", dyp, wt, dx, n, h, w_, cp, cp, ho, wo, cout_pad, cout_pad, r, s, stride, pad, dil, flops,
                     dc, dev, st)
        elif dtap is not None:
            if _take_lazy_res_grad(ctx, dtap) is not None:
                raise _lib.CssHipError("a lazily masked residual gradient reached a convolution that computes no data gradient")
            dx = dtap
        if ctx.needs_input_grad[1]:
            sink = _grad_sink(weight, (cout, r, s, cin)) if (cout_pad == cout and cp == cin) else None
            # workspace for the per-slice partial tiles (plain stores + ordered reduction instead of fp32 atomics); 0 bytes: the
            # shape takes a kernel without that path
            wsb = _lib.query("css_conv2d_wgrad_ws_bytes", n * ho * wo, r * s * cp, cout_pad, dc, dev)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=dy.device) if wsb else None
            if sink is not None:      # the wgrad kernels ADD into dw: straight into param.grad
                call("css_conv2d_wgrad", x, dyp, sink, ws, wsb, n, h, w_, cp, cp, ho, wo, cout, cout, r, s, stride, pad, dil, flops,
                     dc, dev, st)
                _grad_ready(weight)
            else:
                dwp = torch.zeros((cout_pad, r, s, cp), dtype=torch.float32, device=dy.device)
                call("css_conv2d_wgrad", x, dyp, dwp, ws, wsb, n, h, w_, cp, cp, ho, wo, cout_pad, cout_pad, r, s, stride, pad, dil,
                     flops, dc, dev, st)
                dw = dwp[:cout, :, :, :cin].permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2]:
            bias_p = ctx.bias_ref
            sink = _grad_sink(bias_p, (cout,)) if bias_p is not None else None
            cws = torch.empty(_lib.query("css_colsum_ws_bytes", n * ho * wo, cout) // 4, dtype=torch.float32, device=dy.device)
            if sink is not None:
                call("css_colsum", dy, cout, n * ho * wo, cout, sink, cws, dc, dev, st)
                _grad_ready(bias_p)
            else:
                db = torch.zeros((cout,), dtype=torch.float32, device=dy.device)
                call("css_colsum", dy, cout, n * ho * wo, cout, db, cws, dc, dev, st)
        return dx, dw, db, None, None, None, None, None


# --------------------------------------------------------------------------
# the stride-2 stem convolution on the space-to-depth image (csrc/conv_stem.hip)
# --------------------------------------------------------------------------
class S2DInput:
    """The network input staged as [N, ceil(H/2), ceil(W/2), 16] bf16 (ops.stage_inputs(..., s2d=True)): 2 x 2 x 3 image values per pixel."""
    __slots__ = ("t", "hw")

    def __init__(self, t, hw):
        self.t, self.hw = t, hw

    @property
    def shape(self):          # (what the callers of a staged tensor look at: the batch size)
        return (self.t.shape[0], self.hw[0], self.hw[1], 3)


def stem_s2d_ok(conv, dtype) -> bool:
    """The first convolution of the backbone is one of the two stride-2 stems conv_stem_s2d_kernel takes (7x7 s2 p3 or 3x3 s2 p1, 3 -> 64, no bias),
    the compute dtype is bf16 and CSS_NO_STEM_S2D is not set."""
    if dtype != torch.bfloat16 or not _lib.available() or not _lib.query("css_stem_s2d_enabled"):
        return False
    k = getattr(conv, "kernel_size", None)
    return (k in ((7, 7), (3, 3)) and getattr(conv, "stride", None) == (2, 2) and conv.padding == (k[0] // 2, k[0] // 2)
            and conv.dilation == (1, 1) and conv.in_channels == 3 and conv.out_channels == 64 and conv.bias is None)


def _stem_s2d_weight(weight: torch.Tensor) -> torch.Tensor:
    """bf16 [64][TA][TA][16] copy of the fp32 master [64,3,R,R] (cached on the parameter like prepared_weight's layouts)."""
    cache = weight.__dict__.setdefault("_css_wcache", {})
    stamp = (weight.data_ptr(), weight._version, _epoch)
    hit = cache.get("s2d")
    if hit is not None and hit[0] == stamp:
        return hit[1]
    cout, _, r, _ = weight.shape
    ta = (r + 1) // 2
    w = _phys(weight)
    dev, st = dev_stream(w)
    out = torch.empty((cout, ta, ta, 16), dtype=torch.bfloat16, device=w.device)
    call("css_stem_s2d_weights", w, out, cout, r, dev, st)
    cache["s2d"] = (stamp, out)
    return out


class _StemS2D(torch.autograd.Function):
    """y [N,Hs,Ws,64] = conv(image, weight [64,3,R,R], stride 2, pad R/2) from the space-to-depth image; the weight gradient is computed in
    s2d space by the generic weight-gradient kernels (TA x TA taps, stride 1, 16 channels) and folded back.  The image needs no gradient."""

    @staticmethod
    def forward(ctx, xs, weight, stat_groups):
        global _conv_stats_out
        n, hs, ws, c16 = xs.shape
        cout, cin, r, _ = weight.shape
        assert c16 == 16 and xs.dtype == torch.bfloat16 and xs.is_contiguous() and cin == 3 and cout == 64
        w2 = _stem_s2d_weight(weight)
        y = torch.empty((n, hs, ws, cout), dtype=torch.bfloat16, device=xs.device)
        dev, st = dev_stream(xs)
        m = n * hs * ws
        flops = 2.0 * m * cout * r * r * cin
        _conv_stats_out = None
        stats, mg = None, 0
        if stat_groups and m % stat_groups == 0 and m // stat_groups >= 128:
            mg = m // stat_groups
            bt = _lib.query("css_conv2d_stem_s2d_tile_rows")           # the kernel's tile height: asked of the library, never hard-coded here
            stats = torch.empty((2 * ((m + bt - 1) // bt), 2, cout), dtype=torch.float32, device=xs.device)
        call("css_conv2d_stem_s2d_forward", xs, w2, y, stats, mg, n, hs, ws, cout, r, flops, dev, st)
        if stats is not None:
            _conv_stats_out = (stats, mg, stat_groups, cout, bt)
        ctx.save_for_backward(xs, weight)
        ctx.flops = flops
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, weight = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None
        n, hs, ws, _ = xs.shape
        cout, cin, r, _ = weight.shape
        ta = (r + 1) // 2
        dy = dy.contiguous()
        dev, st = dev_stream(dy)
        dc = dtype_code(torch.bfloat16)
        dw2 = torch.zeros((cout, ta, ta, 16), dtype=torch.float32, device=dy.device)
        wsb = _lib.query("css_conv2d_wgrad_ws_bytes", n * hs * ws, ta * ta * 16, cout, dc, dev)
        wsp = torch.empty(wsb // 4, dtype=torch.float32, device=dy.device) if wsb else None
        call("css_conv2d_wgrad", xs, dy, dw2, wsp, wsb, n, hs, ws, 16, 16, hs, ws, cout, cout, ta, ta, 1, ta // 2, 1, ctx.flops, dc, dev, st)
        sink = _grad_sink(weight, (cout, r, r, cin))
        if sink is not None:
            call("css_stem_s2d_fold_wgrad", dw2, sink, cout, r, dev, st)
            _grad_ready(weight)
            return None, None, None
        dwp = torch.zeros((cout, r, r, cin), dtype=torch.float32, device=dy.device)
        call("css_stem_s2d_fold_wgrad", dw2, dwp, cout, r, dev, st)
        return None, dwp.permute(0, 3, 1, 2), None


def conv2d_stem_s2d(x: S2DInput, weight, bn_stats=False):
    y = _StemS2D.apply(x.t, weight, (_bn_groups if bn_stats else 0))
    if _conv_stats_out is not None:
        y._css_bnstats = _conv_stats_out          # (picked up by bn_act: the statistics came out of the convolution's epilogue)
    return y



_conv_stats_out = None

# Residual gradients whose ReLU backward is still to be applied.  A tapped convolution (Bottleneck.conv1) and the batch norm that uses the
# tap as its residual share a _TapLink: created in conv2d(tap=True), owned by the convolution's autograd node (ctx.tap_link) and carried by
# the tap tensor to bn_act, whose backward parks (gradient at the ReLU output, ReLU bit mask) in it instead of writing a masked copy; the
# convolution's backward masks that addend inside its dgrad store.  The hand-over is bound to the consumer: the tensor that
# arrives as the tap's gradient must BE the parked one - if the tap found a second consumer, autograd hands over a sum that contains the
# unmasked gradient, and the backward raises instead of adding it.  Links die with their graph, so an aborted backward leaves nothing
# behind; every backward pass that parked something ends with a check (engine callback) that all of it was consumed.
# CSS_BN_EAGER_DRES=1: bn_bwd_apply writes the masked copy itself (round-2 behaviour; A/B and parity tests).
_lazy_dres = os.environ.get("CSS_BN_EAGER_DRES") != "1"


class _TapLink:
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None          # (da, mask) between _BNAct.backward and the tapped convolution's backward


_parked_links = []                   # links that received a pair during the running backward pass
_parked_task = None                  # autograd graph-task id the list belongs to


def _park_lazy_res_grad(link, da, mask):
    global _parked_task
    task = torch._C._current_graph_task_id()
    if task != _parked_task:         # a new backward pass: whatever an aborted one left is dropped, and this pass gets its end-of-pass check
        for l in _parked_links:
            l.pending = None
        _parked_links.clear()
        _parked_task = task
        if task >= 0:
            torch.autograd.Variable._execution_engine.queue_callback(assert_no_lazy_res_grads)
    link.pending = (da, mask)
    _parked_links.append(link)


def _take_lazy_res_grad(ctx, dtap):
    link = getattr(ctx, "tap_link", None)
    if dtap is None or link is None or link.pending is None:
        return None
    g, mask = link.pending
    link.pending = None
    if dtap.data_ptr() != g.data_ptr() or g.shape != dtap.shape or g.dtype != dtap.dtype or not dtap.is_contiguous():
        raise _lib.CssHipError("lazily masked residual gradient: the tensor that reached the tapped convolution is not the one its batch norm "
                               "parked (the tap has a second consumer?) - the sum would contain an unmasked gradient")
    return g, mask


def assert_no_lazy_res_grads():
    """End of a backward pass: every (gradient, mask) pair a batch norm parked for its tapped convolution was consumed."""
    left = [l for l in _parked_links if l.pending is not None]
    for l in left:
        l.pending = None
    _parked_links.clear()
    if left:
        raise _lib.CssHipError(f"{len(left)} residual gradient(s) left the batch-norm backward unmasked and were never masked by a dgrad store")


def conv2d(x, weight, bias=None, stride=1, pad=0, dil=1, bn_stats=False, tap=False):
    """``bn_stats``: the output feeds a train-mode batch norm; its statistics are then produced by the convolution epilogue
    and travel to ``bn_act`` as the ``_css_bnstats`` attribute of the result.  ``tap``: also return ``x`` itself as a second
    output; using THAT for the residual connection lets the backward fold the residual gradient into the dgrad store
    instead of a separate add over both tensors."""
    out = _Conv2d.apply(x, weight, bias, stride, pad, dil, _bn_groups if bn_stats else 0, tap)
    y = out[0] if tap else out
    if tap:
        # bn_act(res=<this>) may then leave the ReLU backward of the residual gradient to this op's dgrad store (see _TapLink)
        node = y.grad_fn
        link = _TapLink() if node is not None else None
        if node is not None:
            node.tap_link = link
        out[1]._css_tap = link
    if _conv_stats_out is not None:
        y._css_bnstats = _conv_stats_out
    return out


# --------------------------------------------------------------------------
# batch norm (+ residual, + ReLU)
# --------------------------------------------------------------------------
def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on():
    """True when the data-parallel exchanges (SyncBN statistics, prototype sums, gradient all-reduce) must run: more than one
    rank, or CSS_FORCE_COLLECTIVES=1 with an initialised group (a 1-rank RCCL group: exercises every collective call, dtype and
    stream hand-over on a single MI355X; the results are unchanged)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CSS_FORCE_COLLECTIVES") == "1"


def _row_stride(t):
    """Leading dimension of an NHWC tensor that is contiguous or a channel slice of a contiguous buffer, else None."""
    if t.stride(-1) != 1:
        return None
    ld = t.stride(-2)
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.stride(d) != exp:
            return None
        exp *= t.shape[d]
    return ld if ld >= t.shape[-1] else None


def _sync_finalize(stats, g, c, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, dev, st):
    """SyncBN forward: (sum, sum of squares, row count) of every rank -> global statistics, finalised.  Counts may differ per rank: they ride
    behind the sums.  -> the [G] global counts on the device (the backward divides by them)."""
    ex = peer.exchange(stats.device) if peer.enabled() else None      # (None: not asked for, or refused for this group -> RCCL)
    if ex is not None:           # peer-mapped exchange buffers: one launch, no RCCL call (css_amd/peer.py)
        count_t = torch.empty(g, dtype=torch.float64, device=stats.device)
        ex.finalize(stats, g, c, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, count_t)
        return count_t
    dist.all_reduce(stats)
    count_t = stats[g * 2 * c:]
    call("css_bn_finalize", stats, g, 0.0, count_t, gamma, beta, running_mean, running_var, float(momentum), float(eps),
         mean, invstd, scale, shift, c, dev, st)
    return count_t


class _BNAct(torch.autograd.Function):
    """Batch norm (+ residual, + ReLU).  ``groups`` = number of forward passes batched into ``y`` along dim 0: every group
    of rows gets its own batch statistics and one running-statistics update (see include/css_hip.h, batch-norm block)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, res, relu, training, momentum, eps, sync, groups, fused=None, out_into=None,
                tap_link=None, pool=None):
        c = y.shape[-1]
        m = y.numel() // c
        dt = y.dtype
        dev, st = dev_stream(y)
        dc = dtype_code(dt)
        assert y.is_contiguous() and (res is None or res.is_contiguous())
        f32 = dict(dtype=torch.float32, device=y.device)
        g = groups if training else 1
        assert m % g == 0, (m, g)
        mg = m // g
        scale, shift = torch.empty(g * c, **f32), torch.empty(g * c, **f32)
        mean = invstd = mask = None
        count = float(mg)
        count_t = None       # SyncBN: per-group global pixel counts on the device (all-reduced together with the sums)
        if training and fused is not None and fused[1:4] == (mg, g, c):
            # statistics came out of the producing convolution's epilogue (fp32 rows per 128-row slab)
            mean, invstd = torch.empty(g * c, **f32), torch.empty(g * c, **f32)
            if sync and collectives_on():
                stats = torch.empty(g * 2 * c + g, dtype=torch.float64, device=y.device)     # [G][2][C] sums + [G] local row counts
                call("css_bn_reduce_finalize_slabs", fused[0], m, mg, g, count, None, None, None, None, 0.0, 0.0, None, None, None,
                     None, stats, c, y, c, fused[4], dev, st)
                count_t = _sync_finalize(stats, g, c, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, dev, st)
            else:
                call("css_bn_reduce_finalize_slabs", fused[0], m, mg, g, count, gamma, beta, running_mean, running_var,
                     float(momentum), float(eps), mean, invstd, scale, shift, None, c, y, c, fused[4], dev, st)
        elif training:
            nrb = _lib.query("css_bn_nrb", mg, g, c, dc)
            partial = torch.empty((g, nrb, 2 * c), dtype=torch.float64, device=y.device)
            call("css_bn_stats", y, mg, g, c, c, partial, dc, dev, st)
            mean, invstd = torch.empty(g * c, **f32), torch.empty(g * c, **f32)
            if sync and collectives_on():
                stats = torch.empty(g * 2 * c + g, dtype=torch.float64, device=y.device)
                call("css_bn_reduce", partial, nrb, c, g, stats, None, None, 0, float(mg), dev, st)
                count_t = _sync_finalize(stats, g, c, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, dev, st)
            else:
                call("css_bn_reduce_finalize", partial, nrb, g, count, gamma, beta, running_mean, running_var, float(momentum),
                     float(eps), mean, invstd, scale, shift, c, dev, st)
        else:
            call("css_bn_eval_coeff", gamma, beta, running_mean, running_var, float(eps), scale, shift, c, dev, st)
        arg = None
        if pool is not None:
            # the stem's batch norm + ReLU + 3x3 s2 p1 max pool in one pass (css_bn_apply_maxpool): the normalised activation is never written
            assert res is None and out_into is None and y.dim() == 4
            n_, h_, w_, _ = y.shape
            ho_, wo_ = pool
            out = torch.empty((n_, ho_, wo_, c), dtype=dt, device=y.device)
            if training and any(ctx.needs_input_grad):
                arg = torch.empty((n_, ho_, wo_, c), dtype=torch.uint8, device=y.device)
            call("css_bn_apply_maxpool", y, out, arg, scale, shift, n_, h_, w_, c, ho_, wo_, g, int(relu), dc, dev, st)
        elif out_into is not None:      # write straight into a channel slice of a concat buffer (cat_from_views): no copy later
            buf, off = out_into
            out, ldo = buf[..., off:off + c], buf.shape[-1]
            assert buf.is_contiguous() and buf.dtype == dt and buf.shape[:-1] == y.shape[:-1]
            call("css_bn_apply", y, c, res, c, buf.data_ptr() + off * buf.element_size(), ldo, scale, shift, m, c, int(relu), mg, dc, dev, st)
        else:
            out = torch.empty_like(y)
            if training and relu and res is not None and _bn_bit_mask and any(ctx.needs_input_grad):
                # residual layer: the backward passes read the ReLU mask as one byte per 16-byte vector instead of `out` itself
                # (not for the EMA teacher: its parameters and inputs do not require gradients - no backward, no mask to write)
                mask = torch.empty((m, c // vec_of(dt)), dtype=torch.uint8, device=y.device)
                call("css_bn_apply_mask", y, c, res, c, out, c, scale, shift, m, c, int(relu), mg, mask, dc, dev, st)
            else:
                call("css_bn_apply", y, c, res, c, out, c, scale, shift, m, c, int(relu), mg, dc, dev, st)
        if training:
            # ReLU mask in backward: the bit mask (or `out`, CSS_BN_NO_MASK=1) when a residual was added, else recomputed from
            # y*scale+shift (no extra read at all)
            assert out_into is None or not (relu and res is not None)
            ctx.save_for_backward(y, out if (relu and res is not None and mask is None) else None, mean, invstd, gamma, scale, shift, count_t, mask, arg)
        ctx.beta_ref = beta
        ctx.cfg = (relu, training, count, sync, res is not None, g)
        # the residual came out of a tapped convolution: its backward applies this layer's ReLU mask to the gradient itself
        ctx.tap_link = tap_link if (mask is not None and _lazy_dres) else None
        return out

    @staticmethod
    def backward(ctx, da):
        relu, training, count, sync, has_res, g = ctx.cfg
        if not training:
            raise _lib.CssHipError("backward through eval-mode batch norm is not part of the CSS hot path")
        y, a, mean, invstd, gamma, scale, shift, count_t, mask, arg = ctx.saved_tensors
        c = y.shape[-1]
        if arg is not None:      # fused max pool: its adjoint first (the gradient of the pooled tensor -> the gradient of the normalised activation)
            n_, h_, w_, _ = y.shape
            dpool = da.contiguous()
            da = torch.empty_like(y)
            dev_, st_ = dev_stream(dpool)
            call("css_maxpool_bwd", dpool, arg, da, n_, h_, w_, c, dpool.shape[1], dpool.shape[2], 3, 2, 1, dtype_code(y.dtype), dev_, st_)
        m = y.numel() // c
        mg = m // g
        dt = y.dtype
        ldda = _row_stride(da)          # a channel slice of a concat gradient is read in place (cat_from_views)
        if ldda is None or (da.data_ptr() % 16) or (ldda * da.element_size()) % 16:
            da, ldda = da.contiguous(), c
        dev, st = dev_stream(da)
        dc = dtype_code(dt)
        nrb = _lib.query("css_bn_nrb", mg, g, c, dc)
        partial = torch.empty((g, nrb, 2 * c), dtype=torch.float64, device=y.device)
        if mask is not None:
            call("css_bn_bwd_reduce_mask", da, ldda, mask, y, c, mean, invstd, mg, g, c, partial, dc, dev, st)
        else:
            call("css_bn_bwd_reduce", da, ldda, a, c, y, c, mean, invstd, scale, shift, mg, g, c, int(relu), partial, dc, dev, st)
        sums = torch.empty(g * 2 * c, dtype=torch.float64, device=y.device)
        # parameter gradients are LOCAL sums over all groups (DDP / the trainer all-reduce them with the rest)
        sg, sb = _grad_sink(gamma, (c,)), _grad_sink(ctx.beta_ref, (c,))
        if sg is not None and sb is not None:
            call("css_bn_reduce", partial, nrb, c, g, sums, sg, sb, 1, 0.0, dev, st)
            dgamma = dbeta = None
            _grad_ready(gamma, ctx.beta_ref)
        else:
            dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
            call("css_bn_reduce", partial, nrb, c, g, sums, dgamma, dbeta, 0, 0.0, dev, st)
        if sync and collectives_on():
            # SyncBN backward: global sum(dz), sum(dz*xhat) per group
            ex = peer.exchange(y.device) if peer.enabled() else None
            if ex is not None:
                ex.gather(sums)
            else:
                dist.all_reduce(sums)
        dy = torch.empty_like(y)
        lazy = ctx.tap_link is not None and mask is not None and ldda == c and da.is_contiguous() and da.shape == y.shape
        dres = torch.empty_like(y) if (has_res and not lazy) else None
        if mask is not None:
            call("css_bn_bwd_apply_mask", da, ldda, mask, y, c, dy, c, dres, c, mean, invstd, gamma, sums, count, count_t, m, c, mg, dc, dev, st)
            if lazy:
                # no masked copy of `da` for the residual branch: `da` itself travels on, with the mask on the side (see _lazy_res_grads)
                _park_lazy_res_grad(ctx.tap_link, da, mask)
                dres = da
        else:
            call("css_bn_bwd_apply", da, ldda, a, c, y, c, dy, c, dres, c, mean, invstd, gamma, sums, scale, shift, count, count_t, m, c,
                 int(relu), mg, dc, dev, st)
        return dy, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None, None, None


_bn_groups = 1
_nbt_sink = None
# CSS_BN_NO_MASK=1: residual layers re-read their activation tensor for the ReLU mask in backward (round-2 behaviour; A/B and parity tests)
_bn_bit_mask = os.environ.get("CSS_BN_NO_MASK") != "1"


class bn_groups:
    """Context manager: the activations inside hold ``g`` forward passes batched along dim 0 (equal sizes)."""

    def __init__(self, g):
        self.g = g

    def __enter__(self):
        global _bn_groups, _nbt_sink
        self.prev, _bn_groups = _bn_groups, self.g
        self.prev_sink, _nbt_sink = _nbt_sink, []

    def __exit__(self, *a):
        global _bn_groups, _nbt_sink
        # every layer's num_batches_tracked += groups in ONE multi-tensor launch (113 single-element adds per pass otherwise)
        if _nbt_sink:
            torch._foreach_add_(_nbt_sink, self.g)
        _bn_groups, _nbt_sink = self.prev, self.prev_sink


def count_bn_batch(counter):
    """num_batches_tracked bookkeeping of a train-mode forward (torch BatchNorm semantics)."""
    if _nbt_sink is not None:
        _nbt_sink.append(counter)
    else:
        counter += _bn_groups


def bn_act(y, gamma, beta, running_mean, running_var, res=None, relu=True, training=True, momentum=0.1, eps=BN_EPS, sync=True,
           groups=None, out_into=None, pool=None):
    """``pool = (Ho, Wo)``: the 3x3 stride-2 pad-1 max pool that follows the stem's batch norm, fused into the apply pass."""
    return _BNAct.apply(y, gamma, beta, running_mean, running_var, res, relu, training, momentum, eps, sync,
                        _bn_groups if groups is None else groups, getattr(y, "_css_bnstats", None), out_into,
                        getattr(res, "_css_tap", None) if res is not None else None, pool)


def fused_stem_pool() -> bool:
    """CSS_NO_BN_POOL=1: the stem's max pool as a pass of its own behind bn_apply (round-4 behaviour; A/B and parity tests)."""
    return os.environ.get("CSS_NO_BN_POOL", "0") in ("", "0")


def pool_out_size(i, ks, stride, pad, ceil_mode):
    num = i + 2 * pad - ks
    o = (-(-num // stride) if ceil_mode else num // stride) + 1
    if ceil_mode and (o - 1) * stride >= i + pad:
        o -= 1
    return o


# --------------------------------------------------------------------------
# pooling / resize / concat
# --------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ks, stride, pad, ceil_mode):
        n, h, w, c = x.shape

        def osz(i):
            num = i + 2 * pad - ks
            o = (-(-num // stride) if ceil_mode else num // stride) + 1
            if ceil_mode and (o - 1) * stride >= i + pad:
                o -= 1
            return o

        ho, wo = osz(h), osz(w)
        out = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
        arg = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
        dev, st = dev_stream(x)
        call("css_maxpool_fwd", x, out, arg, n, h, w, c, ho, wo, ks, stride, pad, dtype_code(x.dtype), dev, st)
        ctx.save_for_backward(arg)
        ctx.cfg = (x.shape, ks, stride, pad)
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        (n, h, w, c), ks, stride, pad = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty((n, h, w, c), dtype=dout.dtype, device=dout.device)
        dev, st = dev_stream(dout)
        call("css_maxpool_bwd", dout, arg, dx, n, h, w, c, dout.shape[1], dout.shape[2], ks, stride, pad,
             dtype_code(dout.dtype), dev, st)
        return dx, None, None, None, None


def maxpool(x, ks=3, stride=2, pad=1, ceil_mode=False):
    return _MaxPool.apply(x, ks, stride, pad, ceil_mode)


class _Bilinear(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=True) on NHWC tensors; output dtype selectable."""

    @staticmethod
    def forward(ctx, x, hd, wd, out_dtype, out_into=None):
        n, hs, ws, c = x.shape
        dev, st = dev_stream(x)
        ctx.cfg = (x.shape, x.dtype)
        if out_into is not None:      # straight into a channel slice of a concat buffer (cat_from_views)
            buf, off = out_into
            assert buf.dtype == out_dtype and buf.is_contiguous() and tuple(buf.shape[:3]) == (n, hd, wd)
            call("css_bilinear", x, c, buf.data_ptr() + off * buf.element_size(), buf.shape[-1], n, hs, ws, c, hd, wd, dtype_code(x.dtype),
                 dtype_code(out_dtype), 0, dev, st)
            return buf[..., off:off + c]
        out = torch.empty((n, hd, wd, c), dtype=out_dtype, device=x.device)
        call("css_bilinear", x, c, out, c, n, hs, ws, c, hd, wd, dtype_code(x.dtype), dtype_code(out_dtype), 0, dev, st)
        return out

    @staticmethod
    def backward(ctx, dout):
        (n, hs, ws, c), in_dtype = ctx.cfg
        ldo = _row_stride(dout)       # a channel slice of a concat gradient is read in place
        if ldo is None:
            dout, ldo = dout.contiguous(), c
        hd, wd = dout.shape[1], dout.shape[2]
        dx = torch.empty((n, hs, ws, c), dtype=in_dtype, device=dout.device)
        dev, st = dev_stream(dout)
        call("css_bilinear", dout, ldo, dx, c, n, hs, ws, c, hd, wd, dtype_code(dout.dtype), dtype_code(in_dtype), 1, dev, st)
        return dx, None, None, None, None


def bilinear(x, hd, wd, out_dtype=None, out_into=None):
    return _Bilinear.apply(x, hd, wd, out_dtype or x.dtype, out_into)


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, h, w, c = x.shape
        out = torch.empty((n, 1, 1, c), dtype=x.dtype, device=x.device)
        dev, st = dev_stream(x)
        call("css_spatial_sum", x, c, out, n, h * w, c, 1.0 / (h * w), dtype_code(x.dtype), dev, st)
        ctx.cfg = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        n, h, w, c = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty((n, h, w, c), dtype=dout.dtype, device=dout.device)
        dev, st = dev_stream(dout)
        call("css_spatial_bcast", dout, dx, c, n, h * w, c, 1.0 / (h * w), dtype_code(dout.dtype), dev, st)
        return dx


def global_avg_pool(x):
    return _GlobalAvgPool.apply(x)


class _Broadcast(torch.autograd.Function):
    """[N,1,1,C] -> [N,H,W,C] (bilinear resize of a 1x1 map, aspp.py:38)."""

    @staticmethod
    def forward(ctx, x, h, w, out_into=None):
        n, _, _, c = x.shape
        dev, st = dev_stream(x)
        if out_into is not None:
            buf, off = out_into
            call("css_spatial_bcast", x, buf.data_ptr() + off * buf.element_size(), buf.shape[-1], n, h * w, c, 1.0, dtype_code(x.dtype), dev, st)
            return buf[..., off:off + c]
        out = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
        call("css_spatial_bcast", x, out, c, n, h * w, c, 1.0, dtype_code(x.dtype), dev, st)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        n, h, w, c = dout.shape
        dx = torch.empty((n, 1, 1, c), dtype=dout.dtype, device=dout.device)
        dev, st = dev_stream(dout)
        call("css_spatial_sum", dout, c, dx, n, h * w, c, 1.0, dtype_code(dout.dtype), dev, st)
        return dx, None, None, None


def broadcast_hw(x, h, w, out_into=None):
    return _Broadcast.apply(x, h, w, out_into)


class _CatFromViews(torch.autograd.Function):
    """The concatenation whose pieces were WRITTEN IN PLACE into ``buf`` by their producers (``out_into=(buf, offset)`` of
    bn_act / broadcast_hw): forward is free, backward hands every producer its channel slice of the gradient as a view."""

    @staticmethod
    def forward(ctx, buf, *views):
        off = 0
        for v in views:
            assert v.data_ptr() == buf.data_ptr() + off * buf.element_size() and v.shape[:-1] == buf.shape[:-1]
            off += v.shape[-1]
        assert off == buf.shape[-1]
        ctx.cs = [v.shape[-1] for v in views]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        outs, off = [None], 0
        for i, c in enumerate(ctx.cs):
            outs.append(dout[..., off:off + c] if ctx.needs_input_grad[i + 1] else None)
            off += c
        return tuple(outs)


def cat_from_views(buf, *views):
    return _CatFromViews.apply(buf, *views)


class _CatChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        n, h, w, _ = xs[0].shape
        cs = [x.shape[-1] for x in xs]
        ct = sum(cs)
        dt = xs[0].dtype
        out = torch.empty((n, h, w, ct), dtype=dt, device=xs[0].device)
        dev, st = dev_stream(out)
        off = 0
        esz = out.element_size()
        for x, c in zip(xs, cs):
            call("css_copy_channels", x, c, out.data_ptr() + off * esz, ct, n * h * w, c, dtype_code(dt), dtype_code(dt), dev, st)
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        n, h, w, ct = dout.shape
        dev, st = dev_stream(dout)
        dc = dtype_code(dout.dtype)
        esz = dout.element_size()
        outs, off = [], 0
        for i, c in enumerate(ctx.cs):
            if ctx.needs_input_grad[i]:
                g = torch.empty((n, h, w, c), dtype=dout.dtype, device=dout.device)
                call("css_copy_channels", dout.data_ptr() + off * esz, ct, g, c, n * h * w, c, dc, dc, dev, st)
                outs.append(g)
            else:
                outs.append(None)
            off += c
        return tuple(outs)


def cat_channels(*xs):
    return _CatChannels.apply(*xs)


class _Split2(torch.autograd.Function):
    """(x[:b], x[b:]) along dim 0 with a single-copy backward (autograd's own slicing would zero-fill two full tensors)."""

    @staticmethod
    def forward(ctx, x, b):
        ctx.b, ctx.shape, ctx.dt = b, x.shape, x.dtype
        return x.narrow(0, 0, b), x.narrow(0, b, x.shape[0] - b)

    @staticmethod
    def backward(ctx, g0, g1):
        out = torch.empty(ctx.shape, dtype=ctx.dt, device=(g0 if g0 is not None else g1).device)
        for g, sl in ((g0, out.narrow(0, 0, ctx.b)), (g1, out.narrow(0, ctx.b, ctx.shape[0] - ctx.b))):
            if g is None:
                sl.zero_()
            else:
                sl.copy_(g)
        return out, None


def split2(x, b):
    return _Split2.apply(x, b)


def stage_inputs(xs, dtype: torch.dtype, s2d: bool = False):
    """Several [Bi,C,H,W] fp32 NCHW images -> ONE [sum Bi,H,W,Cpad] ``dtype`` tensor (each written into its slice); ``s2d``: the
    space-to-depth staging of the stride-2 stems instead (S2DInput: [sum Bi, ceil(H/2), ceil(W/2), 16] bf16 - half the bytes)."""
    xs = [x.detach() if (x.dtype == torch.float32 and x.is_contiguous()) else x.detach().float().contiguous() for x in xs]
    _, c, h, w = xs[0].shape
    if s2d:
        assert dtype == torch.bfloat16 and c <= 3
        hs, ws = (h + 1) // 2, (w + 1) // 2
        bt = sum(x.shape[0] for x in xs)
        out = torch.empty((bt, hs, ws, 16), dtype=dtype, device=xs[0].device)
        dev, st = dev_stream(out)
        off = 0
        for x in xs:
            call("css_nchw_to_s2d", x, out.data_ptr() + off * hs * ws * 16 * out.element_size(), x.shape[0], c, h, w, dev, st)
            off += x.shape[0]
        return S2DInput(out, (h, w))
    cp = pad_to(c, vec_of(dtype))
    bt = sum(x.shape[0] for x in xs)
    out = torch.empty((bt, h, w, cp), dtype=dtype, device=xs[0].device)
    dev, st = dev_stream(out)
    off = 0
    for x in xs:
        b = x.shape[0]
        call("css_nchw_to_nhwc", x, out.data_ptr() + off * h * w * cp * out.element_size(), b, c, h * w, cp, dtype_code(dtype), dev, st)
        off += b
    return out


def stage_input(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[B,C,H,W] fp32 NCHW image -> [B,H,W,Cpad] ``dtype`` (zero-padded channels). No gradient."""
    x = x.detach()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    b, c, h, w = x.shape
    cp = pad_to(c, vec_of(dtype))
    out = torch.empty((b, h, w, cp), dtype=dtype, device=x.device)
    dev, st = dev_stream(x)
    call("css_nchw_to_nhwc", x, out, b, c, h * w, cp, dtype_code(dtype), dev, st)
    return out
