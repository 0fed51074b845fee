"""The SPADE+Style block (normalization.py:91-192): InstanceNorm statistics, label-map convs and their batched prepass, the
[gamma | beta] branch with its label-sparse forward and backward, the modulation (two-launch and fused into the conv), plain InstanceNorm."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from .. import packing
from .._lib import NORM_SPADE_STYLE_BATCH, NORM_ACCUMULATE_DX, ACT_NONE, AUX_NONE, AUX_RELU_MASK, NORM_SPADE_STYLE
from . import switches
from .core import (GradSink, IN_EPS, LaunchProfiler, ZeroPool, _adjacent, _byref, _cl_dense, _cl_rows, _dt, _grad_dst, _need, _p, _span2, _stream)
from .conv import _CONV_STATS_SLOTS, _conv_plan, _unpack_dw, conv2d_raw, conv2d_wgrad_raw, packed_weight, unpack_weight_grad_into


def in_stats(x, return_sums=False):
    """(N,H,W,C) -> (N,C,2) fp32 {mean, rstd}; not differentiated here (the IN backward lives in
    modulate_bwd, once per consumer of the statistics).
    return_sums: also the raw fp64 per-sample sums (N,C,2) {sum x, sum x^2} (BatchNorm SPADE combines them over the batch)."""
    _need(x)
    n, h, w, c = x.shape
    ws = torch.empty(L.lib().s2e_in_stats_workspace_bytes(_dt(x), n, h * w, c) // 8, dtype=torch.float64, device=x.device)
    stats = torch.empty(n, c, 2, dtype=torch.float32, device=x.device)
    cnt = None          # (the C ABI's one-launch form -- zeroed block counters -- measured slower: include/seg2eye_hip.h)
    LaunchProfiler.run('in_stats', 0.0, lambda: L.check(
        L.lib().s2e_in_stats(_dt(x), _p(x), n, h * w, c, IN_EPS, _p(ws), _p(stats), _p(cnt), _stream()), 's2e_in_stats'),
        nbytes=float(x.numel() * x.element_size()))                   # algorithmic: x read once
    return (stats, ws[:n * c * 2].view(n, c, 2)) if return_sums else stats


def label_conv3x3_raw(label, weight, bias, n, H, W, h, w, cout, relu, dtype):
    """weight: the (Cout, ncls, 3, 3) fp32 conv weight itself (the kernel gathers its table from it)."""
    _need(label, weight, bias)
    out = torch.empty(n, h, w, cout, dtype=dtype, device=label.device)
    ncls = weight.shape[1]
    LaunchProfiler.run('label_conv', 0.0, lambda: L.check(
        L.lib().s2e_label_conv3x3(_dt(out), _p(label), _p(weight), _p(bias), _p(out), n, H, W, h, w, ncls, cout,
                                  int(relu), _stream()), 's2e_label_conv3x3'),
        nbytes=float(out.numel() * out.element_size() + n * h * w))     # algorithmic: the output written once (+ the label bytes)
    return out


class SpadePrepass:
    """The label convs (mlp_shared, normalization.py:97) and per-class tables of ALL SPADE layers of a generator forward as one
    launch each, at the top of the forward, instead of ~19 + ~6 launches of a few microseconds of work spread over it.

    `with prepass.scope(label, dtype):` around the generator's blocks.  The SPADE ops ask `SpadePrepass.actv(...)` /
    `.table(...)`: served from the batched launch when the layer is in the plan, computed on the spot (and, on a learning
    forward, recorded) otherwise.  The plan is learned on the SECOND forward of a shape -- by then the packed [gamma | beta]
    weights live in the PackPlan's persistent buffers, which the table jobs point at -- and dropped when any tensor it points
    at has moved (optimizer arena rebuilt, .cuda())."""
    current = None

    def __init__(self):
        self.plans = {}            # key -> None (seen once) | dict
        self.key = None
        self.rec = None            # learning forward: {'conv': [...], 'table': [...]}
        self.planned = self.missed = False
        self.pre = {}              # this forward's batched results

    # ------------------------------------------------------------------ scope
    class _Scope:
        def __init__(self, owner, label, dtype):
            self.o, self.label, self.dtype = owner, label, dtype

        def __enter__(self):
            o = self.o
            self.prev = SpadePrepass.current
            SpadePrepass.current = o
            n, H, W = self.label.shape
            o.key = (n, H, W, self.dtype, str(self.label.device))
            o.pre, o.rec, o.planned, o.missed = {}, None, False, False
            if o.key not in o.plans:
                o.plans[o.key] = None                              # first forward of this shape: only note it
            elif o.plans[o.key] is None:
                o.rec = {'conv': [], 'table': []}                  # second: learn
            else:
                plan = o.plans[o.key]
                if any(t.data_ptr() != ptr for t, ptr in plan['pins']):
                    o.plans[o.key] = None                          # something moved: learn again next time
                else:
                    o._run(plan, self.label, self.dtype)
                    o.planned = True
            return o

        def __exit__(self, *exc):
            o = self.o
            if o.rec is not None and exc[0] is None and (o.rec['conv'] or o.rec['table']):
                o.plans[o.key] = o._build(o.rec, self.label, self.dtype)
            elif o.planned and o.missed:
                o.plans[o.key] = None
            o.rec, o.pre, o.key = None, {}, None
            SpadePrepass.current = self.prev
            return False

    def scope(self, label, dtype):
        return SpadePrepass._Scope(self, label, dtype)

    # ------------------------------------------------------------------ plan
    def _build(self, rec, label, dtype):
        lib = L.lib()
        n = label.shape[0]
        dev = label.device
        dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
        esz = 2 if dtype == torch.bfloat16 else 4
        plan = {'pins': [], 'conv': None, 'table': None}
        if rec['conv']:
            jobs = (L.LabelConvJob * len(rec['conv']))()
            off, entries = 0, []
            for i, (w_sh, b_sh, h, w, cout, relu) in enumerate(rec['conv']):
                jobs[i].weight, jobs[i].bias = w_sh.data_ptr(), (b_sh.data_ptr() if b_sh is not None else None)
                jobs[i].out_off, jobs[i].h, jobs[i].w, jobs[i].cout, jobs[i].relu = off, h, w, cout, int(relu)
                entries.append((w_sh.data_ptr(), h, w, off, (n, h, w, cout)))
                off += (n * h * w * cout * esz + 255) // 256 * 256
                plan['pins'] += [(w_sh, w_sh.data_ptr())] + ([(b_sh, b_sh.data_ptr())] if b_sh is not None else [])   # (the objects the ops were handed: a re-homed Parameter shows here)
            nb = lib.s2e_label_conv_block_map(dt, C.byref(jobs), len(rec['conv']), n, None)
            bm = np.zeros(3 * nb, dtype=np.int32)
            lib.s2e_label_conv_block_map(dt, C.byref(jobs), len(rec['conv']), n, bm.ctypes.data)
            plan['conv'] = dict(jobs=torch.from_numpy(np.frombuffer(bytes(jobs), dtype=np.uint8).copy()).to(dev),
                                map=torch.from_numpy(bm).to(dev), nb=int(nb), bytes=off, entries=entries,
                                ncls=rec['conv'][0][0].shape[1])
        if rec['table']:
            jobs = (L.ClassTableJob * len(rec['table']))()
            off, entries = 0, []
            for i, (w_sh, b_sh, wp, b_f, c, nh, ncls) in enumerate(rec['table']):
                jobs[i].w_sh, jobs[i].b_sh, jobs[i].w_packed, jobs[i].bias = w_sh.data_ptr(), b_sh.data_ptr(), wp.data_ptr(), b_f.data_ptr()
                jobs[i].table_off, jobs[i].nh, jobs[i].C = off, nh, c
                entries.append((wp.data_ptr(), off, (ncls, 5, 5, 2 * c)))
                off += ncls * 25 * 2 * c * 4
                plan['pins'] += [(w_sh, w_sh.data_ptr()), (b_sh, b_sh.data_ptr()), (wp, wp.data_ptr()), (b_f, b_f.data_ptr())]
            nb = lib.s2e_class_table_block_map(C.byref(jobs), len(rec['table']), None)
            bm = np.zeros(2 * nb, dtype=np.int32)
            lib.s2e_class_table_block_map(C.byref(jobs), len(rec['table']), bm.ctypes.data)
            plan['table'] = dict(jobs=torch.from_numpy(np.frombuffer(bytes(jobs), dtype=np.uint8).copy()).to(dev),
                                 map=torch.from_numpy(bm).to(dev), nb=int(nb), bytes=off, entries=entries, ncls=rec['table'][0][6])
        return plan

    def _run(self, plan, label, dtype):
        n, H, W = label.shape
        dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
        pc = plan['conv']
        if pc is not None:
            buf = torch.empty(pc['bytes'], dtype=torch.uint8, device=label.device)
            LaunchProfiler.run('label_conv', 0.0, lambda: L.check(
                L.lib().s2e_label_conv3x3_batch(dt, _p(label), _p(pc['jobs']), _p(pc['map']), pc['nb'], _p(buf), n, H, W, pc['ncls'],
                                                _stream()), 's2e_label_conv3x3_batch'), nbytes=float(pc['bytes']))
            for ptr, h, w, off, shape in pc['entries']:
                numel = shape[0] * shape[1] * shape[2] * shape[3]
                self.pre[('a', ptr, h, w)] = buf[off:off + numel * (2 if dtype == torch.bfloat16 else 4)].view(dtype).view(shape)
        pt = plan['table']
        if pt is not None:
            tb = torch.empty(pt['bytes'] // 4, dtype=torch.float32, device=label.device)
            L.check(L.lib().s2e_spade_class_table_batch(dt, _p(pt['jobs']), _p(pt['map']), pt['nb'], _p(tb), pt['ncls'], _stream()),
                    's2e_spade_class_table_batch')
            for ptr, off, shape in pt['entries']:
                self.pre[('t', ptr)] = tb[off // 4:off // 4 + shape[0] * 25 * shape[3]].view(shape)

    # ------------------------------------------------------------------ what the SPADE ops call
    @classmethod
    def actv(cls, label, w_sh, b_sh, n, H, W, h, w, nh, dtype):
        """ReLU(mlp_shared(one-hot label at (h, w))) -- label_conv3x3_raw(..., relu=True) -- from the batched launch if planned."""
        wt, bt = _table_of(w_sh), b_sh.detach().float().contiguous()
        cur = cls.current
        if cur is not None:
            hit = cur.pre.get(('a', wt.data_ptr(), h, w))
            if hit is not None and hit.dtype == dtype and hit.shape[0] == n:
                return hit
            if cur.planned:
                cur.missed = True                                   # a planned forward that had to compute on the spot: learn again
            if cur.rec is not None and nh <= 128 and wt.data_ptr() == w_sh.data_ptr() and bt.data_ptr() == b_sh.data_ptr():
                cur.rec['conv'].append((w_sh, b_sh, h, w, nh, True))
        return label_conv3x3_raw(label, wt, bt, n, H, W, h, w, nh, True, dtype)

    @classmethod
    def table(cls, x_dtype, w_sh, b_sh, wp, b_f, ncls, nh, c, stable):
        """The per-class [gamma | beta] table of a label-sparse layer (s2e_spade_class_table), from the batched launch if planned.
        stable: wp and b_f are persistent buffers (PackPlan pack / arena view), i.e. worth pointing a job at."""
        wt, bt = _table_of(w_sh), b_sh.detach().float().contiguous()
        cur = cls.current
        if cur is not None:
            hit = cur.pre.get(('t', wp.data_ptr()))
            if hit is not None and tuple(hit.shape) == (ncls, 5, 5, 2 * c):
                return hit
            if cur.planned and stable:
                cur.missed = True
            if cur.rec is not None and stable and wt.data_ptr() == w_sh.data_ptr() and bt.data_ptr() == b_sh.data_ptr():
                cur.rec['table'].append((w_sh, b_sh, wp, b_f, c, nh, ncls))
        table = torch.empty(ncls, 5, 5, 2 * c, dtype=torch.float32, device=wp.device)
        dt = L.S2E_BF16 if x_dtype == torch.bfloat16 else L.S2E_F32
        L.check(L.lib().s2e_spade_class_table(dt, _p(wt), _p(bt), _p(wp), _p(b_f), _p(table), ncls, nh, c, _stream()), 's2e_spade_class_table')
        return table


def onehot_nhwc_raw(label, img, h, w, ncls, cpad, dtype):
    _need(label, img)
    n, H, W = label.shape
    # inside a trainer step the three SPADEs of a block (and both blocks of a resolution) ask for the same
    # one-hot map in their backward: build it once per (label, resolution) per step
    pool = ZeroPool.active()
    key = (label.data_ptr(), label._version, n, H, W, h, w, ncls, cpad, dtype) if (img is None and pool is not None) else None
    if key is not None and key in pool.step_cache:
        return pool.step_cache[key]
    out = torch.empty(n, h, w, cpad, dtype=dtype, device=label.device)
    L.check(L.lib().s2e_onehot_nhwc(_dt(out), _p(label), _p(img), _p(out), n, H, W, h, w, ncls, cpad, _stream()),
            's2e_onehot_nhwc')
    if key is not None:
        pool.step_cache[key] = out
    return out


def _table_of(weight):
    """The fp32 OIHW weight of a label conv, as s2e_label_conv3x3 takes it."""
    w = weight.detach()
    return w if (w.dtype == torch.float32 and w.is_contiguous()) else w.float().contiguous()


# ------------------------------------------------------------------------------ label-map convs

class LabelConvFn(torch.autograd.Function):
    """conv3x3(one_hot(nearest_down(label))) (+ReLU) -- generator.py:72-73 (fc) and the
    SPADE mlp_shared (normalization.py:85-88)."""

    @staticmethod
    def forward(ctx, label, weight, bias, h, w, relu, dtype):
        n, H, W = label.shape
        out = label_conv3x3_raw(label, _table_of(weight), bias.detach().float().contiguous(), n, H, W, h, w,
                                weight.shape[0], relu, dtype)
        ctx.cfg = (h, w, relu)
        ctx.wdst, ctx.bdst = _grad_dst(weight), _grad_dst(bias)
        ctx.save_for_backward(label, weight, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        label, weight, out = ctx.saved_tensors
        h, w, relu = ctx.cfg
        g = g.contiguous()
        if relu:
            raise NotImplementedError('use SpadeParamFn for the ReLU variant (mask fused into the dgrad)')
        cout, ncls = weight.shape[0], weight.shape[1]
        oh = onehot_nhwc_raw(label, None, h, w, ncls, 8, g.dtype)
        dwp, gb = conv2d_wgrad_raw(oh, g, 3, 3, 1, 1, ACT_NONE, True, ctx.bdst, defer_ok=ctx.wdst is not None and ctx.bdst is not None)
        if ctx.wdst is not None:
            unpack_weight_grad_into(dwp, ctx.wdst, cout, ncls, 3, 3, 8)
            return None, None, gb, None, None, None, None
        return None, _unpack_dw(dwp, cout, ncls, 3, 3, 8), gb, None, None, None, None


def label_conv3x3(label, weight, bias, h, w, relu, dtype):
    return LabelConvFn.apply(label, weight, bias, h, w, relu, dtype)


class SpadeParamFn(torch.autograd.Function):
    """gb = conv3x3(ReLU(conv3x3(one_hot(label_h)))) : the SPADE branch that produces
    [gamma | beta] (normalization.py:97-101) as one 2C-channel tensor.
    When mlp_gamma / mlp_beta weights (and biases) sit back to back in the optimizer arena (Pix2PixModel
    orders them so), [W_gamma; W_beta] is a zero-copy view and the backward accumulates straight into
    the gradient arena; otherwise they are concatenated and the gradients go back through autograd.
    Backward: the ReLU mask is fused into the data-gradient epilogue; the mlp_shared weight gradient is an
    MFMA wgrad against the (tiny) 8-channel one-hot map; bias gradients come out of the wgrad kernels."""

    @staticmethod
    def forward(ctx, label, w_sh, b_sh, w_g, b_g, w_b, b_b, h, w, dtype):
        n, H, W = label.shape
        nh, C = w_sh.shape[0], w_g.shape[0]
        fused, w_gb, b_gb = _gb_operands(w_g, b_g, w_b, b_b, nh)
        actv = SpadePrepass.actv(label, w_sh, b_sh, n, H, W, h, w, nh, dtype)
        plan = packing.current()
        wp = packed_weight(w_gb, dtype, nh, False, None, plan, stable=fused)
        ctx.plan, ctx.plan_gen, ctx.fused = plan, (plan.generation if plan is not None else None), fused
        gb = conv2d_raw(actv, wp, b_gb.float().contiguous(), None, None, (h, w, 2 * C), 3, 3, 1, 1)
        ctx.cfg = (h, w, C)
        _gb_grad_targets(ctx, fused, w_sh, b_sh, w_g, b_g, w_b, b_b, nh)
        ctx.save_for_backward(label, w_sh, w_gb, actv)
        return gb

    @staticmethod
    def backward(ctx, ggb):
        label, w_sh, w_gb, actv = ctx.saved_tensors
        return (None,) + _spade_param_grads(ctx, ggb.contiguous(), label, w_sh, w_gb, actv) + (None, None, None)


def _gb_operands(w_g, b_g, w_b, b_b, nh):
    """[W_gamma; W_beta] and [b_gamma; b_beta] as single tensors: zero-copy views when the four parameters sit back to
    back in the optimizer arena (Pix2PixModel orders them so), concatenated copies otherwise.  -> (fused, w_gb, b_gb)"""
    C = w_g.shape[0]
    # (asked ~90 times per step with the same tensors: the answer for parameters that alias an arena is memoised, keyed on the
    # storage addresses -- the views are of the arena and stay valid as long as the parameters stay where they are)
    key = (w_g.data_ptr(), w_b.data_ptr(), b_g.data_ptr(), b_b.data_ptr(), w_g.stride(), nh, C)
    memo = getattr(w_g, '_s2e_gb_memo', None)               # (kept ON the parameter: it lives exactly as long as the arena it views)
    if memo is not None and memo[0] == key:
        return memo[1]
    fused = _adjacent(w_g, w_b) and _adjacent(b_g, b_b)
    if fused:
        res = (True, _span2(w_g, (2 * C, nh, 3, 3)), _span2(b_g, (2 * C,)))
        try:
            w_g._s2e_gb_memo = (key, res)
        except AttributeError:                               # (a plain tensor slot-less view: no memo)
            pass
        return res
    return False, torch.cat([w_g.detach(), w_b.detach()], 0), torch.cat([b_g.detach(), b_b.detach()], 0)


def _gb_grad_targets(ctx, fused, w_sh, b_sh, w_g, b_g, w_b, b_b, nh):
    """Where the backward may accumulate the SPADE branch's parameter gradients directly (see _grad_dst)."""
    C = w_g.shape[0]
    gwg, gwb, gbg, gbb = _grad_dst(w_g), _grad_dst(w_b), _grad_dst(b_g), _grad_dst(b_b)
    ctx.gb_dst = (_span2(gwg, (2 * C, nh, 3, 3)), _span2(gbg, (2 * C,))) \
        if (fused and _adjacent(gwg, gwb) and _adjacent(gbg, gbb)) else None
    ctx.sh_dst = (_grad_dst(w_sh), _grad_dst(b_sh))


# smallest map side the label-sparse backward takes: at 64 x 64 only the 2 x 2 inner rectangles of 16 can be uniform-interior at all (none
# is, on the bench's maps) and the lists + sums are pure overhead; 96 / 192 measured 18.43-18.48 / 18.40-18.44 ms against 18.44-18.60 at 48
_SPARSE_BWD_MIN = 96      # (round 4 sweep, ms per step at 48 / 96 / 192: 18.44-18.60 / 18.43-18.48 / 18.40-18.44; a constant since round 5)


def _sparse_bwd_lists(ctx, g, h, w, cch, nh, ncls):
    """(cls, work_list, ui_list, counts) for the label-sparse backward of this SPADE layer, or None: the forward ran label-sparse
    on 16 x 16 rectangles (ctx.rects), a trainer step is open (zeroed scratch, deferred flush), mlp_shared's gradients go straight
    to the arena, and the data-gradient's kernel takes a rectangle list.  S2E_SPADE_SPARSE_BWD=0 / S2E_DETERMINISTIC=1 (the sums
    use float atomics): off."""
    rects = getattr(ctx, 'rects', None)
    pool = ZeroPool.active()
    if switches.SPARSE_BWD_OFF or rects is None or pool is None or g.dtype != torch.bfloat16 or ncls > 4 or nh > 128:
        return None
    cls, _, _, _, tw, th = rects
    wdst, bdst = ctx.sh_dst
    if tw != 16 or th != 16 or h < _SPARSE_BWD_MIN or w < _SPARSE_BWD_MIN or 2 * cch not in (128, 256, 512, 1024) or wdst is None or bdst is None or not wdst.is_contiguous():
        return None
    n = g.shape[0]
    d, _ = _conv_plan(False, _dt(g), n, h, w, 2 * cch, h, w, nh, 3, 3, 1, 1, 1, ACT_NONE, ACT_NONE, AUX_RELU_MASK)
    key = ('rects_supported', n, h, w, cch, nh)
    ok = _CONV_STATS_SLOTS.get(key)
    if ok is None:
        ok = _CONV_STATS_SLOTS[key] = bool(L.lib().s2e_conv2d_rects_supported(_dt(g), _byref(d)))
    if not ok:
        return None
    ck = ('rects_bwd', cls.data_ptr(), h, w)
    ent = pool.step_cache.get(ck)
    if ent is None:
        lists = torch.empty(2, cls.numel(), dtype=torch.int32, device=g.device)
        counts = torch.empty(2, dtype=torch.int32, device=g.device)
        L.check(L.lib().s2e_label_rect_lists_bwd(_p(cls), n, h // 16, w // 16, _p(lists[0]), _p(lists[1]), _p(counts), _stream()),
                's2e_label_rect_lists_bwd')
        ent = pool.step_cache[ck] = (cls, lists[0], lists[1], counts)
    return ent


def _sparse_wgrad(g, actv, gb_dst, sp):
    """The [gamma | beta] conv's weight (and bias) gradient over the backward's work rectangles only (s2e_conv2d_wgrad_rects),
    accumulated straight into the channels-last arena slice; the uniform-interior rectangles' part -- rank one per class -- is
    added by s2e_spade_uniform_grads at the flush.  False: this shape's kernel takes no list (the caller runs the dense one)."""
    n, h, w, nh = actv.shape
    c2 = g.shape[-1]
    d, _ = _conv_plan(True, _dt(g), n, h, w, nh, h, w, c2, 3, 3, 1, 1, 0, ACT_NONE, ACT_NONE, AUX_NONE)
    key = ('wgrad_rects_ws', n, h, w, nh, c2)
    wsb = _CONV_STATS_SLOTS.get(key)
    if wsb is None:
        wsb = _CONV_STATS_SLOTS[key] = int(L.lib().s2e_conv2d_wgrad_rects_workspace_bytes(_dt(g), _byref(d)))
    dw, db = gb_dst
    if not wsb or db is None or w_strides_differ(dw):
        return False
    cls, work_list, ui_list, counts = sp
    if GradSink.push_wgrad(actv, g, _cl_rows(dw), db, rects=(work_list, counts)):
        return True
    ws = torch.empty(max(wsb // 4, 4), dtype=torch.float32, device=g.device)
    flops = 2.0 * n * h * w * nh * c2 * 9
    frac = 1.0
    if LaunchProfiler.active():
        frac = float(int(counts[0])) / max(cls.numel(), 1)
    LaunchProfiler.run('conv_wgrad_patch', flops, lambda: L.check(
        L.lib().s2e_conv2d_wgrad_rects(_dt(g), _p(actv), _p(g), _p(_cl_rows(dw)), _p(db), _byref(d), _p(work_list), _p(counts),
                                       _p(ws), wsb, _stream()), 's2e_conv2d_wgrad_rects'),
        tag=lambda: 'W n%d %dx%d c%d->%d k3 s1 sparse' % (n, h, w, nh, c2),
        nbytes=lambda: float((actv.numel() + g.numel()) * frac * g.element_size()), executed=flops * frac)
    return True


def w_strides_differ(dw):
    """The rank-1 update walks dW with the weight's strides: both must be the dense channels-last (co, ky, kx, ci) layout."""
    co, ci, kh, kw = dw.shape
    return tuple(dw.stride()) != (kh * kw * ci, 1, kw * ci, ci)


def _spade_param_grads(ctx, g, label, w_sh, w_gb, actv):
    """Backward of gb = conv3x3(ReLU(conv3x3(one_hot(label)))) given g = d/d[gamma | beta] (N,h,w,2C):
    -> (gw_sh, gb_sh, gw_g, gb_g, gw_b, gb_b), None where the gradient went straight into the arena.
    The ReLU mask is fused into the data-gradient epilogue; the mlp_shared weight gradient is an MFMA wgrad against the
    (tiny) 8-channel one-hot map; bias gradients come out of the wgrad kernels."""
    h, w, C = ctx.cfg
    c2, nh = w_gb.shape[0], w_gb.shape[1]
    ncls = w_sh.shape[1]
    gw_g = gb_g = gw_b = gb_b = gw_sh = gb_sh = None
    sp = _sparse_bwd_lists(ctx, g, h, w, C, nh, ncls)
    uni_gb = None                                            # (dW, db) of [gamma | beta] that take the uniform rectangles' rank-1 part
    if sp is not None and ctx.gb_dst is not None and _cl_dense(ctx.gb_dst[0]) and _sparse_wgrad(g, actv, ctx.gb_dst, sp):
        uni_gb = ctx.gb_dst
    elif ctx.gb_dst is not None and _cl_dense(ctx.gb_dst[0]):     # channels-last arena: straight into [dW_gamma; dW_beta]
        conv2d_wgrad_raw(actv, g, 3, 3, 1, 1, ACT_NONE, True, ctx.gb_dst[1], dw_out=_cl_rows(ctx.gb_dst[0]), defer_ok=True)
    elif ctx.gb_dst is not None:
        dwp, _ = conv2d_wgrad_raw(actv, g, 3, 3, 1, 1, ACT_NONE, True, ctx.gb_dst[1], defer_ok=True)
        unpack_weight_grad_into(dwp, ctx.gb_dst[0], c2, nh, 3, 3, nh)
    else:
        dwp, gb_gb = conv2d_wgrad_raw(actv, g, 3, 3, 1, 1, ACT_NONE, True)
        gw_gb = _unpack_dw(dwp, c2, nh, 3, 3, nh)
        gw_g, gw_b, gb_g, gb_b = gw_gb[:C], gw_gb[C:], gb_gb[:C], gb_gb[C:]
    wpt = packed_weight(w_gb, g.dtype, nh, True, None, ctx.plan, ctx.plan_gen, stable=ctx.fused)
    if sp is not None:
        # label-sparse backward (csrc/spade_sparse_bwd.hip): the data gradient -- only ever used for mlp_shared's gradients -- on the
        # rectangles that cross a label boundary (or touch the image border); the uniform-interior ones contribute through nine
        # shifted sums of dgb per class, folded into mlp_shared's gradients when the step's sink flushes
        cls, work_list, ui_list, counts = sp
        n = g.shape[0]
        # d(actv) exists on the work rectangles only.  When its one reader -- mlp_shared's weight gradient -- is queued with the same
        # rectangle list (round 6) the rest is never read: no zero-fill (0.5 GB per G step at the bench's size).  Otherwise zeros.
        lazy = not switches.C8_SPARSE_OFF and nh == 128 and GradSink.c8_would_queue(g.dtype, h, w, ctx.sh_dst[0], ctx.sh_dst[1]) and not ((h | w) & 15)
        dactv = torch.empty(n, h, w, nh, dtype=g.dtype, device=g.device) if lazy else \
            ZeroPool.take(n * h * w * nh, g.dtype, g.device).view(n, h, w, nh)               # (zero where no conv runs)
        c8_rects = (work_list, counts) if lazy else None
        d, _ = _conv_plan(False, _dt(g), n, h, w, c2, h, w, nh, 3, 3, 1, 1, 1, ACT_NONE, ACT_NONE, AUX_RELU_MASK)
        flops = 2.0 * n * h * w * c2 * nh * 9
        frac = 1.0
        if LaunchProfiler.active():
            frac = float(int(counts[0])) / max(cls.numel(), 1)
        LaunchProfiler.run('conv_patch', flops, lambda: L.check(
            L.lib().s2e_conv2d_rects(_dt(g), _p(g), _p(wpt), None, None, _p(actv), _p(dactv), _byref(d), _p(work_list), _p(counts), _stream()),
            's2e_conv2d_rects'), tag=lambda: 'D n%d %dx%d c%d->%d k3 s1 sparse' % (n, h, w, nh, c2),
            nbytes=lambda: float((g.numel() + 2 * dactv.numel()) * frac * g.element_size() + wpt.numel() * g.element_size()), executed=flops * frac)
        R = ZeroPool.take(L.UNI_REPLICAS * ncls * 9 * c2, torch.float32, g.device)
        A = ZeroPool.take(ncls * nh, torch.float32, g.device)
        LaunchProfiler.run('spade_uniform_bwd', 0.0, lambda: L.check(
            L.lib().s2e_spade_uniform_sums(_dt(g), _p(g), n, h, w, c2, ncls, _p(cls), _p(ui_list), _p(counts), _p(R), _stream()),
            's2e_spade_uniform_sums'), nbytes=float(g.numel() * g.element_size() * (1.0 - frac) * 1.27))
        wdst, bdst = ctx.sh_dst
        assert uni_gb is None or tuple(w_gb.stride()) == tuple(uni_gb[0].stride()), 'weight and gradient arenas are laid out alike'
        ZeroPool.active().sink.uni.append((R, A, w_gb.detach(), w_sh.detach(), ctx.b_sh_f, wdst, bdst,
                                           uni_gb[0] if uni_gb is not None else None, uni_gb[1] if uni_gb is not None else None, c2, nh, ncls,
                                           int(g.dtype == torch.bfloat16)))
    else:
        c8_rects = None
        dactv = conv2d_raw(g, wpt, None, None, actv, (h, w, nh), 3, 3, 1, 1, True, ACT_NONE, ACT_NONE, AUX_RELU_MASK)
    # (a streaming class-bucket kernel for this gradient was tried twice -- LDS float atomics, then per-wave
    # queues of boundary pixels -- and lost to the MFMA wgrad against the 8-channel one-hot map: 1.7 vs 0.75 ms
    # per step; see DESIGN.md "tried and dropped")
    oh = onehot_nhwc_raw(label, None, h, w, ncls, 8, g.dtype)
    wdst, bdst = ctx.sh_dst
    if GradSink.push_c8(oh, dactv, wdst, bdst, ncls, c8_rects):    # inside a trainer step: all mlp_shared gradients in one launch, later
        return gw_sh, gb_sh, gw_g, gb_g, gw_b, gb_b
    if c8_rects is not None:
        raise L.Seg2EyeHipError('label-sparse SPADE backward: d(actv) was left undefined outside the work rectangles but the 8-channel weight gradient was not queued')
    dwp, gb_sh = conv2d_wgrad_raw(oh, dactv, 3, 3, 1, 1, ACT_NONE, True, bdst, defer_ok=wdst is not None and bdst is not None)
    if wdst is not None:
        unpack_weight_grad_into(dwp, wdst, nh, ncls, 3, 3, 8)
    else:
        gw_sh = _unpack_dw(dwp, nh, ncls, 3, 3, 8)
    return gw_sh, gb_sh, gw_g, gb_g, gw_b, gb_b


def spade_params(label, w_sh, b_sh, w_g, b_g, w_b, b_b, h, w, dtype):
    return SpadeParamFn.apply(label, w_sh, b_sh, w_g, b_g, w_b, b_b, h, w, dtype)


# ------------------------------------------------------------------------------ modulation / IN

class ModulateFn(torch.autograd.Function):
    """SPADE+Style modulation (optionally + LeakyReLU).  stats come from in_stats(x) and may be
    shared between consumers (norm_0 and norm_s normalise the same x, architecture.py:44-59).

    style: this layer's (N,2C) fp32 style code -- or, with `off` given, the generator's (N,S) matrix of ALL
    layers' codes (networks/stylebank.py) of which columns [off, off+2C) are this layer's.  In that mode the
    backward ADDS this layer's style gradient into the same columns of `dbig` (the bank's gradient
    accumulator) and hands autograd nothing for `style`: the bank's own backward picks dbig up."""

    @staticmethod
    def forward(ctx, x, gb, style, stats, lrelu, off=None, dbig=None, batch=False, relay=False):
        _need(x, gb, style, stats)
        n, h, w, c = x.shape
        out = torch.empty_like(x)
        ld = 0 if off is None else style.shape[1]
        sp = style.data_ptr() + 4 * (off or 0)
        LaunchProfiler.run('modulate_fwd', 0.0, lambda: L.check(
            L.lib().s2e_modulate_fwd(_dt(x), NORM_SPADE_STYLE, _p(x), _p(gb), _p(stats), sp, _p(out),
                                     n, h * w, c, int(lrelu), ld, _stream()), 's2e_modulate_fwd'),
            nbytes=float(2 * x.numel() * x.element_size()))           # algorithmic: x read, out written (gamma/beta are not)
        ctx.lrelu, ctx.off, ctx.dbig, ctx.batch, ctx.relay = lrelu, off, dbig, bool(batch), bool(relay)
        ctx.save_for_backward(x, gb, style, stats)
        if relay:
            ctx.set_materialize_grads(False)
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, g, g_relay=None):
        x, gb, style, stats = ctx.saved_tensors
        if g is None:                                       # (relay mode: this layer's own output went unused)
            return g_relay, None, None, None, None, None, None, None, None
        dx, dgb, dstyle = _modulate_grads(ctx, g, g_relay, x, gb, None, style, stats)
        return dx, dgb, dstyle, None, None, None, None, None, None


def _modulate_grads(ctx, g, g_relay, x, gb, fout, style, stats):
    """Backward of the SPADE+Style modulation -> (dx, dgb (N,h,w,2C), dstyle or None).  fout None: gb = [gamma | beta];
    else gb = gamma alone and fout = the forward's output (s2e_modulate_bwd_gamma).
    relay: the OTHER consumers of x hang off the node's second output, so their gradient arrives here first and the
    element-wise pass adds this layer's dx to it in place -- instead of autograd summing two full tensors."""
    n, h, w, c = (fout if fout is not None else x).shape          # (x may be the half-resolution tensor: ctx.x_up_w; dx then is too)
    quad = int(getattr(ctx, 'x_up_w', 0) != 0)
    g = g.contiguous()
    acc = g_relay is not None and g_relay.is_contiguous() and g_relay.dtype == x.dtype
    # (a relayed gradient that a queued weight-gradient job still has to read -- the block's conv_1 with the identity shortcut handed
    #  its gy on as the residual's gradient -- is added to OUT of place: dx = g_relay + this layer's gradient, g_relay untouched)
    apart = acc and GradSink.is_pinned(g_relay)
    dx = g_relay if (acc and not apart) else torch.empty_like(x)
    dgb = torch.empty(n, h, w, 2 * c, dtype=x.dtype, device=x.device)
    if ctx.off is None:
        dstyle = ZeroPool.take(style.numel(), torch.float32, x.device).view(style.shape)
        dsp, ld = dstyle.data_ptr(), 0
    else:
        if ctx.dbig is None:
            raise RuntimeError('ModulateFn: banked style without a gradient accumulator')
        dstyle, dsp, ld = None, ctx.dbig.data_ptr() + 4 * ctx.off, style.shape[1]
    sp = style.data_ptr() + 4 * (ctx.off or 0)
    ws = torch.empty(L.lib().s2e_modulate_bwd_workspace_bytes(_dt(x), n, h * w, c) // 8, dtype=torch.float64, device=x.device)
    mode = (NORM_SPADE_STYLE_BATCH if ctx.batch else NORM_SPADE_STYLE) | (NORM_ACCUMULATE_DX if acc else 0)
    # algorithmic bytes: the two-pass structure is forced by the per-(n,c) sums, so g, x, gamma are read by
    # both passes; dgamma, dbeta and dx are written once: 9 accesses per element of x
    # (x handed over before the upsampling: its two reads and the dx write are a quarter each: 6.75 accesses)
    nb = float((6.75 if quad else 9.0) * n * h * w * c * x.element_size())
    from .. import distributed as sdist
    world = sdist.sync_world_size() if ctx.batch else 1

    def launch(stage, count):
        if apart:
            return L.check(L.lib().s2e_modulate_bwd_relay(_dt(x), mode, _p(g), _p(x), _p(gb), _p(fout), _p(stats), sp, _p(dx), _p(g_relay), _p(dgb), dsp,
                                                          _p(ws), n, h * w, c, int(ctx.lrelu), ld, stage, float(count),
                                                          int(getattr(ctx, 'x_up_w', 0)), quad, _stream()),
                           's2e_modulate_bwd_relay')
        return L.check(L.lib().s2e_modulate_bwd_staged(_dt(x), mode, _p(g), _p(x), _p(gb), _p(fout), _p(stats), sp, _p(dx), _p(dgb), dsp,
                                                       _p(ws), n, h * w, c, int(ctx.lrelu), ld, stage, float(count),
                                                       int(getattr(ctx, 'x_up_w', 0)), quad, _stream()),
                       's2e_modulate_bwd_staged')
    if world == 1:
        LaunchProfiler.run('modulate_bwd', 0.0, lambda: launch(0, 0.0), nbytes=nb)
    else:
        # BatchNorm SPADE under data parallelism: the normalisation's backward sums (S0, S1 per channel) run over the samples of
        # ALL replicas -- one 2*C-double all-reduce between the two passes (the backward half of SURVEY 8 f4's exchange)
        launch(1, 0.0)
        sums = ws[:n * c * 4].view(n, c, 4)
        local = sums[:, :, :2].sum(0)
        glob = sdist.all_reduce_sum_(local.clone())
        sums[0, :, :2] += glob - local                       # the coefficient kernel sums over this replica's samples
        launch(2, float(world) * n * h * w)
    if g_relay is not None and not acc:
        dx = dx + g_relay
    return dx, dgb, dstyle
_SPARSE_MIN_RECTS = 256    # (256^2 and 128^2 at batch 8; below that the classification costs what it saves: round 2; a constant since round 5)


def label_rects(label, h, w, dtype, c, nh, flags=0):
    """Classification of the fused launch's rectangles of the (h, w)-downsampled label map into label-uniform and dense ones
    (s2e_label_rect_classify) -> (cls, dense_list, uni_list, counts, tw, th), or None when the label-sparse form is off
    / not worth it for this size.  Inside a trainer step the result is shared by every SPADE of the resolution (and by both
    forwards' layers: it depends on the label batch only)."""
    if switches.SPARSE_OFF:
        return None
    n, H, W = label.shape
    if h < 8 or w < 8:
        return None
    tw, th = C.c_int(0), C.c_int(0)
    dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
    if not L.lib().s2e_spade_conv_modulate_rect(dt, n, h, w, c, nh, int(flags), C.byref(tw), C.byref(th)):
        return None
    tw, th = tw.value, th.value
    rects = n * ((h + th - 1) // th) * ((w + tw - 1) // tw)
    if rects < _SPARSE_MIN_RECTS and not (flags & 4):
        return None
    pool = ZeroPool.active()
    key = ('rects', label.data_ptr(), label._version, n, H, W, h, w, tw, th)
    if pool is not None and key in pool.step_cache:
        return pool.step_cache[key]
    dev = label.device
    cls = torch.empty(rects, dtype=torch.uint8, device=dev)
    lists = torch.empty(2, rects, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    L.check(L.lib().s2e_label_rect_classify(_p(label), n, H, W, h, w, tw, th, _p(cls), _p(lists[0]), _p(lists[1]), _p(counts), _stream()),
            's2e_label_rect_classify')
    res = (cls, lists[0], lists[1], counts, tw, th)
    if pool is not None:
        pool.step_cache[key] = res
    return res


class SpadeFusedFn(torch.autograd.Function):
    """SpadeParamFn + ModulateFn as ONE forward launch for the layers s2e_spade_conv_modulate takes: the [gamma | beta]
    conv's epilogue applies the SPADE+Style modulation, so gamma and beta never reach HBM (normalization.py:91-105,
    163-169, 184-192 in one kernel).  With gradients on, gamma (C channels) is stored for the backward, which takes the
    LeakyReLU mask from the sign of the saved output; the backward itself is the two-stage one (modulation gradients ->
    [dgamma | dbeta] -> the conv's weight / data gradients)."""

    @staticmethod
    def forward(ctx, x, label, w_sh, b_sh, w_g, b_g, w_b, b_b, style, stats, lrelu, off, dbig, batch, relay, flags, grad_mode):
        _need(x, style, stats)
        n, h, w, c = x.shape
        # flags & 8: x is the tensor BEFORE the block's nearest 2x upsampling.  The launches read it at (y/2, x/2); the backward
        # does too and returns the gradient w.r.t. THIS tensor (the 2 x 2 sums: the upsampling's backward folded in as well).
        # Neither the upsampled tensor nor its gradient ever exists.
        up = bool(flags & 8)
        xr = x
        if up:
            h, w = 2 * h, 2 * w
            if batch:
                raise ValueError('spade_style_fused: the folded upsampling (flags 8) is not built for BatchNorm SPADE')
        _, H, W = label.shape
        nh = w_sh.shape[0]
        dtype = x.dtype
        fused, w_gb, b_gb = _gb_operands(w_g, b_g, w_b, b_b, nh)
        actv = SpadePrepass.actv(label, w_sh, b_sh, n, H, W, h, w, nh, dtype)
        plan = packing.current()
        wp = packed_weight(w_gb, dtype, nh, False, None, plan, stable=fused)
        wp_persistent = fused and plan is not None and plan.lookup(w_gb, dtype, nh, False) is wp
        ctx.plan, ctx.plan_gen, ctx.fused = plan, (plan.generation if plan is not None else None), fused
        # (inside forward() grad mode is always off and needs_input_grad is set under torch.no_grad() too: whether a backward can
        # follow is the CALLER's grad mode, handed in.  Without it the D step's no-grad generator forward stored gamma for nothing.)
        train = bool(grad_mode) and any(ctx.needs_input_grad)
        out = torch.empty(n, h, w, c, dtype=dtype, device=x.device)
        gamma = torch.empty_like(out) if train else None
        ld = 0 if off is None else style.shape[1]
        sp = style.data_ptr() + 4 * (off or 0)
        b_f = b_gb.float().contiguous()
        flops = 2.0 * n * h * w * nh * 2 * c * 9
        sparse = None if (flags & 2) else label_rects(label, h, w, x.dtype, c, nh, flags)
        if sparse is None:
            LaunchProfiler.run('conv_patch', flops, lambda: L.check(
                L.lib().s2e_spade_conv_modulate(_dt(x), _p(actv), _p(wp), _p(b_f), _p(xr), _p(stats), sp, ld,
                                                _p(out), _p(gamma), n, h, w, c, nh, int(lrelu), int(flags), _stream()),
                's2e_spade_conv_modulate'),
                tag=lambda: 'F n%d %dx%d c%d->%d k3 s1 +mod%s' % (n, h, w, nh, 2 * c, '' if train else ' nograd'),
                # algorithmic bytes: actv, packed w, x in; out (and gamma when it is kept) out
                nbytes=lambda: float((actv.numel() + wp.numel() + xr.numel() + out.numel() * (2 if train else 1)) * x.element_size()))
        else:
            # label-sparse: the conv runs on the rectangles that cross a label boundary only; the others read gamma | beta from
            # the per-class table (s2e_spade_class_table: this layer's [gamma | beta] branch on one-class maps, all 25 border cases)
            cls, dense_list, uni_list, counts, tw, th = sparse
            ncls = w_sh.shape[1]
            table = SpadePrepass.table(x.dtype, w_sh, b_sh, wp, b_f, ncls, nh, c, wp_persistent and b_f.data_ptr() == b_gb.data_ptr())
            frac = 1.0
            if LaunchProfiler.active():                         # executed work of this launch (a sync: profiling runs only)
                rects = cls.numel()
                frac = float(int(counts[0])) / max(rects, 1)
            LaunchProfiler.run('conv_patch', flops, lambda: L.check(
                L.lib().s2e_spade_conv_modulate_sparse(_dt(x), _p(actv), _p(wp), _p(b_f), _p(xr), _p(stats), sp, ld, _p(out), _p(gamma),
                                                       n, h, w, c, nh, int(lrelu), int(flags), _p(dense_list), _p(counts), _stream()),
                's2e_spade_conv_modulate_sparse'),
                tag=lambda: 'F n%d %dx%d c%d->%d k3 s1 +mod sparse%s' % (n, h, w, nh, 2 * c, '' if train else ' nograd'),
                nbytes=lambda: float((actv.numel() + xr.numel() + out.numel() * (2 if train else 1)) * frac * x.element_size() + wp.numel() * x.element_size()),
                executed=flops * frac)
            LaunchProfiler.run('modulate_fwd', 0.0, lambda: L.check(
                L.lib().s2e_spade_modulate_uniform(_dt(x), _p(xr), _p(stats), sp, ld, _p(table), _p(cls), _p(uni_list), _p(counts), _p(out),
                                                   _p(gamma), n, h, w, c, tw, th, int(lrelu), int(bool(flags & 8)), _stream()), 's2e_spade_modulate_uniform'),
                nbytes=float((xr.numel() + out.numel() * (2 if train else 1)) * (1.0 - frac) * x.element_size()))
        ctx.cfg = (h, w, c)
        ctx.lrelu, ctx.off, ctx.dbig, ctx.batch, ctx.relay = lrelu, off, dbig, bool(batch), bool(relay)
        if train:
            _gb_grad_targets(ctx, fused, w_sh, b_sh, w_g, b_g, w_b, b_b, nh)
            ctx.rects = sparse                                   # (the label-sparse backward reuses the forward's classification)
            ctx.b_sh_f = b_sh.detach().float().contiguous()
            ctx.x_up_w = w if up else 0
            ctx.save_for_backward(xr, label, w_sh, w_gb, actv, gamma, out, style, stats)
        if relay:
            ctx.set_materialize_grads(False)
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, g, g_relay=None):
        x, label, w_sh, w_gb, actv, gamma, out, style, stats = ctx.saved_tensors
        nn_ = (None,) * 8
        if g is None:
            return (g_relay,) + (None,) * 16
        dx, dgb, dstyle = _modulate_grads(ctx, g, g_relay, x, gamma, out, style, stats)
        gw_sh, gb_sh, gw_g, gb_g, gw_b, gb_b = _spade_param_grads(ctx, dgb, label, w_sh, w_gb, actv)
        return (dx, None, gw_sh, gb_sh, gw_g, gb_g, gw_b, gb_b, dstyle) + (None,) * 8


def spade_fused_supported(x, nh, flags=0):
    """Does s2e_spade_conv_modulate take this layer (x: (N,h,w,C) NHWC, nh = mlp_shared's width; flags & 8: x is the
    half-resolution tensor of a folded upsampling)?"""
    n, h, w, c = x.shape
    if flags & 8:
        h, w = 2 * h, 2 * w
    if switches.FUSED_OFF:
        return False
    return bool(L.lib().s2e_spade_conv_modulate_supported(_dt(x), n, h, w, c, nh, int(flags)))


def spade_style_fused(x, label, w_sh, b_sh, w_g, b_g, w_b, b_b, style, stats, lrelu, off=None, dbig=None, batch=False,
                      relay=False, flags=0):
    """SPADE+Style block forward in one conv launch (see SpadeFusedFn); same arguments as spade_params +
    spade_style_modulate."""
    if off is None:
        style = style.float().contiguous()
    return SpadeFusedFn.apply(x, label, w_sh, b_sh, w_g, b_g, w_b, b_b, style, stats, lrelu, off, dbig, batch, relay, flags,
                              torch.is_grad_enabled())


def spade_style_modulate(x, gb, style, stats, lrelu, off=None, dbig=None, batch=False, relay=False):
    """batch: `stats` are batch statistics (BatchNorm SPADE) -- the same row for every sample.
    relay: returns (out, x') with x' an alias of x; feed x' to the OTHER consumers of x (the block's second SPADE, the
    residual) and their gradient reaches this layer's backward, which adds its own dx into it in place."""
    if off is None:
        style = style.float().contiguous()
    return ModulateFn.apply(x, gb, style, stats, lrelu, off, dbig, batch, relay)


class InstanceNormFn(torch.autograd.Function):
    """InstanceNorm2d(affine=False) (+ LeakyReLU 0.2): discriminator.py:91-94, encoder.py layers."""

    @staticmethod
    def forward(ctx, x, lrelu):
        _need(x)
        n, h, w, c = x.shape
        ws = torch.empty(L.lib().s2e_in_stats_workspace_bytes(_dt(x), n, h * w, c) // 8, dtype=torch.float64, device=x.device)
        stats = torch.empty(n, c, 2, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        LaunchProfiler.run('modulate_fwd', 0.0, lambda: L.check(
            L.lib().s2e_instance_norm_fwd(_dt(x), _p(x), _p(out), _p(stats), _p(ws), n, h * w, c, IN_EPS, int(lrelu), _stream()),
            's2e_instance_norm_fwd'),
            nbytes=float(3 * x.numel() * x.element_size()))           # algorithmic: x read for the statistics and again to normalise, out written
        ctx.lrelu = lrelu
        ctx.save_for_backward(x, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        x, stats = ctx.saved_tensors
        n, h, w, c = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        ws = torch.empty(L.lib().s2e_modulate_bwd_workspace_bytes(_dt(x), n, h * w, c) // 8, dtype=torch.float64, device=x.device)
        LaunchProfiler.run('modulate_bwd', 0.0, lambda: L.check(
            L.lib().s2e_instance_norm_bwd(_dt(x), _p(g), _p(x), _p(stats), _p(dx), _p(ws), n, h * w, c, int(ctx.lrelu), _stream()),
            's2e_instance_norm_bwd'),
            nbytes=float(5 * x.numel() * x.element_size()))           # algorithmic: g, x read twice (sums, then dx), dx written
        return dx, None


def instance_norm(x, lrelu=False):
    return InstanceNormFn.apply(x, lrelu)
