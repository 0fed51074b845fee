"""PyTorch-ROCm custom ops over the C-ABI kernels (include/seg2eye_hip.h).

torch is plumbing here: it owns device memory (caching allocator), the stream and
the autograd tape; every forward/backward body below is one or more HIP kernel
launches through ctypes on torch's current stream.  Internal activations are
NHWC-contiguous 4-D tensors (N, H, W, C) in the compute dtype (bf16 or fp32).

There is no CPU path: calling an op with a non-CUDA tensor raises.
"""
import ctypes as C
import os

import numpy as np

import torch

from . import _lib as L
from . import packing
from ._lib import (NORM_SPADE_STYLE_BATCH, NORM_ACCUMULATE_DX, ConvDesc, ACT_NONE, ACT_LRELU, ACT_TANH, AUX_NONE, AUX_RELU_MASK, AUX_LRELU_GRAD,
                   NORM_SPADE_STYLE, NORM_PLAIN_IN, LOSS_NEG_MEAN, LOSS_HINGE_REAL, LOSS_HINGE_FAKE, LOSS_L1)

IN_EPS = 1e-5      # nn.InstanceNorm2d default (models/networks/normalization.py:41,73)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return L.S2E_BF16
    if t.dtype == torch.float32:
        return L.S2E_F32
    raise TypeError('seg2eye_amd ops take bf16 or fp32 tensors, got %s' % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    # the raw handle of torch's current stream, without building a torch.cuda.Stream object per launch (that path resolves
    # the device index through four Python layers: 2.4 ms of host time per eager step of ~1000 launches)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _need(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise L.Seg2EyeHipError('seg2eye_amd ops run on the GPU only (got a %s tensor); '
                                        'there is no CPU fallback' % t.device)
            if not t.is_contiguous():
                raise L.Seg2EyeHipError('seg2eye_amd ops need contiguous tensors')


# ------------------------------------------------------------------------------ per-launch timing
class LaunchProfiler:
    """Optional HIP-event timing of the kernels, per C-ABI call, on the stream they are launched on (torch's current
    stream).  bench.py / tools create one, install it with `LaunchProfiler.install(p)` and read `p.summary()`; with none
    installed (the default) `run` is a plain call.  Families: the MFMA kernels by `s2e_conv2d_kernel_kind` (conv_patch /
    conv_igemm / conv_small and the weight-gradient ones), the HBM-bound ones by entry point (in_stats, modulate_fwd,
    modulate_bwd, label_conv, adam, ...), each with its ALGORITHMIC FLOPs / bytes."""
    current = None        # the installed profiler (one per process at a time: it times whatever runs on this thread)

    def __init__(self):
        self.records = []     # (family, algorithmic_flops, start_event, end_event, tag, algorithmic_bytes, executed_flops)

    @classmethod
    def install(cls, prof):
        cls.current = prof

    @classmethod
    def active(cls):
        return cls.current is not None

    @classmethod
    def run(cls, family, flops, fn, tag='', nbytes=0.0, executed=None):
        """executed: the FLOPs the launch really performs when that is less than its algorithmic count (the label-sparse
        SPADE launch computes only the rectangles that cross a label boundary); default = flops."""
        prof = cls.current
        if prof is None:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        # (family / tag / nbytes may be callables: evaluated only here, i.e. only while a profiler is installed -- formatting a
        # tag string and summing tensor sizes for each of ~1000 launches cost ~2 ms of host time per eager step)
        prof.records.append((family() if callable(family) else family, flops, s, e, tag() if callable(tag) else tag,
                             nbytes() if callable(nbytes) else nbytes, flops if executed is None else executed))
        return r

    def summary(self):
        """family -> dict(launches, flops, ms, bytes); call after a device synchronize."""
        out = {}
        for fam, fl, s, e, _, nb, ex in self.records:
            d = out.setdefault(fam, dict(launches=0, flops=0.0, ms=0.0, bytes=0.0, executed_flops=0.0))
            d['launches'] += 1
            d['flops'] += fl
            d['executed_flops'] += ex
            d['bytes'] += nb
            d['ms'] += s.elapsed_time(e)
        return out

    def reset(self):
        self.records = []


# ------------------------------------------------------------------------------ zero-filled scratch
class ZeroPool:
    """Zero-initialised scratch for the steps of ONE trainer, filled by ONE launch per step.

    A G or D step needs ~200 small zero-filled buffers (packed weight-gradient accumulators, fp64 reduction scratch of
    the statistics / modulation kernels, the spectral-norm dot products).  Zeroing each with its own 4-5 us launch cost
    ~1 ms of a 35 ms step.  Inside `with pool.scope(key)` they are bump-allocated from the pool's device buffer, whose
    used prefix (the high-water mark of earlier scopes with the same key) is cleared by a single fill at scope entry; a
    take beyond the cleared prefix clears its own slice.  The ops ask `ZeroPool.take(...)`, which serves from the pool
    whose scope is open on this process (scopes do not nest) and is plain torch.zeros when none is -- stand-alone ops,
    inference models and tests behave as before.  Everything taken inside a scope must be dead when the pool's next
    scope starts: true for the scratch listed above, NOT for tensors handed to the caller (losses, parameter
    gradients) -- those never come from a pool.  After `freeze()` (a hipGraph holds raw pointers into the buffer) the
    buffer is never re-allocated; overflow falls back to torch.zeros.

    Each Pix2PixTrainer owns its pool (and with it the queue of deferred weight-gradient re-layouts, GradSink): two
    trainers -- or a trainer and an inference model -- in one process share nothing."""
    ALIGN = 256
    _active = None     # the pool whose scope is open
    serial = 0         # scopes begun so far, over all pools (lets per-scope state elsewhere notice a new step)
    _zeroed = {}       # gradient arena base pointer -> (bytes, ZeroPool.serial when optim.FlatAdam.zero_grad last cleared it)

    @classmethod
    def arena_zeroed(cls, flat_g):
        """optim.FlatAdam.zero_grad reports here: this gradient arena is all zeros as of now.  Entries of arenas that no longer
        exist -- their memory now (partly) belongs to this one -- are dropped: a lookup by address must find THIS arena's entry, not
        a dead optimizer's (found in round 5 as a test-order-dependent failure: `arena_touched` marked the stale entry, the live
        arena stayed "fresh" and a chain-ruled gradient was rewritten in place)."""
        base, nbytes = flat_g.data_ptr(), flat_g.numel() * flat_g.element_size()
        for b in [b for b, (nb, _) in cls._zeroed.items() if b != base and b < base + nbytes and base < b + nb]:
            del cls._zeroed[b]
        cls._zeroed[base] = (nbytes, cls.serial)

    @classmethod
    def arena_touched(cls, g):
        """A gradient that is NOT the raw sum of this step's contributions was (or is about to be) accumulated into `g`'s arena
        outside the in-place protocol -- e.g. a spectral-normed layer's chain-ruled gradient through the accumulate path: the
        arena no longer counts as fresh until the next zero_grad."""
        ptr = g.data_ptr()
        for base, (nbytes, _) in cls._zeroed.items():
            if base <= ptr < base + nbytes:
                cls._zeroed[base] = (nbytes, -1)
                return

    @classmethod
    def grad_is_fresh(cls, g):
        """Is `g` (a view of a gradient arena) known to have been ZERO when the open scope began -- cleared by zero_grad after
        the previous scope and before this one?  Only then may a kernel sequence that REWRITES the gradient (spectral norm's
        in-place chain rule) stand in for one that accumulates."""
        if cls._active is None:
            return False
        ptr = g.data_ptr()
        for base, (nbytes, serial) in cls._zeroed.items():
            if base <= ptr < base + nbytes:
                return serial == cls.serial - 1
        return False

    def __init__(self, device):
        self.device = torch.device(device)
        self.buf = None
        self.cap = 0            # bytes allocated
        self.bump = 0           # bytes handed out in the current scope
        self.clean = 0          # [bump, clean) is known to be zero
        self.need = 0           # largest total any scope asked for (drives growth)
        self.high = {}          # key -> high-water mark
        self.key = None
        self.frozen = False
        self.step_cache = {}    # per-scope memo of derived read-only tensors (cleared at scope entry and exit)
        self.tails, self.tail_i = {}, 0     # (scope key, i) -> (live, persistent zero-tailed gradient buffer): _live_tail_buffer
        self.sink = GradSink()

    def scope(self, key):
        return _ZeroScope(self, key)

    def freeze(self):
        self.frozen = True

    def unfreeze(self):
        self.frozen = False

    @classmethod
    def active(cls):
        """The pool whose scope is open, or None."""
        return cls._active

    def _begin(self, key):
        if ZeroPool._active is not None:
            raise RuntimeError('ZeroPool scopes do not nest')
        if not self.frozen and self.need > self.cap:
            self.cap = (int(self.need * 1.25) + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.buf = torch.zeros(self.cap, dtype=torch.uint8, device=self.device)
            self.clean = self.cap
        else:
            hw = min(self.high.get(key, 0), self.cap)
            if hw:
                self.buf[:hw].zero_()
            self.clean = hw
        self.key, self.bump, self.tail_i = key, 0, 0
        self.step_cache = {}
        self.sink.inplace_done = set()
        self.sink.wg_done = set()
        ZeroPool._active = self
        ZeroPool.serial += 1

    def _end(self):
        self.high[self.key] = max(self.high.get(self.key, 0), self.bump)
        self.need = max(self.need, self.bump)
        self.key = None
        self.step_cache = {}
        ZeroPool._active = None

    @classmethod
    def take(cls, numel, dtype, device):
        pool = cls._active
        if pool is None:
            return torch.zeros(numel, dtype=dtype, device=device)
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        off = pool.bump
        end = off + (nbytes + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN
        pool.bump = end                                  # counts overflow too: that is how the pool learns its size
        if end > pool.cap or pool.buf.device != torch.device(device):
            return torch.zeros(numel, dtype=dtype, device=device)
        if end > pool.clean:
            pool.buf[max(off, pool.clean):end].zero_()
            pool.clean = end
        return pool.buf[off:off + nbytes].view(dtype)


class _ZeroScope:
    def __init__(self, pool, key):
        self.pool, self.key = pool, key

    def __enter__(self):
        self.pool._begin(self.key)

    def __exit__(self, *exc):
        try:
            if exc[0] is None:
                self.pool.sink.flush()                       # all queued weight-gradient re-layouts: two launches
            else:
                self.pool.sink.jobs = []
                self.pool.sink.c8 = []
                self.pool.sink.uni = []
                self.pool.sink.wg = []
        finally:
            self.pool._end()
        return False


# ------------------------------------------------------------------------------ deferred weight-gradient re-layout
class GradSink:
    """Inside a ZeroPool scope (a trainer step) the per-layer "packed dW -> OIHW gradient arena" conversions -- plain
    re-layout, or the spectral-norm chain rule dW_orig = (dW - <dW, W_sn> u v^T)/sigma -- are not launched one by one
    (~95 launches of a few microseconds of work each, 1.2 ms per step) but queued and done by TWO launches at scope
    exit (`s2e_weight_grads_batched`).  The packed buffers are ZeroPool slices, alive until the next scope.  The
    device job table is cached by content: in steady state (and always under a hipGraph) every pointer repeats.
    One sink per pool (= per trainer)."""

    def __init__(self):
        self.jobs = []
        self.c8 = []               # deferred 8-channel weight gradients (mlp_shared): (onehot, d actv, dw, db, ncls)
        self.inplace = []          # deferred in-place spectral-norm chain rules (channels-last masters): push_inplace
        self.inplace_done = set()  # gradient slices whose chain rule has already RUN in the open scope (see inplace_allowed)
        self.uni = []              # deferred label-sparse SPADE backward jobs (uniform rectangles' closed-form gradients): push_uniform
        self.wg = []               # deferred patch-resident 3x3 weight gradients (one persistent launch per flush): push_wgrad
        self.wg_done = set()       # dW slices a wgrad flush of the open scope has already written (a later job must ADD to them)
        self.tables = {}
        self.keepalive = None
        self.keep_c8 = None

    @staticmethod
    def push(dwp, dst, cout, cin, taps, cin_pad, w_orig=None, u=None, v=None, sigma=None):
        """True if queued (caller must not touch dst until flush); False: no scope active, do it now."""
        pool = ZeroPool.active()
        if pool is None:
            return False
        pool.sink.jobs.append((dwp, dst, w_orig, u, v, sigma, int(cout), int(cin), int(taps), int(cin_pad)))
        return True

    @staticmethod
    def inplace_allowed(wdst):
        """May the weight-gradient kernel accumulate a SPECTRAL-NORMED layer's raw gradient straight into `wdst` (its channels-
        last .grad), to be rewritten in place by the chain rule g <- g/sigma - (<g, W>/sigma^2) u v^T at the next flush?  The
        rewrite equals "accumulate the chain-ruled gradient" only if wdst held ZEROS before this step's contributions and the
        rule runs ONCE over their sum (it is linear in g).  So: inside a trainer step (ZeroPool scope) whose gradient arena
        zero_grad cleared right before the scope, and not after this slice's rule has already run in the scope (a second
        backward behind a flush).  Everything else -- stand-alone ops, gradient accumulation over several backwards, plain
        .grad tensors -- takes the packed scratch + accumulate path."""
        pool = ZeroPool.active()
        return pool is not None and ZeroPool.grad_is_fresh(wdst) and wdst.data_ptr() not in pool.sink.inplace_done

    @staticmethod
    def push_inplace(g_rows, weight, u, v, sigma, rows, cin, taps):
        """g_rows (rows, taps*cin): a spectral-normed conv's weight gradient, accumulated by the wgrad kernel straight into the
        parameter's channels-last arena slice; weight: weight_orig (same layout).  Applies dW_orig = g/sigma - (<g, W>/sigma^2) u v^T
        in place at the next flush of the step's sink (all layers: one launch pair).  Callers ask `inplace_allowed` first.  A layer
        used twice before a flush queues ONE job: both raw contributions are in g already and the rule is linear."""
        if cin % 8:
            raise ValueError('GradSink.push_inplace: Cin = %d is not a multiple of 8' % cin)
        pool = ZeroPool.active()
        if pool is None:
            raise RuntimeError('GradSink.push_inplace outside a trainer step: the in-place chain rule needs a gradient known to be fresh')
        if any(j[0].data_ptr() == g_rows.data_ptr() for j in pool.sink.inplace):
            return
        pool.sink.inplace.append((g_rows, _cl_rows(weight), u, v, sigma, int(rows), int(cin), int(taps)))

    @staticmethod
    def _run_inplace(jobs, cache):
        key = tuple(tuple(t.data_ptr() for t in j[:5]) + j[5:] for j in jobs)
        ent = cache.get(('inplace', key)) if cache is not None else None
        if ent is None:
            dev = jobs[0][0].device
            arr = (L.SnGradJob * len(jobs))()
            for i, (g, w, u, v, sg, rows, cin, taps) in enumerate(jobs):
                a = arr[i]
                a.g, a.w, a.u, a.v, a.sigma = g.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(), sg.data_ptr()
                a.rows, a.cin, a.taps = rows, cin, taps
            nb = L.lib().s2e_sngrad_block_map(C.byref(arr), len(jobs), None)
            bm = np.zeros(2 * nb, dtype=np.int32)
            L.lib().s2e_sngrad_block_map(C.byref(arr), len(jobs), bm.ctypes.data)       # (also fills part0 / nparts of the jobs)
            jobs_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
            nscratch = int(L.lib().s2e_sngrad_scratch_floats(C.byref(arr), len(jobs)))
            ent = (jobs_dev, torch.from_numpy(bm).to(dev), int(nb), torch.empty(nscratch, dtype=torch.float32, device=dev))
            if cache is not None:
                cache[('inplace', key)] = ent
        jobs_dev, map_dev, nb, partials = ent
        nbytes = float(sum(j[0].numel() * 16 for j in jobs))          # g and W read for the dot product, g read and written
        LaunchProfiler.run('weight_grad_relayout', 0.0, lambda: L.check(
            L.lib().s2e_sn_grads_inplace(jobs_dev.data_ptr(), map_dev.data_ptr(), nb, partials.data_ptr(), _stream()),
            's2e_sn_grads_inplace'), nbytes=nbytes)

    @staticmethod
    def push_wgrad(x, gy, dw_rows, dbias, rects=None, tag=None, gy_shared=False):
        """Queue the weight (and bias) gradient of a 3x3 stride-1 pad-1 conv -- x (N,H,W,Cin), gy (N,H,W,Cout) bf16 -- to be ACCUMULATED
        into dw_rows (Cout, 9*Cin) fp32 row-major / dbias (Cout) at the next flush: all queued layers as ONE persistent launch
        (s2e_wgrad_batch, csrc/conv_wgrad_batch.hip) instead of a launch + a 75-MB partial-tile round trip per layer.  rects =
        (rect_list, counts): the label-sparse form.  x, gy (and the list) stay referenced until the next flush.  False: not
        queued -- no trainer step open, a shape the batch does not take, a dW already queued in this flush (single-owner tiles are
        added without atomics), or S2E_WGRAD_BATCH=0 / S2E_DETERMINISTIC=1.
        gy_shared: the caller hands the SAME tensor on as somebody's gradient (a conv with a residual input returns it as the
        residual's gradient, and the block's first SPADE then adds its own dx into it IN PLACE -- ModulateFn's relay): the queue
        keeps a copy, the deferred launch must not see that sum."""
        pool = ZeroPool.active()
        if pool is None or _WGRAD_BATCH_OFF or x.dtype != torch.bfloat16 or gy.dtype != torch.bfloat16:
            return False
        n, h, w, cin = x.shape
        cout = gy.shape[-1]
        key = (n, h, w, cin, cout, rects is not None)
        ok = _WGRAD_BATCH_OK.get(key)
        if ok is None:
            ok = _WGRAD_BATCH_OK[key] = bool(L.lib().s2e_wgrad_batch_supported(L.S2E_BF16, n, h, w, cin, cout)) and \
                (rects is None or (h % 16 == 0 and w % 16 == 0))
        if not ok:
            return False
        for j in pool.sink.wg:
            if j[2].data_ptr() == dw_rows.data_ptr():
                j[6] = False                                 # (that dW receives another contribution before the flush: add, do not store)
                return False
        _need(x, gy, dw_rows, dbias)
        # a gradient arena that zero_grad cleared right before this step, and nothing queued for it yet: single-owner tiles are stored
        fresh = ZeroPool.grad_is_fresh(dw_rows) and dw_rows.data_ptr() not in pool.sink.wg_done
        pool.sink.wg.append([x, gy.clone() if gy_shared else gy, dw_rows, dbias, rects, tag, fresh])
        return True

    def _flush_wgrad(self):
        jobs, self.wg = self.wg, []
        self.wg_done.update(j[2].data_ptr() for j in jobs)
        dev = jobs[0][0].device
        arr = (L.WgradBatchJob * len(jobs))()
        flops = executed = nbytes = 0.0
        for a, (x, gy, dw, db, rects, _, fresh) in zip(arr, jobs):
            a.flags = 1 if fresh else 0                      # S2E_WGRAD_BATCH_DW_ZERO
            n, h, w, cin = x.shape
            cout = gy.shape[-1]
            a.x, a.gy, a.dw, a.dbias = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
            a.N, a.H, a.W, a.Cin, a.Cout = n, h, w, cin, cout
            f, frac = 2.0 * n * h * w * cin * cout * 9, 1.0
            if rects is not None:
                a.rect_list, a.rect_count = rects[0].data_ptr(), rects[1].data_ptr()
                if LaunchProfiler.active():
                    frac = float(int(rects[1][0])) / max(n * (h // 16) * (w // 16), 1)
            flops += f
            executed += f * frac
            nbytes += (x.numel() + gy.numel()) * 2.0 * frac + dw.numel() * 4.0
        ws = self.__dict__.get('_wg_ws')
        wsb = L.lib().s2e_wgrad_batch_workspace_bytes()
        if ws is None or ws.device != dev or ws.numel() * 4 < wsb:
            ws = self._wg_ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)     # (kept: the same 151 MB every flush)
        LaunchProfiler.run('conv_wgrad_patch', flops, lambda: L.check(
            L.lib().s2e_wgrad_batch(L.S2E_BF16, C.byref(arr), len(jobs), _p(ws), wsb, _stream()), 's2e_wgrad_batch'),
            tag='W k3 s1 x%d batched' % len(jobs), nbytes=nbytes, executed=executed)
        self.keep_wg = jobs                                  # the tensors stay referenced until the next flush (stream order covers the rest)

    @staticmethod
    def push_c8(oh, dactv, dw, db, ncls):
        """Queue the weight / bias gradient of a 3x3 conv on the 8-channel one-hot map `oh` (N,h,w,8) with output gradient
        `dactv` (N,h,w,128), accumulated straight into dw (128,ncls,3,3) / db (128) fp32 at the next flush -- all queued layers
        in one launch per slab shape (s2e_wgrad_c8_batch).  False: not queued (no scope, or a shape the batch does not take)."""
        pool = ZeroPool.active()
        if pool is None or _C8_BATCH_OFF or dw is None or db is None or oh.dtype != torch.bfloat16:
            return False
        n, h, w, _ = oh.shape
        if dactv.shape[-1] != 128 or not L.lib().s2e_wgrad_c8_batch_supported(L.S2E_BF16, h, w, 128):
            return False
        pool.sink.c8.append((oh, dactv, dw, db, int(ncls)))
        return True

    def _flush_c8(self):
        c8, self.c8 = self.c8, []
        n = c8[0][0].shape[0]
        rest = [j for j in c8 if j[0].shape[0] != n]         # (a launch shares one batch size: other sizes go in a round of their own)
        if rest:
            c8 = [j for j in c8 if j[0].shape[0] == n]
            self.c8 = rest
        arr = (L.WgradC8Job * len(c8))()
        for i, (oh, dactv, dw, db, ncls) in enumerate(c8):
            arr[i].x, arr[i].gy, arr[i].dw_oihw, arr[i].dbias = oh.data_ptr(), dactv.data_ptr(), dw.data_ptr(), db.data_ptr()
            arr[i].H, arr[i].W, arr[i].ncls = oh.shape[1], oh.shape[2], ncls
        wsb = L.lib().s2e_wgrad_c8_batch_workspace_bytes(n, C.byref(arr), len(c8))
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=c8[0][0].device)
        flops = sum(2.0 * n * j[0].shape[1] * j[0].shape[2] * 8 * 128 * 9 for j in c8)
        LaunchProfiler.run('conv_wgrad_patch', flops, lambda: L.check(
            L.lib().s2e_wgrad_c8_batch(L.S2E_BF16, n, C.byref(arr), len(c8), _p(ws), wsb, _stream()), 's2e_wgrad_c8_batch'),
            tag='W n%d c8->128 k3 s1 x%d batched' % (n, len(c8)),
            nbytes=float(sum((j[0].numel() + j[1].numel()) * 2 for j in c8)))
        self.keep_c8 = (c8, ws, self.keep_c8 if rest else None)   # alive until the next flush (stream order covers the rest)
        if rest:
            self._flush_c8()

    def _flush_uniform(self):
        """All queued SPADE layers' uniform-rectangle gradients (s2e_spade_uniform_grads): two launches per 16 layers."""
        jobs, self.uni = self.uni, []
        for i in range(0, len(jobs), 16):
            chunk = jobs[i:i + 16]
            arr = (L.SpadeUniJob * len(chunk))()
            for a, j in zip(arr, chunk):
                R, A, w_gb, w_sh, b_sh, dw_sh, db_sh, dw_gb, db_gb, c2, nh, ncls, act_bf16 = j
                a.R, a.A, a.w_gb = R.data_ptr(), A.data_ptr(), w_gb.data_ptr()
                a.w_sc, a.w_sk = w_gb.stride(0), w_gb.stride(1)
                a.w_st = w_gb.stride(3)                          # tap t = 3 ky + kx: stride(2) == 3 * stride(3) in both layouts
                a.w_sh, a.b_sh = w_sh.data_ptr(), b_sh.data_ptr()
                a.dw_sh = dw_sh.data_ptr() if dw_sh is not None else None
                a.db_sh = db_sh.data_ptr() if db_sh is not None else None
                a.dw_gb = dw_gb.data_ptr() if dw_gb is not None else None
                a.db_gb = db_gb.data_ptr() if db_gb is not None else None
                a.C2, a.nh, a.ncls, a.act_bf16 = c2, nh, ncls, act_bf16
            LaunchProfiler.run('spade_uniform_bwd', 0.0, lambda: L.check(
                L.lib().s2e_spade_uniform_grads(C.byref(arr), len(chunk), _stream()), 's2e_spade_uniform_grads'),
                nbytes=float(sum(j[2].numel() * 4 for j in chunk)))
        self.keep_uni = jobs                                 # the tensors stay referenced until the next flush

    def flush(self):
        if self.wg:
            self._flush_wgrad()                              # first: the re-layout / chain-rule / rank-1 jobs below read or add to its results
        if self.uni:
            self._flush_uniform()
        if self.c8:
            self._flush_c8()
        if self.inplace:
            jobs, self.inplace = self.inplace, []
            self.inplace_done.update(j[0].data_ptr() for j in jobs)
            GradSink._run_inplace(jobs, self.tables)
            self.keep_inplace = jobs                         # the tensors stay referenced until the next flush
        if not self.jobs:
            return
        jobs, self.jobs = self.jobs, []
        key = tuple((j[0].data_ptr(), j[1].data_ptr()) + tuple(0 if t is None else t.data_ptr() for t in j[2:6]) + j[6:] for j in jobs)
        dev = jobs[0][0].device
        ent = self.tables.get(key)
        if ent is None:
            arr = (L.GradJob * len(jobs))()
            nsn = 0
            for i, (dwp, dst, w, u, v, sg, cout, cin, taps, cin_pad) in enumerate(jobs):
                a = arr[i]
                a.gw_packed, a.out = dwp.data_ptr(), dst.data_ptr()
                a.cout, a.cin, a.taps, a.cin_pad = cout, cin, taps, cin_pad
                if w is not None:
                    a.w_orig, a.u, a.v, a.sigma, a.dot_index = w.data_ptr(), u.data_ptr(), v.data_ptr(), sg.data_ptr(), nsn
                    nsn += 1
                else:
                    a.dot_index = -1
            import numpy as np
            nb = L.lib().s2e_grad_block_map(C.byref(arr), len(jobs), None)
            bm = np.zeros(3 * nb, dtype=np.int32)
            L.lib().s2e_grad_block_map(C.byref(arr), len(jobs), bm.ctypes.data)
            jobs_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
            map_dev = torch.from_numpy(bm).to(dev)
            ent = (jobs_dev, map_dev, int(nb), max(j[8] for j in jobs), nsn)
            if len(self.tables) > 64:                        # (a step flushes once per all-reduce group: up to 2 x 7 tables + the in-place ones;
                self.tables.clear()                          #  a table rebuilt inside a hipGraph capture would be a host-to-device copy there)
            self.tables[key] = ent
        jobs_dev, map_dev, nb, max_taps, nsn = ent
        dots = ZeroPool.take(max(nsn, 1), torch.float32, dev)
        nbytes = float(sum(j[0].numel() * 4 * (3 if j[2] is not None else 2) + (j[0].numel() * 4 if j[2] is not None else 0) for j in jobs))
        LaunchProfiler.run('weight_grad_relayout', 0.0, lambda: L.check(
            L.lib().s2e_weight_grads_batched(jobs_dev.data_ptr(), map_dev.data_ptr(), nb, max_taps, int(nsn > 0),
                                             dots.data_ptr(), _stream()), 's2e_weight_grads_batched'), nbytes=nbytes)
        self.keepalive = jobs                                # the tensors of this flush stay referenced until the next one


# ------------------------------------------------------------------------------ raw launchers

def pack_weight(w_oihw, dtype, cin_pad=None, transposed=False, sigma=None):
    """OIHW fp32 -> MFMA B-operand matrix in the compute dtype; divided by the device scalar `sigma`
    (spectral norm) on the fly when given."""
    w = w_oihw.detach()
    cout, cin, kh, kw = w.shape
    cin_pad = cin if cin_pad is None else cin_pad
    # a weight stored channels-last (optim.FlatAdam) is packed from where it lies: rows in, rows out
    cl = w.dtype == torch.float32 and not w.is_contiguous() and _cl_dense(w) and cin_pad == cin and cin % 8 == 0
    if not cl and (w.dtype != torch.float32 or not w.is_contiguous()):
        w = w.float().contiguous()
    _need(_cl_rows(w) if cl else w, sigma)
    transposed = int(bool(transposed)) | (2 if cl else 0)
    dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
    lib = L.lib()
    rows = lib.s2e_conv_cout_pad(cin_pad if (transposed & 1) else cout)
    kpad = lib.s2e_conv_k_pad(dt, kh * kw * (cout if (transposed & 1) else cin_pad))
    out = torch.empty(rows, kpad, dtype=dtype, device=w.device)
    L.check(lib.s2e_pack_conv_weight(dt, _p(w), _p(out), _p(sigma), cout, cin, kh, kw, cin_pad, int(transposed), _stream()),
            's2e_pack_conv_weight')
    return out


def packed_weight(w, dtype, cin_pad, transposed, sigma, plan, generation=None, stable=True):
    """The packed matrix from the network's PackPlan (packing.py) when it has one for THIS forward, else an
    individual pack -- which also teaches the plan, so the next forward packs it in the batched launch.
    stable=False: `w` is a temporary (its address means nothing next time): never recorded."""
    if plan is not None and stable:
        wp = plan.lookup(w, dtype, cin_pad, transposed, generation)
        if wp is not None:
            return wp
        if generation is None or generation == plan.generation:
            plan.record(w, dtype, cin_pad, transposed, sigma)
    return pack_weight(w, dtype, cin_pad, transposed, sigma)


# profiler families = the kernel s2e_conv2d / s2e_conv2d_wgrad choose for the shape (S2E_KERNEL_GENERIC / SMALL / PATCH)
_CONV_FAMILY = ('conv_igemm', 'conv_small', 'conv_patch')
_WGRAD_FAMILY = ('conv_wgrad', 'conv_wgrad_small', 'conv_wgrad_patch')


_CONV_PLANS = {}


def _conv_plan(wgrad, dt, *shape):
    """(s2e_conv_desc, workspace bytes) of a launch, memoised per shape: a step repeats the same ~150 shapes, and building the
    ctypes structure + asking the library for the workspace size cost ~3 us of host time per launch."""
    key = (wgrad, dt) + shape
    ent = _CONV_PLANS.get(key)
    if ent is None:
        d = ConvDesc(*shape)
        wsb = (L.lib().s2e_conv2d_wgrad_workspace_bytes if wgrad else L.lib().s2e_conv2d_workspace_bytes)(dt, C.byref(d))
        if len(_CONV_PLANS) > 8192:
            _CONV_PLANS.clear()
        ent = _CONV_PLANS[key] = (d, wsb)
    return ent


# A/B switch: 0 = every patch-resident weight gradient as its own launch (round 4); S2E_DETERMINISTIC keeps the fixed-order per-layer path
_WGRAD_BATCH_OFF = os.environ.get('S2E_WGRAD_BATCH', '1') == '0' or os.environ.get('S2E_DETERMINISTIC', '0') == '1'
_WGRAD_BATCH_OK = {}
_CONV_STATS_OFF = os.environ.get('S2E_CONV_STATS', '1') == '0'      # A/B switch: 0 = InstanceNorm statistics always by a pass of their own
_CONV_STATS_SLOTS = {}


def _conv_stats_slots(dt, d, *shape):
    """s2e_conv2d_stats_slots, memoised per shape."""
    key = (dt,) + shape
    v = _CONV_STATS_SLOTS.get(key)
    if v is None:
        v = _CONV_STATS_SLOTS[key] = int(L.lib().s2e_conv2d_stats_slots(dt, C.byref(d)))
    return v


def conv2d_raw(x, wp, bias, residual, aux, out_hw_c, kh, kw, stride, pad, transposed=False,
               in_act=ACT_NONE, out_act=ACT_NONE, aux_mode=AUX_NONE, out=None, stats_out=None):
    """stats_out: None, or a list that receives the InstanceNorm statistics (N, Cout, 2) {mean, rstd} of the result when the
    kernel this shape takes produces their partial sums in its epilogue (s2e_conv2d_stats; then the caller needs no pass over y);
    left empty otherwise."""
    _need(x, wp, bias, residual, aux)
    n, hi, wi, cin = x.shape
    ho, wo, cout = out_hw_c
    y = torch.empty(n, ho, wo, cout, dtype=x.dtype, device=x.device) if out is None else out
    if out is not None and (tuple(out.shape) != (n, ho, wo, cout) or out.dtype != x.dtype or not out.is_contiguous()):
        raise ValueError('conv2d_raw: out must be a contiguous %s tensor with the shape of the result' % (x.dtype,))
    dt = _dt(x)
    d, wsb = _conv_plan(False, dt, n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, int(transposed), in_act, out_act, aux_mode)
    # algorithmic FLOPs = those of the forward conv this launch computes or differentiates (a stride-2
    # data-gradient executes 4x that on structural zeros; not counted)
    pix = hi * wi if transposed else ho * wo
    flops = 2.0 * n * pix * cin * cout * kh * kw
    if stats_out is not None and aux is None and not transposed and not _CONV_STATS_OFF:
        slots = _conv_stats_slots(dt, d, n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, in_act, out_act)
        if slots:
            part = torch.empty(n * slots * cout * 2, dtype=torch.float32, device=x.device)
            LaunchProfiler.run('conv_patch', flops, lambda: L.check(
                L.lib().s2e_conv2d_stats(dt, _p(x), _p(wp), _p(bias), _p(residual), _p(y), C.byref(d), _p(part), _stream()), 's2e_conv2d_stats'),
                tag=lambda: 'F n%d %dx%d c%d->%d k%d s%d +stats' % (n, hi, wi, cin, cout, kh, stride),
                nbytes=lambda: float((x.numel() + y.numel() + wp.numel() + (residual.numel() if residual is not None else 0)) * x.element_size()))
            ws = torch.empty(n * cout * 2, dtype=torch.float64, device=x.device)
            stats = torch.empty(n, cout, 2, dtype=torch.float32, device=x.device)
            LaunchProfiler.run('in_stats', 0.0, lambda: L.check(
                L.lib().s2e_in_stats_from_partials(_p(part), n, slots, cout, ho * wo, IN_EPS, _p(ws), _p(stats), _stream()),
                's2e_in_stats_from_partials'), nbytes=float(part.numel() * 4))
            stats_out.append(stats)
            return y
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
    LaunchProfiler.run(lambda: _CONV_FAMILY[L.lib().s2e_conv2d_kernel_kind(dt, C.byref(d))], flops, lambda: L.check(
        L.lib().s2e_conv2d(dt, _p(x), _p(wp), _p(bias), _p(residual), _p(aux), _p(y), C.byref(d), _p(ws), wsb,
                           _stream()), 's2e_conv2d'),
        tag=lambda: '%s n%d %dx%d c%d->%d k%d s%d' % ('D' if transposed else 'F', n, hi, wi, cin, cout, kh, stride),
        # algorithmic bytes: every operand once (x, packed w, y, + residual / mask tensor when present)
        nbytes=lambda: float((x.numel() + y.numel() + wp.numel() + (residual.numel() if residual is not None else 0)
                              + (aux.numel() if aux is not None else 0)) * x.element_size()))
    return y


def conv2d_wgrad_raw(x, gy, kh, kw, stride, pad, in_act=ACT_NONE, want_bias=False, dbias_out=None, dw_out=None, gy_shared=False):
    """-> (dw, db): dw (Cout, KH*KW*Cin) fp32 in packed order; db (Cout) fp32 or None.  Both live in one
    zero-filled buffer (ZeroPool scratch when the bias gradient is not returned).  dbias_out: an fp32 (Cout) tensor to ACCUMULATE the bias
    gradient into instead (e.g. the parameter's slice of the gradient arena); then db is None.
    dw_out: an fp32 (Cout, KH*KW*Cin) row-major tensor to ACCUMULATE the weight gradient into instead of a fresh zeroed buffer
    (the gradient of a parameter stored channels-last: _cl_rows(p.grad)); returned as dw.
    Inside a trainer step the patch-resident 3x3 shapes are QUEUED (GradSink.push_wgrad): dw / dbias_out then receive the sums at
    the step's next flush.  gy_shared: gy is also handed on as another tensor's gradient (see push_wgrad)."""
    _need(x, gy, dbias_out, dw_out)
    n, hi, wi, cin = x.shape
    _, ho, wo, cout = gy.shape
    k = kh * kw * cin
    own_b = want_bias and dbias_out is None
    if dw_out is not None:
        if tuple(dw_out.shape) != (cout, k) or dw_out.dtype != torch.float32 or not dw_out.is_contiguous():
            raise ValueError('conv2d_wgrad_raw: dw_out must be a contiguous fp32 (%d, %d) tensor' % (cout, k))
        dw, db = dw_out, (torch.zeros(cout, dtype=torch.float32, device=x.device) if own_b else None)
    elif own_b:                                          # db goes back to autograd (may become a .grad): never pooled
        buf = torch.zeros(cout * k + cout, dtype=torch.float32, device=x.device)
        dw, db = buf[:cout * k].view(cout, k), buf[cout * k:]
    else:
        dw, db = ZeroPool.take(cout * k, torch.float32, x.device).view(cout, k), None
    dbp = db if own_b else dbias_out
    if (kh == 3 and kw == 3 and stride == 1 and pad == 1 and in_act == ACT_NONE and not own_b and ho == hi and wo == wi
            and GradSink.push_wgrad(x, gy, dw, dbp, gy_shared=gy_shared)):
        return dw, db                                        # accumulated at the step's next flush, with every other queued layer
    d, wsb = _conv_plan(True, _dt(x), n, hi, wi, cin, ho, wo, cout, kh, kw, stride, pad, 0, in_act, ACT_NONE, AUX_NONE)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
    LaunchProfiler.run(lambda: _WGRAD_FAMILY[L.lib().s2e_conv2d_wgrad_kernel_kind(_dt(x), C.byref(d))],
                       2.0 * n * ho * wo * cin * cout * kh * kw, lambda: L.check(
        L.lib().s2e_conv2d_wgrad(_dt(x), _p(x), _p(gy), _p(dw), _p(dbp), C.byref(d), _p(ws), wsb, _stream()),
        's2e_conv2d_wgrad'),
        tag=lambda: 'W n%d %dx%d c%d->%d k%d s%d' % (n, hi, wi, cin, cout, kh, stride),
        nbytes=lambda: float((x.numel() + gy.numel()) * x.element_size() + dw.numel() * 4))
    return dw, db


def _cl_dense(t):
    """A 4-D tensor whose memory is one dense block in [d0][d2][d3][d1] order: a conv weight stored channels-last
    (optim.FlatAdam), i.e. already in the packed order of the MFMA kernels."""
    return t is not None and t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def _cl_rows(t):
    """(Cout, KH*KW*Cin) row-major view of a channels-last conv weight's (or gradient's) memory."""
    co, ci, kh, kw = t.shape
    return t.detach().permute(0, 2, 3, 1).reshape(co, kh * kw * ci)


def _grad_dst(p):
    """The tensor a backward kernel may accumulate this parameter's gradient into directly: its .grad when
    that already exists as a contiguous fp32 tensor (optim.FlatAdam keeps .grad as a view of the gradient
    arena and zeroes it at the start of every step).  None -> return the gradient to autograd instead."""
    if p is None or not p.is_leaf:                       # (a non-leaf's .grad is never an arena view; asking for it warns)
        return None
    g = getattr(p, 'grad', None)
    if g is None or g.dtype != torch.float32 or not (g.is_contiguous() or _cl_dense(g)) or not g.is_cuda:
        return None
    return g


def _adjacent(a, b):
    """b starts exactly where a ends in the same storage (both dense: contiguous, or channels-last conv weights)."""
    return (a is not None and b is not None and (a.is_contiguous() or _cl_dense(a)) and (b.is_contiguous() or _cl_dense(b))
            and a.is_contiguous() == b.is_contiguous()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel())


def _span2(a, shape):
    """View of `a`'s storage starting at a, with `shape` (covers a and the tensor laid out right after it, which is stacked
    along dimension 0); in a's memory order -- row-major, or channels-last for a conv weight stored that way."""
    if len(shape) == 4 and not a.is_contiguous() and _cl_dense(a):
        co, ci, kh, kw = shape
        return a.detach().as_strided(shape, (kh * kw * ci, 1, kw * ci, ci))
    strides, st = [], 1
    for d in reversed(shape):
        strides.append(st)
        st *= d
    return a.detach().as_strided(shape, tuple(reversed(strides)))


def unpack_weight_grad_into(dwp, dst, cout, cin, kh, kw, cin_pad, accumulate=True):
    if accumulate and GradSink.push(dwp, dst, cout, cin, kh * kw, cin_pad):
        return
    L.check(L.lib().s2e_unpack_weight_grad(_p(dwp), _p(dst), cout, cin, kh, kw, cin_pad, int(accumulate), _stream()),
            's2e_unpack_weight_grad')


def colsum(g):
    _need(g)
    c = g.shape[-1]
    out = torch.zeros(c, dtype=torch.float32, device=g.device)
    L.check(L.lib().s2e_colsum(_dt(g), _p(g), g.numel() // c, c, _p(out), _stream()), 's2e_colsum')
    return out


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
            if _PREPASS_OFF:
                return o
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


_C8_BATCH_OFF = os.environ.get('S2E_WGRAD_C8_BATCH', '1') == '0'      # A/B switch: every mlp_shared weight gradient as its own launches
_PREPASS_OFF = os.environ.get('S2E_SPADE_PREPASS', '1') == '0'      # A/B switch: every label conv / class table as its own launch


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


def _unpack_dw(dw, cout, cin, kh, kw, cin_pad):
    """(Cout, KH*KW*cin_pad) packed fp32 gradient -> (Cout, Cin, KH, KW) view."""
    return dw.view(cout, kh, kw, cin_pad)[..., :cin].permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------ conv2d

class LivePrefix:
    """Batches of which only the first `n` samples carry a gradient.  In the G step netD runs on [fake | real] and the
    real half only serves as the (detached) target of the feature-matching loss: its gradient is exactly zero all the way
    down.  A conv recorded inside `with LivePrefix.of(n)` computes its DATA gradient for the first n samples only and
    zero-fills the rest -- the same numbers for half the MFMA work.  netD gates its outputs (live_prefix_gate) so that the
    premise holds whatever the caller does with them."""
    n = None

    class of:
        def __init__(self, n):
            self.n = n

        def __enter__(self):
            self.prev, LivePrefix.n = LivePrefix.n, self.n

        def __exit__(self, *exc):
            LivePrefix.n = self.prev


def _live_tail_buffer(x, live):
    """A gradient buffer shaped like x whose samples live.. are ZERO, for a data gradient that only writes samples ..live.
    Inside a trainer step the i-th such request of a step gets the i-th PERSISTENT buffer of the step's pool (same sequence
    every step): its tail was zeroed when it was made and nothing writes there -- the data-gradient kernel fills the head, the
    consumers (FeatTapFn, the IN backward) read it or accumulate into the head only -- so the ten zero-fill launches of a G
    step's discriminator backward disappear.  Stand-alone: a fresh tensor and one fill."""
    pool = ZeroPool.active()
    if pool is not None:
        key = (pool.key, pool.tail_i)
        pool.tail_i += 1
        ent = pool.tails.get(key)
        if ent is not None and ent[0] == live and ent[1].shape == x.shape and ent[1].dtype == x.dtype and ent[1].device == x.device:
            return ent[1]
        if not pool.frozen:                                  # (a captured graph must not start using new persistent memory)
            t = torch.zeros_like(x)
            pool.tails[key] = (live, t)
            return t
    gx = torch.empty_like(x)
    gx[live:].zero_()
    return gx


class _LivePrefixGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, n):
        ctx.n = n
        return h.view_as(h)

    @staticmethod
    def backward(ctx, g):
        pool = ZeroPool.active()
        if pool is not None and any(live == ctx.n and t.data_ptr() == g.data_ptr() and t.shape == g.shape for live, t in pool.tails.values()):
            return g, None                                   # one of the step's zero-tailed buffers (_live_tail_buffer): nothing to do
        g = g.clone()
        g[ctx.n:].zero_()
        return g, None


def live_prefix_gate(h, n):
    """Identity whose backward zeroes the gradient of samples n.. (see LivePrefix)."""
    return _LivePrefixGate.apply(h, n)


class Conv2dFn(torch.autograd.Function):
    """y = out_act(conv(in_act(x), W) + b + residual) on NHWC tensors.  x may carry more channels
    than W has input channels (structural zero padding).  With (u, v, sigma) given, W = weight/sigma
    (spectral norm): the division happens inside the pack kernel and the gradient returned for
    `weight` is the one w.r.t. weight_orig, through sigma."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, pad, in_act, out_act, u, v, sigma, stats_out=None):
        n, hi, wi, cx = x.shape
        cout, cin, kh, kw = weight.shape
        if cx < cin:
            raise ValueError('input has %d channels, weight expects %d' % (cx, cin))
        ho = (hi + 2 * pad - kh) // stride + 1
        wo = (wi + 2 * pad - kw) // stride + 1
        plan = packing.current()
        wp = packed_weight(weight, x.dtype, cx, False, sigma, plan)
        ctx.plan, ctx.plan_gen = plan, (plan.generation if plan is not None else None)
        b = None if bias is None else bias.detach().float().contiguous()
        y = conv2d_raw(x, wp, b, residual, None, (ho, wo, cout), kh, kw, stride, pad, False, in_act, out_act, stats_out=stats_out)
        ctx.cfg = (stride, pad, in_act, out_act, bias is not None, residual is not None)
        ctx.live = LivePrefix.n
        ctx.wdst = _grad_dst(weight)                       # direct accumulation targets (or None)
        ctx.bdst = _grad_dst(bias) if bias is not None else None
        ctx.save_for_backward(x, weight, y if out_act != ACT_NONE else None, u, v, sigma)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y, u, v, sigma = ctx.saved_tensors
        stride, pad, in_act, out_act, has_bias, has_res = ctx.cfg
        n, hi, wi, cx = x.shape
        cout, cin, kh, kw = weight.shape
        g = gy.contiguous()
        live = ctx.live
        if (live is not None and 0 < live < n and ctx.needs_input_grad[0] and not ctx.needs_input_grad[1]
                and not (has_bias and ctx.needs_input_grad[2]) and not (has_res and ctx.needs_input_grad[3])
                and out_act in (ACT_NONE, ACT_LRELU)):
            # only the first `live` samples carry a gradient (LivePrefix): data gradient of that prefix, zeros behind it
            gl = g[:live]
            if out_act == ACT_LRELU:
                g2 = torch.empty_like(gl)
                L.check(L.lib().s2e_lrelu_bwd(_dt(gl), _p(gl), _p(y), _p(g2), gl.numel(), _stream()), 's2e_lrelu_bwd')
                gl = g2
            gx = _live_tail_buffer(x, live)                    # (samples live.. are zero already)
            wpt = packed_weight(weight, x.dtype, cx, True, sigma, ctx.plan, ctx.plan_gen)
            conv2d_raw(gl, wpt, None, None, x[:live] if in_act == ACT_LRELU else None, (hi, wi, cx), kh, kw, stride, pad,
                       True, ACT_NONE, ACT_NONE, AUX_LRELU_GRAD if in_act == ACT_LRELU else AUX_NONE, out=gx[:live])
            return gx, None, None, None, None, None, None, None, None, None, None, None
        if out_act == ACT_TANH:
            g2 = torch.empty_like(g)
            L.check(L.lib().s2e_tanh_bwd(_dt(g), _p(g), _p(y), _p(g2), g.numel(), _stream()), 's2e_tanh_bwd')
            g = g2
        elif out_act == ACT_LRELU:
            g2 = torch.empty_like(g)
            L.check(L.lib().s2e_lrelu_bwd(_dt(g), _p(g), _p(y), _p(g2), g.numel(), _stream()), 's2e_lrelu_bwd')
            g = g2
        gx = gw = gb = gres = None
        if ctx.needs_input_grad[0]:
            wpt = packed_weight(weight, x.dtype, cx, True, sigma, ctx.plan, ctx.plan_gen)
            gx = conv2d_raw(g, wpt, None, None, x if in_act == ACT_LRELU else None, (hi, wi, cx), kh, kw, stride, pad,
                            True, ACT_NONE, ACT_NONE, AUX_LRELU_GRAD if in_act == ACT_LRELU else AUX_NONE)
        want_b = has_bias and ctx.needs_input_grad[2]
        wdst = ctx.wdst
        shared = bool(has_res and ctx.needs_input_grad[3])   # g goes on as the residual's gradient (and may be added to in place there)
        direct = ctx.needs_input_grad[1] and wdst is not None and cx == cin and cin % 8 == 0 and _cl_dense(wdst)
        if direct and sigma is not None and not GradSink.inplace_allowed(wdst):
            direct = False                                   # (the chain rule must ACCUMULATE here: packed scratch, below)
        if ctx.needs_input_grad[1] and sigma is not None and not direct and ctx.wdst is not None:
            ZeroPool.arena_touched(ctx.wdst)                 # (.grad now holds a chain-ruled part: no in-place rewrite before zero_grad)
        if direct:
            # the parameter's gradient lies in the packed order (channels-last arena, or any 1x1 conv; Cin % 8 == 0 -- a 1-channel
            # weight is "channels-last" too, but the in-place kernels work on 16-byte groups of one tap): the kernel accumulates
            # straight into it; spectral norm's chain rule is then applied in place (queued: one launch pair per step)
            _, gb = conv2d_wgrad_raw(x, g, kh, kw, stride, pad, in_act, want_b, ctx.bdst if want_b else None, dw_out=_cl_rows(wdst),
                                     gy_shared=shared)
            if sigma is not None:
                GradSink.push_inplace(_cl_rows(wdst), weight, u, v, sigma, cout, cin, kh * kw)
        elif ctx.needs_input_grad[1]:
            bdst = ctx.bdst if want_b else None
            dwp, gb = conv2d_wgrad_raw(x, g, kh, kw, stride, pad, in_act, want_b, bdst, gy_shared=shared)
            if wdst is not None and not wdst.is_contiguous():
                wdst = None                                  # (a channels-last .grad fed a channel-padded input: through autograd)
            w_oihw = weight.detach() if weight.is_contiguous() else weight.detach().contiguous()
            if sigma is None:
                if wdst is not None:
                    unpack_weight_grad_into(dwp, wdst, cout, cin, kh, kw, cx)
                else:
                    gw = _unpack_dw(dwp, cout, cin, kh, kw, cx)
            else:
                acc = wdst is not None
                if not (acc and GradSink.push(dwp, wdst, cout, cin, kh * kw, cx, w_oihw, u, v, sigma)):
                    out = wdst if acc else torch.empty(cout, cin, kh, kw, dtype=torch.float32, device=x.device)
                    dot = ZeroPool.take(1, torch.float32, x.device)
                    L.check(L.lib().s2e_sn_weight_grad(_p(dwp), _p(w_oihw), _p(u), _p(v), _p(sigma), _p(dot), _p(out),
                                                       cout, cin, kh, kw, cx, int(acc), _stream()), 's2e_sn_weight_grad')
                    gw = None if acc else out
        elif want_b:
            gb = colsum(g)
        if has_res and ctx.needs_input_grad[3]:
            gres = g
        return gx, gw, gb, gres, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, residual=None, stride=1, pad=0, in_act=ACT_NONE, out_act=ACT_NONE, sn=None, stats_out=None):
    """stats_out: see conv2d_raw -- a list that receives in_stats(result) when the conv kernel can produce it."""
    u, v, sigma = sn if sn is not None else (None, None, None)
    return Conv2dFn.apply(x, weight, bias, residual, stride, pad, in_act, out_act, u, v, sigma, stats_out)


def conv2d_m(x, conv, residual=None, stride=1, pad=0, in_act=ACT_NONE, out_act=ACT_NONE, stats_out=None):
    """conv2d on an nn.Conv2d parameter container (spectral-normed or not)."""
    from .spectral import conv_params
    weight, bias, sn = conv_params(conv)
    return conv2d(x, weight, bias, residual, stride, pad, in_act, out_act, sn, stats_out)


# ------------------------------------------------------------------------------ the encoder's head
class FcHeadFn(torch.autograd.Function):
    """y = fc(LeakyReLU(x).view(M, -1)) for an NHWC feature map x (M,h,w,C) and an nn.Linear whose input features are torch's
    (c, y, x) flattening (reference models/networks/encoder.py:68-71): s2e_fc_head_fwd / _bwd.  The weight / bias gradients are
    accumulated into the parameters' .grad when that is an fp32 arena view (None then goes back to autograd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope):
        _need(x, weight, bias)
        m, h, w, c = x.shape
        n = weight.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        L.check(L.lib().s2e_fc_head_fwd(_dt(x), _p(x), _p(weight), _p(bias), _p(y), m, h * w, c, n, float(slope), _stream()), 's2e_fc_head_fwd')
        ctx.slope = float(slope)
        ctx.wdst, ctx.bdst = _grad_dst(weight), _grad_dst(bias)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        m, h, w, c = x.shape
        n = weight.shape[0]
        g = gy.float().contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        wdst = ctx.wdst if (ctx.wdst is not None and ctx.wdst.is_contiguous()) else None
        dw = wdst if wdst is not None else (torch.zeros_like(weight) if ctx.needs_input_grad[1] else None)
        db = ctx.bdst if ctx.bdst is not None else (torch.zeros(n, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None)
        L.check(L.lib().s2e_fc_head_bwd(_dt(x), _p(x), _p(weight), _p(g), _p(dx), _p(dw), _p(db), m, h * w, c, n, ctx.slope, _stream()),
                's2e_fc_head_bwd')
        return dx, (None if wdst is not None else dw), (None if ctx.bdst is not None else db), None


_FC_HEAD_OFF = os.environ.get('S2E_FC_HEAD', '1') == '0'      # A/B switch: the encoder's head as a 4x4 valid convolution


def fc_head(x, weight, bias, slope=0.2):
    """-> (M, N) fp32, or None when the shape is outside the kernel's range (the caller then takes the convolution form)."""
    if (_FC_HEAD_OFF or weight.dtype != torch.float32 or not weight.is_contiguous() or bias is None or weight.shape[1] != x.shape[1] * x.shape[2] * x.shape[3]
            or weight.shape[1] * 4 > 48 * 1024 or not L.lib().s2e_fc_head_supported(x.shape[0], weight.shape[0])):
        return None
    return FcHeadFn.apply(x.contiguous(), weight, bias, slope)


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
        dwp, gb = conv2d_wgrad_raw(oh, g, 3, 3, 1, 1, ACT_NONE, True, ctx.bdst)
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


_SPARSE_BWD_OFF = os.environ.get('S2E_SPADE_SPARSE_BWD', '1') == '0' or os.environ.get('S2E_DETERMINISTIC', '0') == '1'
_byref = C.byref          # (functions below use C for a channel count)


# smallest map side the label-sparse backward takes: at 64 x 64 only the 2 x 2 inner rectangles of 16 can be uniform-interior at all (none
# is, on the bench's maps) and the lists + sums are pure overhead; 96 / 192 measured 18.43-18.48 / 18.40-18.44 ms against 18.44-18.60 at 48
_SPARSE_BWD_MIN = int(os.environ.get('S2E_SPARSE_BWD_MIN', '96'))


def _sparse_bwd_lists(ctx, g, h, w, cch, nh, ncls):
    """(cls, work_list, ui_list, counts) for the label-sparse backward of this SPADE layer, or None: the forward ran label-sparse
    on 16 x 16 rectangles (ctx.rects), a trainer step is open (zeroed scratch, deferred flush), mlp_shared's gradients go straight
    to the arena, and the data-gradient's kernel takes a rectangle list.  S2E_SPADE_SPARSE_BWD=0 / S2E_DETERMINISTIC=1 (the sums
    use float atomics): off."""
    rects = getattr(ctx, 'rects', None)
    pool = ZeroPool.active()
    if _SPARSE_BWD_OFF or rects is None or pool is None or g.dtype != torch.bfloat16 or ncls > 4 or nh > 128:
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
        conv2d_wgrad_raw(actv, g, 3, 3, 1, 1, ACT_NONE, True, ctx.gb_dst[1], dw_out=_cl_rows(ctx.gb_dst[0]))
    elif ctx.gb_dst is not None:
        dwp, _ = conv2d_wgrad_raw(actv, g, 3, 3, 1, 1, ACT_NONE, True, ctx.gb_dst[1])
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
        dactv = ZeroPool.take(n * h * w * nh, g.dtype, g.device).view(n, h, w, nh)       # (zero where no conv runs)
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
        dactv = conv2d_raw(g, wpt, None, None, actv, (h, w, nh), 3, 3, 1, 1, True, ACT_NONE, ACT_NONE, AUX_RELU_MASK)
    # (a streaming class-bucket kernel for this gradient was tried twice -- LDS float atomics, then per-wave
    # queues of boundary pixels -- and lost to the MFMA wgrad against the 8-channel one-hot map: 1.7 vs 0.75 ms
    # per step; see DESIGN.md "tried and dropped")
    oh = onehot_nhwc_raw(label, None, h, w, ncls, 8, g.dtype)
    wdst, bdst = ctx.sh_dst
    if GradSink.push_c8(oh, dactv, wdst, bdst, ncls):        # inside a trainer step: all mlp_shared gradients in one launch, later
        return gw_sh, gb_sh, gw_g, gb_g, gw_b, gb_b
    dwp, gb_sh = conv2d_wgrad_raw(oh, dactv, 3, 3, 1, 1, ACT_NONE, True, bdst)
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
    dx = g_relay if acc else torch.empty_like(x)
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
    from . import distributed as sdist
    world = sdist.sync_world_size() if ctx.batch else 1

    def launch(stage, count):
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


_SPARSE_OFF = os.environ.get('S2E_SPADE_SPARSE', '1') == '0'      # A/B switch: the dense fused launch everywhere
_SPARSE_MIN_RECTS = int(os.environ.get('S2E_SPADE_SPARSE_RECTS', '256'))


def label_rects(label, h, w, dtype, c, nh, flags=0):
    """Classification of the fused launch's rectangles of the (h, w)-downsampled label map into label-uniform and dense ones
    (s2e_label_rect_classify) -> (cls, dense_list, uni_list, counts, tw, th), or None when the label-sparse form is off
    / not worth it for this size.  Inside a trainer step the result is shared by every SPADE of the resolution (and by both
    forwards' layers: it depends on the label batch only)."""
    if _SPARSE_OFF:
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
    if _FUSED_OFF:
        return False
    return bool(L.lib().s2e_spade_conv_modulate_supported(_dt(x), n, h, w, c, nh, int(flags)))


_FUSED_OFF = os.environ.get('S2E_SPADE_FUSED', '1') == '0'      # A/B switch: the two-launch path everywhere


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


# ------------------------------------------------------------------------------ resampling

class Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need(x)
        n, h, w, c = x.shape
        y = torch.empty(n, 2 * h, 2 * w, c, dtype=x.dtype, device=x.device)
        LaunchProfiler.run('resample', 0.0, lambda: L.check(
            L.lib().s2e_upsample2x_fwd(_dt(x), _p(x), _p(y), n, h, w, c, _stream()), 's2e_upsample2x_fwd'),
            nbytes=float(5 * x.numel() * x.element_size()))
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        n, h2, w2, c = gy.shape
        gx = torch.empty(n, h2 // 2, w2 // 2, c, dtype=gy.dtype, device=gy.device)
        L.check(L.lib().s2e_upsample2x_bwd(_dt(gy), _p(gy), _p(gx), n, h2 // 2, w2 // 2, c, _stream()), 's2e_upsample2x_bwd')
        return gx


def upsample2x(x):
    return Upsample2xFn.apply(x)


class BilinearResizeFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear') of single-channel images (N,1,H,W) fp32 -> (N,h,w,1) NHWC in `dtype`
    (encoder.py:54-55)."""

    @staticmethod
    def forward(ctx, x, h, w, dtype):
        a = _single_channel(x.detach().float())
        _need(a)
        n, H, W = a.shape
        y = torch.empty(n, h, w, 1, dtype=dtype, device=a.device)
        L.check(L.lib().s2e_bilinear_resize_fwd(_dt(y), _p(a), _p(y), n, H, W, h, w, _stream()), 's2e_bilinear_resize_fwd')
        ctx.shape = (tuple(x.shape), H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, H, W = ctx.shape
        gy = gy.contiguous()
        n, h, w, _ = gy.shape
        gx = torch.zeros(n, H, W, dtype=torch.float32, device=gy.device)
        L.check(L.lib().s2e_bilinear_resize_bwd(_dt(gy), _p(gy), _p(gx), n, H, W, h, w, _stream()), 's2e_bilinear_resize_bwd')
        return gx.view(shape), None, None, None


def bilinear_resize(x, h, w, dtype):
    return BilinearResizeFn.apply(x, h, w, dtype)


class AvgPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need(x)
        n, h, w, c = x.shape
        y = torch.empty(n, (h + 1) // 2, (w + 1) // 2, c, dtype=x.dtype, device=x.device)
        L.check(L.lib().s2e_avgpool3x3s2_fwd(_dt(x), _p(x), _p(y), n, h, w, c, _stream()), 's2e_avgpool3x3s2_fwd')
        ctx.hw = (h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        h, w = ctx.hw
        n, _, _, c = gy.shape
        gx = torch.empty(n, h, w, c, dtype=gy.dtype, device=gy.device)
        L.check(L.lib().s2e_avgpool3x3s2_bwd(_dt(gy), _p(gy), _p(gx), n, h, w, c, _stream()), 's2e_avgpool3x3s2_bwd')
        return gx


def avgpool3x3s2(x):
    return AvgPool3x3s2Fn.apply(x)


class SegImageConcatFn(torch.autograd.Function):
    """cat([one_hot(label), image], channel) as an NHWC tensor zero-padded to `cpad` channels
    (pix2pix_model.py:328-336).  Differentiable w.r.t. the image only."""

    @staticmethod
    def forward(ctx, label, img, ncls, cpad):
        n, H, W = label.shape
        ctx.ncls = ncls
        return onehot_nhwc_raw(label, img.contiguous(), H, W, ncls, cpad, img.dtype)

    @staticmethod
    def backward(ctx, g):
        return None, g[..., ctx.ncls].contiguous(), None, None


def seg_image_concat(label, img, ncls=4, cpad=8):
    """label (N,H,W) uint8, img (N,H,W) -> (N,H,W,cpad)."""
    return SegImageConcatFn.apply(label, img, ncls, cpad)


class DInputFn(torch.autograd.Function):
    """The discriminator's input  cat_batch([cat_ch(one_hot(label), fake); cat_ch(one_hot(label), real)])  (pix2pix_model.py:
    328-342) as ONE (2N,H,W,cpad) NHWC tensor written by two launches -- one per half, straight from the label map and the two
    single-channel image batches: no concatenated image batch, no doubled label map.  Differentiable w.r.t. `fake` only; its
    gradient is channel `ncls` of the first half (one strided copy instead of select_backward's zero-fill + copy)."""

    @staticmethod
    def forward(ctx, label, fake, real, ncls, cpad):
        n, H, W = label.shape
        f, r = fake.reshape(n, H, W), real.reshape(n, H, W).to(fake.dtype)
        f, r = (f if f.is_contiguous() else f.contiguous()), (r if r.is_contiguous() else r.contiguous())
        _need(label, f, r)
        out = torch.empty(2 * n, H, W, cpad, dtype=fake.dtype, device=label.device)
        for half, img in ((out[:n], f), (out[n:], r)):
            L.check(L.lib().s2e_onehot_nhwc(_dt(out), _p(label), _p(img), _p(half), n, H, W, H, W, ncls, cpad, _stream()),
                    's2e_onehot_nhwc')
        ctx.ncls, ctx.shape = ncls, fake.shape
        return out

    @staticmethod
    def backward(ctx, g):
        n = g.shape[0] // 2
        return None, g[:n, :, :, ctx.ncls].contiguous().view(ctx.shape), None, None, None


def d_input(label, fake, real, ncls=4, cpad=8):
    """label (N,H,W) uint8, fake / real (N,1,H,W) or (N,H,W) -> (2N,H,W,cpad): see DInputFn."""
    return DInputFn.apply(label, fake, real, ncls, cpad)


class SplitHalvesFn(torch.autograd.Function):
    """t -> (t[:n], t[n:]), n = half the batch (Pix2PixModel.divide_pred).  Two plain slices cost a zero-filled full tensor and
    a copy EACH on the way back, plus the add that joins them; here the backward fills one tensor with the two halves (or
    zeros where a half got no gradient)."""

    @staticmethod
    def forward(ctx, t):
        ctx.set_materialize_grads(False)
        n = t.shape[0] // 2
        ctx.n = n
        ctx.like = t.detach()
        return t[:n], t[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None
        out = torch.empty_like(ctx.like)
        for dst, g in ((out[:ctx.n], ga), (out[ctx.n:], gb)):
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g)
        return out


def split_halves(t):
    return SplitHalvesFn.apply(t)


# ------------------------------------------------------------------------------ losses

def _loss_slot(device, pooled):
    """A zeroed fp32 scalar for s2e_loss_reduce to accumulate into.  Pool memory is recycled by the trainer's next step: only
    for terms that are consumed inside the step (see loss_sum)."""
    if pooled and ZeroPool.active() is not None:
        return ZeroPool.take(1, torch.float32, device).view(())
    return torch.zeros((), dtype=torch.float32, device=device)


class LossSumFn(torch.autograd.Function):
    """scale * sum_i f(a_i, b_i) as a 0-dim fp32 tensor (see s2e_loss_reduce for f)."""

    @staticmethod
    def forward(ctx, a, b, mode, scale, pooled=False):
        _need(a, b)
        out = _loss_slot(a.device, pooled)
        L.check(L.lib().s2e_loss_reduce(_dt(a), mode, _p(a), _p(b), a.numel(), float(scale), _p(out), _stream()),
                's2e_loss_reduce')
        ctx.cfg = (mode, float(scale))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        mode, scale = ctx.cfg
        gs = gout.detach().float().contiguous()
        da = torch.empty_like(a)
        L.check(L.lib().s2e_loss_grad(_dt(a), mode, _p(a), _p(b), a.numel(), scale, _p(gs), _p(da), 0, _stream()),
                's2e_loss_grad')
        return da, None, None, None, None


class HalfLossFn(torch.autograd.Function):
    """scale * sum_i f(t_i) over ONE half of a [fake | real] batch t (2N, ...), NHWC-contiguous; the other half gets no gradient.
    == loss_sum(t[:N] or t[N:], ...) without the slice: the backward writes the element-wise gradient into the head of a
    zero-tailed buffer (first half, inside a trainer step: _live_tail_buffer -- one launch, and the LivePrefix gate behind it
    recognises the buffer) instead of slice_backward's zero-fill + copy."""

    @staticmethod
    def forward(ctx, t, second, mode, scale, pooled):
        _need(t)
        n = t.shape[0] // 2
        half = t[n:] if second else t[:n]
        out = _loss_slot(t.device, pooled)
        L.check(L.lib().s2e_loss_reduce(_dt(t), mode, _p(half), None, half.numel(), float(scale), _p(out), _stream()), 's2e_loss_reduce')
        ctx.cfg = (bool(second), mode, float(scale), n)
        ctx.save_for_backward(t)
        return out

    @staticmethod
    def backward(ctx, gout):
        t, = ctx.saved_tensors
        second, mode, scale, n = ctx.cfg
        gs = gout.detach().float().contiguous()
        if second:
            gt = torch.empty_like(t)
            gt[:n].zero_()
            dst, src = gt[n:], t[n:]
        else:
            gt = _live_tail_buffer(t, n)
            dst, src = gt[:n], t[:n]
        L.check(L.lib().s2e_loss_grad(_dt(t), mode, _p(src), None, src.numel(), scale, _p(gs), _p(dst), 0, _stream()), 's2e_loss_grad')
        return gt, None, None, None, None


class PairLossFn(torch.autograd.Function):
    """(scale * sum f_a(t[:N]), scale * sum f_b(t[N:])) for a [fake | real] batch t: the discriminator's two hinge terms from
    the undivided prediction; the backward fills ONE gradient tensor with two launches (no slice_backward, no add)."""

    @staticmethod
    def forward(ctx, t, mode_a, mode_b, scale, pooled):
        _need(t)
        n = t.shape[0] // 2
        outs = []
        for half, mode in ((t[:n], mode_a), (t[n:], mode_b)):
            out = _loss_slot(t.device, pooled)
            L.check(L.lib().s2e_loss_reduce(_dt(t), mode, _p(half), None, half.numel(), float(scale), _p(out), _stream()), 's2e_loss_reduce')
            outs.append(out)
        ctx.cfg = (mode_a, mode_b, float(scale), n)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(t)
        return tuple(outs)

    @staticmethod
    def backward(ctx, ga, gb):
        t, = ctx.saved_tensors
        mode_a, mode_b, scale, n = ctx.cfg
        if ga is None and gb is None:
            return None, None, None, None, None
        gt = torch.empty_like(t)
        for dst, src, mode, g in ((gt[:n], t[:n], mode_a, ga), (gt[n:], t[n:], mode_b, gb)):
            if g is None:
                dst.zero_()
                continue
            gs = g.detach().float().contiguous()
            L.check(L.lib().s2e_loss_grad(_dt(t), mode, _p(src), None, src.numel(), scale, _p(gs), _p(dst), 0, _stream()), 's2e_loss_grad')
        return gt, None, None, None, None


def half_loss(t, second, mode, scale, pooled=False):
    return HalfLossFn.apply(t, second, mode, scale, pooled)


def pair_loss(t, mode_a, mode_b, scale, pooled=False):
    return PairLossFn.apply(t, mode_a, mode_b, scale, pooled)


def loss_sum(a, b, mode, scale, pooled=False):
    """pooled: the caller only COMBINES the result with other terms inside the step (a sum over scales, a stack) and never
    hands it out: the accumulator may then be a slice of the step's zero pool instead of its own zero-fill launch."""
    return LossSumFn.apply(a, b, mode, scale, pooled)


class FeatTapFn(torch.autograd.Function):
    """Identity on a discriminator feature map h = [fake | real] (2N,H,W,C) that also yields the GAN feature-
    matching term  scale * sum |h[:N] - h[N:].detach()|  (pix2pix_model.py:231-241 of the reference).

    Why not slice-then-loss: the slice's backward materialises a zero (2N,...) tensor, copies the half in and
    autograd then ADDS it to the gradient arriving from the next layer -- three passes over every feature map
    (~0.65 ms per G step).  Here the next layer's gradient arrives first (this node sits on the only path to
    it) and the L1 gradient is accumulated into its fake half in place by s2e_loss_grad(accumulate=1)."""

    @staticmethod
    def forward(ctx, h, scale, pooled=False):
        _need(h)
        n = h.shape[0] // 2
        a, b = h[:n], h[n:]
        out = _loss_slot(h.device, pooled)
        L.check(L.lib().s2e_loss_reduce(_dt(h), LOSS_L1, _p(a), _p(b), a.numel(), float(scale), _p(out), _stream()),
                's2e_loss_reduce')
        ctx.scale = float(scale)
        ctx.save_for_backward(h)
        ctx.set_materialize_grads(False)
        return h.view_as(h), out

    @staticmethod
    def backward(ctx, gh, gloss):
        h, = ctx.saved_tensors
        n = h.shape[0] // 2
        if gloss is None:
            return gh, None, None
        if gh is None:
            gh = torch.zeros_like(h)
        elif not gh.is_contiguous():
            gh = gh.contiguous()
        a, b, ga = h[:n], h[n:], gh[:n]
        gs = gloss.detach().float().contiguous()
        L.check(L.lib().s2e_loss_grad(_dt(h), LOSS_L1, _p(a), _p(b), a.numel(), ctx.scale, _p(gs), _p(ga), 1, _stream()),
                's2e_loss_grad')
        return gh, None, None


def feat_tap(h, scale, pooled=False):
    """-> (h, term): see FeatTapFn.  pooled: as in loss_sum."""
    return FeatTapFn.apply(h, scale, pooled)


# ------------------------------------------------------------------------------ optimizer

def adam_flat_step(p, g, m, v, hyper):
    """One torch.optim.Adam step over flat fp32 arenas (pix2pix_model.py:92-110 semantics).
    hyper: 6-float DEVICE tensor {lr, beta1, beta2, eps, completed steps, grad_scale}."""
    _need(p, g, m, v, hyper)
    LaunchProfiler.run('adam', 0.0, lambda: L.check(
        L.lib().s2e_adam_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _stream()), 's2e_adam_flat'),
        nbytes=float(7 * 4 * p.numel()))                              # read p, g, m, v + write p, m, v


# ------------------------------------------------------------------------------ OpenEDS validation metric
def _single_channel(x):
    """(N,1,H,W) / (N,H,W,1) / (N,H,W) -> contiguous (N,H,W) view of the same dtype."""
    if x.dim() == 4 and x.shape[1] == 1:
        x = x[:, 0]
    elif x.dim() == 4 and x.shape[-1] == 1:
        x = x[..., 0]
    if x.dim() != 3:
        raise ValueError('single-channel image batch expected, got shape %s' % (tuple(x.shape),))
    return x.contiguous()


def openeds_error(produced, target):
    """Per-image OpenEDS error of two batches in [-1, 1] (models/networks/loss.py:135-155 `calculate_mse_for_tensors`):
    both mapped to 0..255 with the reference's int truncation, then sqrt(sum d^2) / (H*W).  -> fp32 (N,), no gradient."""
    a, b = _single_channel(produced.detach()), _single_channel(target.detach().to(produced.dtype))
    _need(a, b)
    n, h, w = a.shape
    err = torch.empty(n, dtype=torch.float32, device=a.device)
    L.check(L.lib().s2e_openeds_error(_dt(a), _p(a), _p(b), n, h, w, _p(err), _stream()), 's2e_openeds_error')
    return err


def openeds_error_u8(produced, target):
    """The same on uint8 images that already are 0..255 (loss.py:116-133 `calculate_mse_for_images`)."""
    a, b = _single_channel(produced), _single_channel(target)
    if a.dtype != torch.uint8 or b.dtype != torch.uint8:
        raise ValueError('uint8 images expected')
    _need(a, b)
    n, h, w = a.shape
    err = torch.empty(n, dtype=torch.float32, device=a.device)
    L.check(L.lib().s2e_openeds_error_u8(_p(a), _p(b), n, h, w, _p(err), _stream()), 's2e_openeds_error_u8')
    return err


def resize_to255(x, w=400, h=640):
    """Bilinear resize (cv2.INTER_LINEAR rule) of single-channel [-1, 1] images to (h, w), then 0..255 with int truncation
    (data/postprocessor.py:92-107 `to_255resized_imagebatch`).  -> uint8 (N,1,h,w)."""
    a = _single_channel(x.detach())
    _need(a)
    n, hi, wi = a.shape
    out = torch.empty(n, 1, h, w, dtype=torch.uint8, device=a.device)
    L.check(L.lib().s2e_resize_to255(_dt(a), _p(a), n, hi, wi, _p(out), h, w, _stream()), 's2e_resize_to255')
    return out
