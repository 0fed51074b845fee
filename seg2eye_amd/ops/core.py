"""What every op family shares: dtype / pointer / stream helpers, the per-launch profiler, the trainer step's zero-filled scratch
(ZeroPool) with its queue of deferred weight-side launches (GradSink), and the views of channels-last parameter memory."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from . import switches

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
                self.pool.sink.gwg = []
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
        self.gwg = []              # deferred GENERIC weight gradients (one multi-job launch per flush): push_gwg
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
        residual's gradient, and the block's first SPADE then adds its own dx to it -- ModulateFn's relay): the deferred launch must not
        see that sum.  Rounds 4-5 queued a COPY (135 MB of copies per G step); since round 6 the relay asks is_pinned() and writes its
        sum to a new tensor instead of in place when the one it was handed is still to be read by a queued job."""
        pool = ZeroPool.active()
        if pool is None:
            return False
        # FIRST, whatever the shape: a dW that already has a queued job receives another contribution before the flush -- that job must
        # ADD, not store (the scan used to sit behind the early returns below, so a second use of the weight at a shape the
        # batch does not take left the queued job's `fresh` flag set and the flush overwrote the immediate contribution)
        for j in pool.sink.wg:
            if j[2].data_ptr() == dw_rows.data_ptr():
                j[6] = False
                return False

        def declined():
            # the caller now accumulates into dw_rows at once: a job queued for it LATER in this scope must add as well
            pool.sink.wg_done.add(dw_rows.data_ptr())
            return False
        if switches.WGRAD_BATCH_OFF or x.dtype != torch.bfloat16 or gy.dtype != torch.bfloat16:
            return declined()
        n, h, w, cin = x.shape
        cout = gy.shape[-1]
        key = (n, h, w, cin, cout, rects is not None)
        ok = _WGRAD_BATCH_OK.get(key)
        if ok is None:
            ok = _WGRAD_BATCH_OK[key] = bool(L.lib().s2e_wgrad_batch_supported(L.S2E_BF16, n, h, w, cin, cout)) and \
                (rects is None or (h % 16 == 0 and w % 16 == 0))
        if not ok:
            return declined()
        _need(x, gy, dw_rows, dbias)
        # a gradient arena that zero_grad cleared right before this step, and nothing queued for it yet: single-owner tiles are stored
        fresh = ZeroPool.grad_is_fresh(dw_rows) and dw_rows.data_ptr() not in pool.sink.wg_done
        pool.sink.wg.append([x, gy, dw_rows, dbias, rects, tag, fresh])          # (gy_shared: see is_pinned)
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
            ws = self._wg_ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)     # (kept: the same ~216 MiB every flush: 3 slots x 256 workgroups x 288 KB)
        LaunchProfiler.run('conv_wgrad_patch', flops, lambda: L.check(
            L.lib().s2e_wgrad_batch(L.S2E_BF16, C.byref(arr), len(jobs), _p(ws), wsb, _stream()), 's2e_wgrad_batch'),
            tag='W k3 s1 x%d batched' % len(jobs), nbytes=nbytes, executed=executed)
        self.keep_wg = jobs                                  # the tensors stay referenced until the next flush (stream order covers the rest)

    @staticmethod
    def is_pinned(t):
        """Is `t` the output gradient of a weight-gradient job queued in the open trainer step (not launched yet)?  Whoever would
        modify it in place (ModulateFn's relay) must write elsewhere: the queue holds the tensor itself, not a copy."""
        pool = ZeroPool.active()
        if pool is None or t is None:
            return False
        p = t.data_ptr()
        return any(j[1].data_ptr() == p for j in pool.sink.wg) or any(j[1].data_ptr() == p for j in pool.sink.gwg)

    @staticmethod
    def push_gwg(x, gy, dw, dbias, desc_key, gy_shared=False):
        """Queue a GENERIC weight gradient (a shape s2e_conv2d_wgrad would run in its implicit-GEMM kernel: the 1x1 shortcuts, netE's
        stride-2 layers, the PatchGAN's 4x4 layers, the 8x8 maps) to be accumulated into dw (Cout, KH*KW*Cin) / dbias at the next
        flush, every queued job in ONE launch (s2e_conv2d_wgrad_multi).  desc_key: the ConvDesc fields of the forward conv.  False:
        not queued (no trainer step open, not bf16, not a generic shape, a dW that already has a queued job, S2E_WGRAD_MULTI=0)."""
        pool = ZeroPool.active()
        if pool is None or switches.WGRAD_MULTI_OFF or x.dtype != torch.bfloat16 or gy.dtype != torch.bfloat16:
            return False
        ok = _WGRAD_MULTI_OK.get(desc_key)
        if ok is None:
            d = L.ConvDesc(*desc_key)
            ok = _WGRAD_MULTI_OK[desc_key] = bool(L.lib().s2e_conv2d_wgrad_multi_supported(L.S2E_BF16, C.byref(d)))
        if not ok or any(j[2].data_ptr() == dw.data_ptr() for j in pool.sink.gwg):      # (the reduction adds without atomics: one job per dW)
            return False
        _need(x, gy, dw, dbias)
        pool.sink.gwg.append((x, gy, dw, dbias, desc_key))                          # (gy_shared: see is_pinned)
        return True

    def _flush_gwg(self):
        jobs, self.gwg = self.gwg, []
        dev = jobs[0][0].device
        arr = (L.WgradMultiJob * len(jobs))()
        flops = nbytes = 0.0
        for a, (x, gy, dw, db, key) in zip(arr, jobs):
            a.x, a.gy, a.dw, a.dbias = x.data_ptr(), gy.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
            a.d = L.ConvDesc(*key)
            flops += 2.0 * key[0] * key[4] * key[5] * key[3] * key[6] * key[7] * key[8]
            nbytes += (x.numel() + gy.numel()) * 2.0 + dw.numel() * 4.0
        wsb = int(L.lib().s2e_conv2d_wgrad_multi_workspace_bytes(L.S2E_BF16, C.byref(arr), len(jobs)))
        ws = self.__dict__.get('_gwg_ws')
        if wsb and (ws is None or ws.device != dev or ws.numel() * 4 < wsb):
            if ZeroPool._active is not None and ZeroPool._active.frozen:
                ws, wsb = None, 0                            # (a captured graph must not start using new memory: those jobs add with atomics)
            else:
                ws = self._gwg_ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
        LaunchProfiler.run('conv_wgrad', flops, lambda: L.check(
            L.lib().s2e_conv2d_wgrad_multi(L.S2E_BF16, C.byref(arr), len(jobs), _p(ws) if wsb else None, wsb, _stream()), 's2e_conv2d_wgrad_multi'),
            tag='W generic x%d multi' % len(jobs), nbytes=nbytes)
        self.keep_gwg = jobs                                 # the tensors stay referenced until the next flush (stream order covers the rest)

    @staticmethod
    def c8_would_queue(dtype, h, w, dw, db):
        """Would push_c8 take this layer (so that its caller may leave d(actv) undefined outside the rectangle list it passes)?"""
        return (ZeroPool.active() is not None and dw is not None and db is not None and dtype == torch.bfloat16
                and bool(L.lib().s2e_wgrad_c8_batch_supported(L.S2E_BF16, h, w, 128)))

    @staticmethod
    def push_c8(oh, dactv, dw, db, ncls, rects=None):
        """Queue the weight / bias gradient of a 3x3 conv on the 8-channel one-hot map `oh` (N,h,w,8) with output gradient
        `dactv` (N,h,w,128), accumulated straight into dw (128,ncls,3,3) / db (128) fp32 at the next flush -- all queued layers
        in one launch per slab shape (s2e_wgrad_c8_batch).  rects = (rect_list, counts): only the pixels of those 16 x 16 rectangles
        contribute (the label-sparse backward: dactv is defined there only).  False: not queued (no scope, or a shape the batch
        does not take)."""
        pool = ZeroPool.active()
        if pool is None or dw is None or db is None or oh.dtype != torch.bfloat16:
            return False
        n, h, w, _ = oh.shape
        if dactv.shape[-1] != 128 or not L.lib().s2e_wgrad_c8_batch_supported(L.S2E_BF16, h, w, 128):
            return False
        if rects is not None and ((h | w) & 15):
            return False
        pool.sink.c8.append((oh, dactv, dw, db, int(ncls), rects))
        return True

    def _flush_c8(self):
        c8, self.c8 = self.c8, []
        n = c8[0][0].shape[0]
        rest = [j for j in c8 if j[0].shape[0] != n]         # (a launch shares one batch size: other sizes go in a round of their own)
        if rest:
            c8 = [j for j in c8 if j[0].shape[0] == n]
            self.c8 = rest
        arr = (L.WgradC8Job * len(c8))()
        for i, (oh, dactv, dw, db, ncls, rects) in enumerate(c8):
            arr[i].x, arr[i].gy, arr[i].dw_oihw, arr[i].dbias = oh.data_ptr(), dactv.data_ptr(), dw.data_ptr(), db.data_ptr()
            arr[i].H, arr[i].W, arr[i].ncls = oh.shape[1], oh.shape[2], ncls
            if rects is not None:
                arr[i].rect_list, arr[i].rect_count = rects[0].data_ptr(), rects[1].data_ptr()
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
        # launches of <= 16 jobs; two jobs that add into the SAME parameter gradients (a SPADE module applied twice before a flush)
        # never share one: the apply kernel adds with plain read-modify-writes, one job per blockIdx.y
        chunks, cur, seen = [], [], set()
        for j in jobs:
            keys = {t.data_ptr() for t in (j[5], j[6], j[7], j[8]) if t is not None}
            if len(cur) == 16 or (keys & seen):
                chunks.append(cur)
                cur, seen = [], set()
            cur.append(j)
            seen |= keys
        if cur:
            chunks.append(cur)
        for chunk in chunks:
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
        if (self.wg and self.gwg and switches.FLUSH_STREAMS and self.gwg[0][0].is_cuda
                and not ({j[2].data_ptr() for j in self.wg} & {j[2].data_ptr() for j in self.gwg})):      # (never a dW in both launches)
            # the two big launches of a flush -- the batched patch-resident weight gradients (one persistent workgroup per CU) and the
            # multi-job generic ones -- share no output and leave half of each CU's registers / LDS free: side by side (16.35 -> 16.29 ms, two same-box pairs)
            main = torch.cuda.current_stream()
            side = self.__dict__.get('_side')
            if side is None or side.device != main.device:
                side = self._side = torch.cuda.Stream(device=main.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._flush_gwg()
            self._flush_wgrad()
            main.wait_stream(side)
        if self.wg:
            self._flush_wgrad()                              # first: the re-layout / chain-rule / rank-1 jobs below read or add to its results
        if self.gwg:
            self._flush_gwg()                                # (likewise: its dW buffers feed the re-layout / chain-rule jobs)
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
_WGRAD_BATCH_OK = {}
_WGRAD_MULTI_OK = {}


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


def colsum(g):
    _need(g)
    c = g.shape[-1]
    out = torch.zeros(c, dtype=torch.float32, device=g.device)
    L.check(L.lib().s2e_colsum(_dt(g), _p(g), g.numel() // c, c, _p(out), _stream()), 's2e_colsum')
    return out
_byref = C.byref          # (functions below use C for a channel count)


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
