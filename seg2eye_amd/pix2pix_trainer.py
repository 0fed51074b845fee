"""Pix2PixTrainer (reference trainers/pix2pix_trainer.py:8-88): owns the model and the two optimizers,
runs one G step / one D step, LR decay, save.  Multi-GPU: when torch.distributed is initialised the
flat gradient arenas are sum-all-reduced (RCCL) between backward and the Adam launch."""
import os

from .distributed import FlatGradSync, broadcast_buffers, broadcast_flat, exchange_active, world_size
from .ops import ZeroPool
from .pix2pix_model import Pix2PixModel


def _total(losses):
    """sum(losses.values()).mean() (trainers/pix2pix_trainer.py:31,41) without the launches that change nothing: no `0 + l` to
    start the sum, no mean over a single element."""
    vals = list(losses.values())
    t = vals[0]
    for v in vals[1:]:
        t = t + v
    return t.view(()) if t.numel() == 1 else t.mean()


class _StepGraph:
    """One captured step body: a single hipGraph, or a chain of segments cut where a gradient group completes
    (`groups[k]` = the arena group that is final after segment k)."""

    def __init__(self, segments, groups):
        self.segments, self.groups = segments, groups

    def pool(self):
        return self.segments[0].pool()

    def replay(self, launch=None):
        for k, g in enumerate(self.segments):
            g.replay()
            if launch is not None and k < len(self.groups):
                launch(self.groups[k])


class Pix2PixTrainer:
    def __init__(self, opt):
        self.opt = opt
        self.pix2pix_model = Pix2PixModel(opt)
        self.pix2pix_model_on_one_gpu = self.pix2pix_model
        self.generated = None
        self.g_losses, self.d_losses = {}, {}
        self._static, self.graph_G, self.graph_D = None, None, None
        self.pool = ZeroPool(self.pix2pix_model.device())    # this trainer's zero-filled scratch + deferred gradient re-layouts
        if opt.isTrain:
            self.optimizer_G, self.optimizer_D = self.pix2pix_model_on_one_gpu.create_optimizers(opt)
            self.old_lr = opt.lr
            how = dict(payload=getattr(opt, 'grad_dtype', 'fp32'), algorithm=getattr(opt, 'grad_exchange', 'allreduce'))
            self.sync_G = FlatGradSync(self.optimizer_G.flat_g, groups=self.pix2pix_model.grad_groups_G, **how)
            self.sync_D = FlatGradSync(self.optimizer_D.flat_g, **how)
            self._segments = None                            # while the G step is being captured in segments: (graphs, groups, mode)
            self._quiet_hooks = False                        # capture warm-up passes: hooks flush but exchange nothing
            if exchange_active() and not getattr(opt, 'no_overlap_allreduce', False):
                # Overlap the gradient exchange with the backward pass: the generator reports when a group of
                # blocks has all its gradients (networks/generator.py); we flush that group's queued gradient re-layouts and
                # start the all-reduce of its arena slice.  With --hip_graphs the G step is captured as one graph SEGMENT per
                # group (round 4): a hook that fires during capture ends the current segment and begins the next, and a replayed
                # step is `replay segment k; start group k's all-reduce` -- the collective runs on the backend's stream while the
                # next segment replays.  (Rounds 1-3 had to choose: eager launches + overlap, or graphs + one exposed exchange.)
                self.pix2pix_model.netG.__dict__['grad_ready'] = self._group_ready
            if exchange_active() and 'batch' in str(getattr(opt, 'norm_G', '')) and getattr(opt, 'hip_graphs', False):
                # BatchNorm SPADE under data parallelism exchanges its batch statistics between the replicas INSIDE every forward
                # and backward (normalization.py::spade_stats): a collective cannot be captured into a hipGraph (gloo synchronises
                # the stream on the host; attempting it took the process down), so these runs launch eagerly
                from .distributed import get_rank
                if get_rank() == 0:
                    import sys
                    print('seg2eye_amd: --norm_G %s with %d replicas exchanges batch statistics inside the step: the steps run '
                          'as individual launches (no hipGraphs)' % (opt.norm_G, world_size()), file=sys.stderr)
                self.opt.hip_graphs = False
            broadcast_flat(self.optimizer_G.flat_p)          # identical replicas at step 0: parameters ...
            broadcast_flat(self.optimizer_D.flat_p)
            self.sync_replica_buffers()                      # ... and spectral-norm u, v / BatchNorm running statistics

    def sync_replica_buffers(self):
        """Every replica takes rank 0's spectral-norm u, v and BatchNorm running buffers (no-op on one process).  Called at
        construction and by train.py after rank 0 alone ran a validation pass: that pass runs in train mode (the reference never
        calls eval()) and so advances rank 0's u, v and running statistics."""
        m = self.pix2pix_model
        broadcast_buffers([m.netG, m.netD, m.netE])

    def _group_ready(self, i):
        """Backward hook (data parallel): parameter group i of the G arena is final.  Only during the G step's backward --
        the D step's no-grad generator forward registers no hooks."""
        pool = ZeroPool.active()
        if pool is not self.pool or pool.key != 'G':
            return
        pool.sink.flush()                                    # the group's queued packed-dW -> arena conversions: now
        if self._segments is not None:
            self._next_segment(i)                            # capturing: the segment ends here, replay launches the exchange
        elif not self._quiet_hooks:
            self.sync_G.launch(i)

    def _next_segment(self, group):
        import torch
        segs, groups, mode = self._segments
        segs[-1].capture_end()
        groups.append(group)
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=segs[0].pool(), capture_error_mode=mode)
        segs.append(g)

    def _train_mode(self):
        """model.train() -- when something put it or ANY of its modules into eval mode (a validation pass, a caller freezing one
        SPADE block: spectral norm's power iteration and BatchNorm's running statistics stop there).  Reading the ~260 cached
        flags costs ~15 us; walking `modules()` to see whether the module LIST changed (a module added or replaced
        later must be seen) costs a few hundred us, so that walk runs on the first call and then every 64th --
        `invalidate_modules()` forces it at once."""
        mods = self.__dict__.get('_all_modules')
        tick = self.__dict__['_mode_tick'] = self.__dict__.get('_mode_tick', 0) + 1
        if mods is None or tick % 64 == 1:
            now = list(self.pix2pix_model.modules())
            if mods is None or len(mods) != len(now) or any(a is not b for a, b in zip(mods, now)):
                mods = self.__dict__['_all_modules'] = now
        if not all(x.training for x in mods):
            self.pix2pix_model.train()

    def invalidate_modules(self):
        """Call after adding / replacing a module of the model: the next step re-reads the module list."""
        self.__dict__.pop('_all_modules', None)

    def _one(self):
        """d(total)/d(total) as a persistent device scalar (autograd would launch a ones_like per backward)."""
        t = self.__dict__.get('_one_t')
        if t is None:
            import torch
            t = self.__dict__['_one_t'] = torch.ones((), dtype=torch.float32, device=self.pix2pix_model.device())
        return t

    # ---- step bodies: zero_grad + forward + backward (what a hipGraph captures) -------------------
    def _g_body(self, data):
        self.optimizer_G.zero_grad()
        try:
            with self.pool.scope('G'):                           # all zero-filled scratch of the step: one fill
                g_losses, generated = self.pix2pix_model(data, mode='generator')
                _total(g_losses).backward(self._one())
        except BaseException:
            self.sync_G.reset()                                  # exchanges the backward hooks started for a step that failed
            raise
        # keep detached copies only: a live autograd graph would pin last iteration's AccumulateGrad nodes
        # (and their stream), which breaks hipGraph capture
        self.g_losses = {k: v.detach() for k, v in g_losses.items()}
        self.generated = generated.detach()

    def _d_body(self, data):
        self.optimizer_D.zero_grad()
        with self.pool.scope('D'):
            d_losses = self.pix2pix_model(data, mode='discriminator')
            _total(d_losses).backward(self._one())
        self.d_losses = {k: v.detach() for k, v in d_losses.items()}

    def run_generator_one_step(self, data):
        """trainers/pix2pix_trainer.py:26-35.  With opt.hip_graphs the body is one graph replay."""
        self._train_mode()
        if self.use_graphs and self._stage_inputs(data):     # (captures on first use; turns graphs off if that fails)
            self.graph_G.replay(self.sync_G.launch)          # (segment k, then group k's exchange beside segment k+1)
            for k, v in getattr(self, '_static_log', {}).items():   # the replay refreshed these in place: log this step's values
                self.pix2pix_model.add_to_loss_log(k, v.clone())
        else:
            self._g_body(data)
        self.optimizer_G.step(grad_scale=self.sync_G.all_reduce())

    def run_discriminator_one_step(self, data):
        """trainers/pix2pix_trainer.py:37-45."""
        self._train_mode()
        if self.use_graphs and self._stage_inputs(data):
            self.graph_D.replay()
        else:
            self._d_body(data)
        self.optimizer_D.step(grad_scale=self.sync_D.all_reduce())

    # ---- hipGraph capture ---------------------------------------------------------------------------
    @property
    def use_graphs(self):
        return bool(getattr(self.opt, 'hip_graphs', False))

    def _stage_inputs(self, data):
        """Copy the batch into the static input buffers the graphs read (capturing on first use).  True: replay.  False: this
        step runs as individual launches -- the capture failed (graphs are off from now on), or this batch has another shape
        than the one the graphs were captured for (they stay for the batches that have it)."""
        if self._static is None:
            try:
                self._capture(data)
            except Exception as e:                           # noqa: BLE001 -- any capture failure: run eager instead
                # Eager runs the same kernels (10 % slower on a slow host: DESIGN 6); a failed capture must not take a multi-GPU
                # job down.  The step that triggered the capture has not run yet.
                import sys
                import torch
                print('seg2eye_amd: hipGraph capture failed (%s: %s) -- continuing without graphs' % (type(e).__name__, e),
                      file=sys.stderr)
                if os.environ.get('S2E_DEBUG_CAPTURE'):
                    import traceback
                    traceback.print_exc()
                torch.cuda.synchronize()
                self.opt.hip_graphs = False
                self._static, self.graph_G, self.graph_D = None, None, None
                self.pool.unfreeze()
                return False
        if any(tuple(data[k].shape) != tuple(buf.shape) for k, buf in self._static.items()):
            return False                                     # (an epoch's ragged last batch: this step runs as individual launches)
        for k, buf in self._static.items():
            src = data[k]
            if src.data_ptr() != buf.data_ptr():
                buf.copy_(src, non_blocking=True)
        return True

    def _capture(self, data):
        import torch
        m = self.pix2pix_model
        dev = m.device()
        self._static = {k: data[k].to(dev).clone() for k in ('label', 'style_image', 'target')}
        # the warm-up passes run the spectral-norm power iterations for real: snapshot every bank's u|v
        # arena and put it back afterwards so capturing does not change the training trajectory (weights
        # are untouched: the bodies contain no optimizer step)
        from .spectral import ensure_bank
        banks = [b for b in (ensure_bank(net) for net in (m.netG, m.netD, m.netE)) if b is not None]
        snap = [b.uv_arena.clone() for b in banks]
        # ... and so do BatchNorm's running statistics (--norm_G spectralspadebatch3x3, the reference's default)
        bn_bufs = [t for net in (m.netG, m.netD, m.netE) for mod in net.modules()
                   if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) and mod.track_running_stats
                   for t in (mod.running_mean, mod.running_var, mod.num_batches_tracked)]
        bn_snap = [t.clone() for t in bn_bufs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self._quiet_hooks = True                             # (the warm-up passes exchange nothing: every rank runs them alike)
        try:
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._g_body(self._static)
                    self._d_body(self._static)
        finally:
            self._quiet_hooks = False
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.pool.freeze()                                   # the graphs hold raw pointers into the pool
        # With a process group up, RCCL's watchdog thread polls events while we capture: only calls made by the
        # capturing threads may invalidate the capture (the collectives themselves stay outside the graphs).
        multi = torch.distributed.is_available() and torch.distributed.is_initialized()
        mode = {'capture_error_mode': 'thread_local'} if multi else {}
        # No CYCLIC garbage collection while capturing: a collection that happens to start inside the captured body may destroy
        # objects of an earlier trainer (hipGraph executables, streams) -- runtime calls a capture does not survive: the full test suite
        # aborted once in "Garbage-collecting" under _g_body.  Reference counting frees everything else as always.
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            if m.netG.__dict__.get('grad_ready') is not None:
                graph_G = self._capture_segmented(self._g_body, mode.get('capture_error_mode', 'global'))
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    self._g_body(self._static)
                graph_G = _StepGraph([g], [])
            graph_D = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_D, pool=graph_G.pool(), **mode):
                self._d_body(self._static)
            self.graph_G, self.graph_D = graph_G, graph_D
        finally:
            if gc_was:
                gc.enable()
            with torch.no_grad():
                for b, s0 in zip(banks, snap):
                    b.uv_arena.copy_(s0)
                for t, s0 in zip(bn_bufs, bn_snap):
                    t.copy_(s0)
            # the '*/raw' log entries (pix2pix_model.add_to_loss_log) appended while capturing are the graph's static output
            # tensors: remember them, a replay refreshes their values and run_generator_one_step re-registers them
            self._static_log = {k: v[-1] for k, v in m.loss_log.items() if len(v)}
            m.reset_loss_log()

    def _capture_segmented(self, body, mode):
        """Capture `body` as a chain of hipGraphs that share one memory pool: `_group_ready` -- a backward hook -- closes the
        current segment and opens the next wherever a gradient group completes.  The backward runs on THIS thread while
        capturing (autograd's device thread would end a capture another thread began)."""
        import gc
        import torch
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        segs, groups = [torch.cuda.CUDAGraph()], []
        with torch.cuda.stream(stream), torch.autograd.set_multithreading_enabled(False):
            segs[0].capture_begin(capture_error_mode=mode)
            self._segments = (segs, groups, mode)
            try:
                body(self._static)
            except BaseException:
                self._segments = None
                try:
                    segs[-1].capture_end()                   # (an already invalidated capture raises here: keep the FIRST error)
                except Exception:                            # noqa: BLE001
                    pass
                raise
            self._segments = None
            segs[-1].capture_end()
        torch.cuda.current_stream().wait_stream(stream)
        return _StepGraph(segs, groups)

    def get_latest_losses(self, include_log_losses=False):
        losses = {**self.g_losses, **self.d_losses}
        if include_log_losses:
            losses = {**losses, **self.pix2pix_model_on_one_gpu.get_loss_log()}
            self.pix2pix_model_on_one_gpu.reset_loss_log()
        return losses

    def get_latest_generated(self):
        return self.generated

    def save(self, epoch):
        self.pix2pix_model_on_one_gpu.save(epoch)

    def update_learning_rate(self, epoch):
        """Constant for niter epochs, then linear to 0 over niter_decay, keeping TTUR's 1/2 : 2 ratio
        (pix2pix_trainer.py:68-88)."""
        if epoch > self.opt.niter:
            new_lr = self.old_lr - self.opt.lr / self.opt.niter_decay
        else:
            new_lr = self.old_lr
        if new_lr != self.old_lr:
            new_lr_G, new_lr_D = (new_lr, new_lr) if self.opt.no_TTUR else (new_lr / 2, new_lr * 2)
            for g in self.optimizer_D.param_groups:
                g['lr'] = new_lr_D
            for g in self.optimizer_G.param_groups:
                g['lr'] = new_lr_G
            print('update learning rate: %f -> %f' % (self.old_lr, new_lr))
            self.old_lr = new_lr
