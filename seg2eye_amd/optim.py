"""Flat-arena Adam: every parameter of a group lives in ONE fp32 buffer (and its gradient in another),
so the optimizer step is a single HIP launch (s2e_adam_flat) and the data-parallel gradient exchange is
an all-reduce over contiguous slices with no packing.  Mirrors torch.optim.Adam as the reference
configures it (models/pix2pix_model.py:92-110): TTUR betas (0, 0.9), eps 1e-8, weight_decay 0."""
import os

import torch

from . import ops

_ALIGN = 4          # elements; keeps every parameter 16-byte aligned inside the arena
_ALIGN_CL = 64      # elements; channels-last conv weights start on 256-byte boundaries (see FlatAdam.__init__)
# S2E_WEIGHTS_CL=0: every parameter keeps torch's (Cout, Cin, KH, KW) memory order in the arenas (A/B switch)
_WEIGHTS_CL = os.environ.get('S2E_WEIGHTS_CL', '1') != '0'


def _arena_view(flat, off, p, cl):
    """The parameter-shaped view of its arena slice: plain, or -- conv weights, cl -- the (Cout, Cin, KH, KW) view of a slice
    stored [Cout][KH][KW][Cin] (torch's channels_last): same values, same logical shape, other strides."""
    t = flat[off:off + p.numel()]
    if cl:
        co, ci, kh, kw = p.shape
        return t.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return t.view(p.shape)


class FlatAdam:
    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, never_updated=(), channels_last=None):
        """channels_last (default: on, S2E_WEIGHTS_CL): a conv weight whose input-channel count is a multiple of 8 is stored
        [Cout][KH][KW][Cin] in the arenas -- the order the MFMA kernels' packed matrices and weight gradients use -- and exposed
        as a (Cout, Cin, KH, KW) view: state_dict values, shapes and every element-wise use are unchanged, while the
        weight-gradient kernels accumulate straight into `.grad` and the packs read rows instead of gathering.
        never_updated: parameters that are in the optimizer but never receive a gradient (netE.fc_var: `logvar` enters no
        loss, pix2pix_model.py:307-314).  torch.optim.Adam SKIPS a parameter whose .grad is None -- no moment update and, what
        matters, no weight decay -- so they are laid out at the END of the arenas and the Adam launch stops before them."""
        skip = {id(p) for p in never_updated}
        seen, plist = set(), []
        for p in list(params) + [p for p in never_updated]:
            if id(p) not in seen and id(p) not in skip:
                seen.add(id(p))
                plist.append(p)
        n_active_params = len(plist)
        for p in never_updated:
            if id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError('FlatAdam got an empty parameter list')
        dev = plist[0].device
        use_cl = _WEIGHTS_CL if channels_last is None else bool(channels_last)
        # (by default only for parameters on the GPU -- the layout exists for the HIP kernels; an explicit True lays out CPU tensors too)
        self.cl = [bool(use_cl and p.dim() == 4 and p.shape[1] % 8 == 0 and (p.is_cuda or channels_last is True)) for p in plist]
        self.offsets, n = [], 0
        for p, cl in zip(plist, self.cl):
            if p.dtype != torch.float32:
                raise TypeError('FlatAdam keeps fp32 master parameters')
            if cl and n % _ALIGN_CL:
                # a channels-last weight starts on a 256-byte boundary: its gradient is the target of the weight-gradient
                # kernels' 128-byte row-segment atomics, which cost per cache line touched (measured +25 % on the 16-byte-aligned
                # slices).  Stacked weights ([W_gamma; W_beta]: sizes are multiples of 64 elements) stay back to back.
                n = (n + _ALIGN_CL - 1) // _ALIGN_CL * _ALIGN_CL
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.numel_active = self.offsets[n_active_params] if n_active_params < len(plist) else n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off, cl in zip(plist, self.offsets, self.cl):
                view = _arena_view(self.flat_p, off, p, cl)
                view.copy_(p.data)
                p.data = view                                           # parameter now aliases the arena
                p.grad = _arena_view(self.flat_g, off, p, cl)           # autograd accumulates in place
        self.params = plist
        self._grad_strides = [p.grad.stride() for p in plist]
        self.betas = (float(betas[0]), float(betas[1]))                   # (0, 0.9) must be floats
        self.eps = float(eps)
        self.step_count = 0
        self.param_groups = [{'params': plist, 'lr': float(lr)}]         # update_learning_rate writes ['lr']
        # device-resident hyper-parameters {lr, beta1, beta2, eps, completed steps, grad_scale, weight_decay}
        self._wd = float(weight_decay)
        self.hyper = torch.tensor([float(lr), self.betas[0], self.betas[1], self.eps, 0.0, 1.0, float(weight_decay)],
                                  dtype=torch.float32, device=dev)
        self._hyper_host = (float(lr), 1.0)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        self._g_is_last_step = False
        ops.ZeroPool.arena_zeroed(self.flat_g)               # (lets the in-place spectral-norm chain rule know the arena is fresh)

    def rebind_grads(self):
        """Re-attach .grad views if something replaced them (e.g. a zero_grad(set_to_none=True))."""
        base, strides = self.flat_g.data_ptr(), self._grad_strides
        for i, p in enumerate(self.params):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * self.offsets[i] and g.stride() == strides[i]:
                continue                                     # (the common case, checked without building a view per parameter)
            want = _arena_view(self.flat_g, self.offsets[i], p, self.cl[i])
            if g is not None:
                want.copy_(g)
            p.grad = want

    def sync_hyper(self, grad_scale=1.0):
        """Push lr / grad_scale to the device only when they changed (never inside a graph)."""
        cur = (float(self.param_groups[0]['lr']), float(grad_scale))
        if cur != self._hyper_host:
            self.hyper[0:1].fill_(cur[0])
            self.hyper[5:6].fill_(cur[1])
            self._hyper_host = cur

    def step(self, grad_scale=1.0):
        self.rebind_grads()
        self.sync_hyper(grad_scale)
        self.step_count += 1
        k = self.numel_active
        ops.adam_flat_step(self.flat_p[:k], self.flat_g[:k], self.flat_m[:k], self.flat_v[:k], self.hyper,
                           skips_m=self.betas[0] == 0.0 and self._wd == 0.0)
        self._g_is_last_step = True

    def _layout(self):
        """What the flat moment arrays mean: element i belongs to which parameter, in which memory order.  `shapes` is the
        per-parameter signature in arena order: a state is only ever loaded into an arena whose parameters have the same
        shapes in the same order (a length check alone accepts a permuted parameter list)."""
        return {'offsets': list(self.offsets), 'channels_last': [bool(c) for c in self.cl],
                'shapes': [tuple(int(d) for d in p.shape) for p in self.params]}

    def _first_moment(self):
        """The first-moment arena as torch.optim.Adam would hold it.  With beta1 == 0 and no weight decay (the reference's TTUR
        setting) s2e_adam_flat does not touch flat_m -- m_t = g_t * grad_scale exactly, whatever m was -- so it is formed here from
        the gradient arena when that still holds the gradients of the last step (between step() and the next zero_grad()).
        (With beta1 == 0 the saved m never influences a resumed run: the next step overwrites it.)"""
        if self.betas[0] == 0.0 and self._wd == 0.0 and self.step_count > 0 and self.__dict__.get('_g_is_last_step', False):
            k = self.numel_active
            m = self.flat_m.clone()
            m[:k] = self.flat_g[:k] * self._hyper_host[1]
            return m
        return self.flat_m

    def state_dict(self):
        """`m_valid`: False when the saved first moment is NOT the optimizer's -- beta1 == 0 / no weight decay steps leave
        flat_m untouched, and outside the window between step() and the next zero_grad() the gradient it would be formed from is
        gone.  Harmless for a resume at beta1 == 0 (the next step overwrites m); a run resumed with another beta1 (--no_TTUR: 0.5)
        must not trust an invalid m: load_state_dict then starts it from zeros."""
        skipped = self.betas[0] == 0.0 and self._wd == 0.0 and self.step_count > 0
        valid = (not skipped) or bool(self.__dict__.get('_g_is_last_step', False))
        return {'step': self.step_count, 'm': self._first_moment(), 'v': self.flat_v, 'lr': self.param_groups[0]['lr'], 'layout': self._layout(),
                'm_valid': valid}

    def load_state_dict(self, sd, trust_param_order=False):
        """Moments saved by another FlatAdam.  With a `layout` that carries `shapes` the parameter signature must match; the
        memory order (channels-last flags, alignment gaps) may differ and is converted per parameter by logical index.  A state
        WITHOUT a signature (saved before round 5) is loaded only into the identical arena; a state without any layout (saved
        before channels-last masters / with S2E_WEIGHTS_CL=0) only with `trust_param_order=True`: nothing in it says which
        parameter a moment belongs to, and this build orders the generator's arena differently from the builds that wrote such
        states (pix2pix_model.create_optimizers), so a positional load would attach moments to the wrong weights silently."""
        lay = sd.get('layout')
        mine = self._layout()
        if lay is not None:
            lay = {k: ([tuple(x) for x in v] if k == 'shapes' else list(v)) for k, v in lay.items()}
        if lay is not None and 'shapes' in lay:
            if lay['shapes'] != mine['shapes']:
                raise ValueError('FlatAdam.load_state_dict: the saved moments belong to another parameter list (shapes / order '
                                 'differ from this optimizer\'s): refusing a positional load')
            if lay['offsets'] != mine['offsets'] or lay['channels_last'] != mine['channels_last']:
                return self._load_converted(sd, lay['offsets'], lay['channels_last'])
        elif lay is not None:
            if lay != {k: v for k, v in mine.items() if k != 'shapes'}:
                raise ValueError('FlatAdam.load_state_dict: the saved moments were laid out for another arena (parameter order / '
                                 'channels-last weights differ) and carry no parameter signature')
        else:
            if not trust_param_order:
                raise ValueError('FlatAdam.load_state_dict: the state has no layout / parameter signature (saved before '
                                 'channels-last masters or with S2E_WEIGHTS_CL=0); pass trust_param_order=True only if the '
                                 'parameter ORDER of the optimizer that wrote it is known to equal this one\'s')
            if len(sd['m']) != sum((p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for p in self.params):
                raise ValueError('FlatAdam.load_state_dict: layout-less state of another total length')
            return self._load_torch_order(sd)
        self.step_count = int(sd['step'])
        self.flat_m.copy_(sd['m'])
        self.flat_v.copy_(sd['v'])
        self._finish_load(sd)

    def _finish_load(self, sd):
        if not sd.get('m_valid', True) and not (self.betas[0] == 0.0 and self._wd == 0.0):
            self.flat_m.zero_()                              # (saved by a beta1 == 0 run outside the step window: see state_dict)
        self.step_count = int(sd['step'])
        self.param_groups[0]['lr'] = float(sd['lr'])
        self.hyper[4:5].fill_(float(self.step_count))
        self.sync_hyper(self._hyper_host[1])

    def _load_converted(self, sd, offsets, cl_flags):
        """Same parameters (signature checked by the caller), other memory order: copied per parameter through views."""
        with torch.no_grad():
            for i, p in enumerate(self.params):
                for flat, key in ((self.flat_m, 'm'), (self.flat_v, 'v')):
                    src = _arena_view(sd[key].to(flat.device), offsets[i], p, cl_flags[i])
                    _arena_view(flat, self.offsets[i], p, self.cl[i]).copy_(src)
        self._finish_load(sd)

    def _load_torch_order(self, sd):
        """A state dict without 'layout': moments saved when every parameter lay in torch's order, back to back (4-element
        alignment) -- before channels-last masters, or with S2E_WEIGHTS_CL=0.  Copied per parameter BY LOGICAL INDEX through
        the arena views, so channels-last slices and their 256-byte alignment gaps do not matter."""
        off = 0
        with torch.no_grad():
            for i, p in enumerate(self.params):
                n = p.numel()
                for flat, key in ((self.flat_m, 'm'), (self.flat_v, 'v')):
                    _arena_view(flat, self.offsets[i], p, self.cl[i]).copy_(sd[key][off:off + n].view(p.shape))
                off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self._finish_load(sd)
