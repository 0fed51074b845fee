"""SpectralBank: fused spectral normalisation for every spectral-normed conv of one network.

torch.nn.utils.spectral_norm stays registered on the conv modules -- it defines the state_dict layout
(`weight_orig`, `weight_u`, `weight_v`) -- but its pre-forward hook is never run on the HIP path.
Instead the owning network calls `bank.step(training)` once at the top of its forward: one batched
power iteration (4 launches for all layers), after which each conv's sigma is a device scalar that
the weight-pack kernel divides by on the fly.  Semantics are those of the hook (reference
architecture.py:30-34, normalization.py:25-26): one iteration per forward in train
mode updating u, v in place, none in eval mode.

u and v of all layers are re-homed into one flat arena (like the parameters in optim.FlatAdam) so a
single clone per forward snapshots the values the backward needs."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

SN_EPS = 1e-12
_CHAIN_OFF = False      # (S2E_SN_CHAIN, retired in round 5: the two-launch power iteration of the small banks measured 0.88 -> 0.75 ms per step in round 2)


def lib_chain_max_cols():
    return L.lib().s2e_sn_chain_max_cols()


def sn_convs(root):
    """The spectral-normed convs under `root`, memoised on the module (every block's forward asks; walking the module tree
    each time cost 2 ms of host time per eager step).  The memo is keyed on the number of modules in the whole subtree
    (~50 us for the generator's 260), so adding, removing or wrapping a submodule at any depth re-scans."""
    memo = root.__dict__.get('_sn_convs_memo')
    n = sum(1 for _ in root.modules())
    if memo is not None and memo[0] == n and all(hasattr(m, 'weight_orig') for m in memo[1]):
        return memo[1]
    convs = [m for m in root.modules() if isinstance(m, torch.nn.Conv2d) and hasattr(m, 'weight_orig')]
    root.__dict__['_sn_convs_memo'] = (n, convs)
    return convs


class SpectralBank:
    def __init__(self, root):
        self.root = root
        self.convs = sn_convs(root)
        self.n = len(self.convs)
        for i, c in enumerate(self.convs):
            c.__dict__['_sn_bank'] = self
            c.__dict__['_sn_index'] = i
        self._ptrs = None
        self.sigma = None
        self.uv_snap = None
        self._snap_bufs, self._sigma_bufs = [], []
        self._scope_serial, self._scope_calls = -1, 0

    # ------------------------------------------------------------------ device tables
    def _build(self):
        dev = self.convs[0].weight_orig.device
        rows = [c.weight_orig.shape[0] for c in self.convs]
        cols = [c.weight_orig[0].numel() for c in self.convs]
        # flat arena for u | v of every layer; module buffers become views of it
        sizes = []
        for r, c_ in zip(rows, cols):
            sizes += [r, c_]
        offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])])
        arena = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self.uv_off = []
        with torch.no_grad():
            for i, c in enumerate(self.convs):
                ou, ov = int(offs[2 * i]), int(offs[2 * i + 1])
                uview, vview = arena[ou:ou + rows[i]], arena[ov:ov + cols[i]]
                uview.copy_(c.weight_u)
                vview.copy_(c.weight_v)
                c._buffers['weight_u'] = uview
                c._buffers['weight_v'] = vview
                self.uv_off.append((ou, ov))
        self.uv_arena = arena
        # t | s (64-bit fixed point), same offsets; twice: small banks alternate between two accumulator pairs (chain mode)
        tot = int(offs[-1])
        self.scratch = torch.zeros(2 * tot, dtype=torch.int64, device=dev)
        self.chain = int(max(cols) <= lib_chain_max_cols() and not _CHAIN_OFF)
        lib = L.lib()
        shapes = []
        for which in (0, 1):                                  # the W tile one workgroup takes: W v pass, W^T u pass
            br, bc = C.c_int(), C.c_int()
            lib.s2e_sn_block_shape(which, C.byref(br), C.byref(bc))
            shapes.append((br.value, bc.value))
        table = (L.SnLayer * self.n)()
        maps = ([], [])
        for i, c in enumerate(self.convs):
            ou, ov = self.uv_off[i]
            table[i].w = c.weight_orig.data_ptr()
            table[i].u = arena.data_ptr() + 4 * ou
            table[i].v = arena.data_ptr() + 4 * ov
            table[i].s = self.scratch.data_ptr() + 8 * ou
            table[i].t = self.scratch.data_ptr() + 8 * ov
            table[i].s2 = self.scratch.data_ptr() + 8 * (tot + ou)
            table[i].t2 = self.scratch.data_ptr() + 8 * (tot + ov)
            table[i].rows, table[i].cols = rows[i], cols[i]
            w = c.weight_orig
            if w.dim() == 4 and not w.is_contiguous():
                # a master stored channels-last (optim.FlatAdam): W's memory columns run (tap, ci); weight_v keeps torch's order
                # and the kernels translate (s2e_sn_layer.cin / taps)
                if not w.permute(0, 2, 3, 1).is_contiguous() or w.shape[1] % 8:
                    raise L.Seg2EyeHipError('spectral norm: weight_orig of %s is neither contiguous nor channels-last' % (tuple(w.shape),))
                table[i].cin, table[i].taps = w.shape[1], w.shape[2] * w.shape[3]
            for bm, (_BR, _BC) in zip(maps, shapes):
                for r0 in range(0, rows[i], _BR):
                    for c0 in range(0, cols[i], _BC):
                        bm.append((i, r0, c0))
        raw = np.frombuffer(bytes(table), dtype=np.uint8).copy()
        self.table_dev = torch.from_numpy(raw).to(dev)
        self.block_map = torch.tensor(maps[0], dtype=torch.int32, device=dev)
        self.block_map_t = torch.tensor(maps[1], dtype=torch.int32, device=dev)
        self.rows, self.cols = rows, cols
        self._ptrs = self._current_ptrs()

    def ensure_built(self):
        """Build (or rebuild after the parameters moved) the device tables now instead of at the next forward."""
        if self.n and (self._ptrs is None or self._ptrs != self._current_ptrs()):
            self._build()

    def _current_ptrs(self):
        return tuple((c.weight_orig.data_ptr(), c.weight_u.data_ptr(), c.weight_v.data_ptr(), c.weight_orig.stride()) for c in self.convs)

    # ------------------------------------------------------------------ per-forward
    def step(self, training, iterations=1):
        """Run the power iteration(s) for every layer; afterwards sigma_of()/uv_of() serve this forward."""
        if self.n == 0:
            return
        w = self.convs[0].weight_orig
        if not w.is_cuda:
            raise L.Seg2EyeHipError('spectral norm runs on the GPU only (no CPU fallback)')
        if self._ptrs is None or self._ptrs != self._current_ptrs():
            self._build()                                    # first use, or storage moved (.cuda(), optimizer arena)
        # Inside a trainer step (ops.ZeroPool scope) sigma and the u|v snapshot live in PERSISTENT buffers: same addresses
        # every step, which the batched gradient kernels (device job tables cached by pointer) and hipGraphs need.  They
        # then hold the values of the LATEST forward -- the trainer always runs a network's backward before its next
        # forward.  Anywhere else every forward gets its own sigma / snapshot, so an older forward can still be
        # differentiated after a newer one.
        from .ops import ZeroPool
        persistent = ZeroPool.active() is not None
        if persistent:
            # a network may run more than once per step (netE encodes the real styles and, with the style-consistency
            # losses on, the generated image): the i-th forward of a step gets the i-th persistent buffer pair
            if self._scope_serial != ZeroPool.serial:
                self._scope_serial, self._scope_calls = ZeroPool.serial, 0
            i = self._scope_calls
            self._scope_calls += 1
            while len(self._sigma_bufs) <= i:
                self._sigma_bufs.append(torch.empty(self.n, dtype=torch.float32, device=w.device))
                self._snap_bufs.append(torch.empty_like(self.uv_arena))
            if self._sigma_bufs[i].device != w.device or self._snap_bufs[i].shape != self.uv_arena.shape:
                self._sigma_bufs[i] = torch.empty(self.n, dtype=torch.float32, device=w.device)
                self._snap_bufs[i] = torch.empty_like(self.uv_arena)
            self.sigma = self._sigma_bufs[i]
        else:
            self.sigma = torch.empty(self.n, dtype=torch.float32, device=w.device)
        from .ops import LaunchProfiler
        wbytes = 4.0 * sum(r * c for r, c in zip(self.rows, self.cols))
        LaunchProfiler.run('spectral_norm', 0.0, lambda: L.check(L.lib().s2e_sn_power_iteration(
            self.table_dev.data_ptr(), self.n, self.block_map_t.data_ptr(), self.block_map_t.shape[0],
            self.block_map.data_ptr(), self.block_map.shape[0],
            self.scratch.data_ptr(), self.scratch.numel() * 8, self.sigma.data_ptr(), int(bool(training)),
            int(iterations), SN_EPS, self.chain, torch.cuda.current_stream().cuda_stream), 's2e_sn_power_iteration'),
            nbytes=wbytes * (2 * int(iterations) if training else 1))     # algorithmic: W^T u and W v each read W once per iteration
        # the backward of this forward needs u, v as they are NOW (later forwards update them in place)
        if not torch.is_grad_enabled():
            self.uv_snap = self.uv_arena
        elif persistent:
            self._snap_bufs[i].copy_(self.uv_arena)
            self.uv_snap = self._snap_bufs[i]
        else:
            self.uv_snap = self.uv_arena.clone()

    def handle(self, conv):
        """(u, v, sigma) tensors for one conv, valid for the forward that called step()."""
        i = conv._sn_index
        ou, ov = self.uv_off[i]
        return (self.uv_snap[ou:ou + self.rows[i]], self.uv_snap[ov:ov + self.cols[i]], self.sigma[i:i + 1])


def ensure_bank(root):
    """Create (and build the device tables of) the bank a network would create on its first forward,
    without running a power iteration.  Returns None for a network without spectral-normed convs."""
    if not sn_convs(root):
        return None
    bank = root.__dict__.get('_sn_owned_bank')
    if bank is None:
        bank = SpectralBank(root)
        root.__dict__['_sn_owned_bank'] = bank
    if bank._ptrs is None or bank._ptrs != bank._current_ptrs():
        bank._build()
    return bank


def sn_begin(root, iterations=1):
    """Call at the top of a network's forward.  A module steps the bank only if it owns it, so a block
    used stand-alone (tests) works, and inside a network the network's single step covers it."""
    bank = root.__dict__.get('_sn_owned_bank')
    convs = sn_convs(root)
    if not convs:
        return None
    outer = convs[0].__dict__.get('_sn_bank')
    if outer is not None and outer.root is not root and bank is None:
        return outer                                         # an enclosing network owns and steps it
    if bank is None:
        bank = SpectralBank(root)
        root.__dict__['_sn_owned_bank'] = bank
    else:
        for i, c in enumerate(bank.convs):                   # re-claim (a stand-alone use may have re-pointed them)
            c.__dict__['_sn_bank'] = bank
            c.__dict__['_sn_index'] = i
    bank.step(root.training, iterations)
    return bank


def conv_params(conv):
    """(weight, bias, sn) for ops.conv2d: sn = (u, v, sigma) when the conv is spectral-normed."""
    bias = getattr(conv, 'bias', None)
    if hasattr(conv, 'weight_orig'):
        bank = conv.__dict__.get('_sn_bank')
        if bank is None or bank.sigma is None:
            raise RuntimeError('spectral-normed conv used before sn_begin() of its network')
        return conv.weight_orig, bias, bank.handle(conv)
    return conv.weight, bias, None
