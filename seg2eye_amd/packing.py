"""PackPlan: the weight packs of one network as one launch per forward.

Every conv on the HIP path consumes its weight as an MFMA B-operand matrix in the compute dtype
(`s2e_pack_conv_weight`; spectral norm's 1/sigma folded in), and the data-gradient consumes the
transposed pack.  A generator forward needs ~45 such matrices, ~45 more for its backward; as individual
launches of 5-15 us each (most layers are small) they cost 1.6 ms of a 35 ms step.  A plan learns the set
of packs a network asks for during its first forward/backward (those still go one by one), then packs all
of them with ONE `s2e_pack_conv_weights` launch at the top of every later forward, into persistent
buffers that `lookup` hands to the convs.

Validity: the buffers hold the weights (and sigma) of the LAST forward of the network.  Each forward
bumps `generation`; a backward whose forward is not the latest one falls back to packing for itself.
The plan keeps references to the weight tensors it reads, so a re-homed parameter can never leave a
dangling pointer behind -- its old entry is simply never asked for again."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_current = None          # the plan of the network whose forward is running


def current():
    return _current


class PackPlan:
    def __init__(self):
        self.jobs = {}               # key -> dict(index, w, out, sigma_index, ...)
        self.dirty = False           # jobs added since the tables were built
        self.tables = None           # dtype code -> (jobs_dev, map_dev, n_fwd_blocks, n_all_blocks, max_taps)
        self.generation = 0
        self.packed_tr = False       # whether the last batched launch included the transposed packs
        self.hits = 0                # lookups served from the batched launch (tests)

    @staticmethod
    def key(w, dtype, cin_pad, transposed, plane=False):
        return (w.data_ptr(), tuple(w.shape), tuple(w.stride()), dtype, int(cin_pad), bool(transposed), bool(plane))

    # ---------------------------------------------------------------- learning
    def record(self, w, dtype, cin_pad, transposed, sigma, plane=False):
        k = self.key(w, dtype, cin_pad, transposed, plane)
        if k in self.jobs:
            return
        sidx = -1 if sigma is None else int(sigma.storage_offset())
        from .ops import _cl_dense
        wd = w.detach()
        cl = bool(wd.dtype == torch.float32 and not wd.is_contiguous() and _cl_dense(wd) and int(cin_pad) == wd.shape[1] and wd.shape[1] % 8 == 0)
        if not cl and (wd.dtype != torch.float32 or not wd.is_contiguous()):
            return                                           # (needs a converted copy: packed one by one, ops.pack_weight)
        self.jobs[k] = dict(w=wd, dtype=dtype, cin_pad=int(cin_pad), transposed=bool(transposed), sigma_index=sidx, out=None, cl=cl,
                            plane=bool(plane))
        self.dirty = True

    def lookup(self, w, dtype, cin_pad, transposed, generation=None, plane=False):
        """The pre-packed matrix, or None (not learned yet / packed by an older or newer forward)."""
        if self.tables is None or (generation is not None and generation != self.generation):
            return None
        if transposed and not self.packed_tr:
            return None
        j = self.jobs.get(self.key(w, dtype, cin_pad, transposed, plane))
        if j is None or j.get('stale', True):
            return None
        self.hits += 1
        return j['out']

    # ---------------------------------------------------------------- per forward
    def _build(self):
        lib = L.lib()
        by_dtype = {}
        for k, j in self.jobs.items():
            by_dtype.setdefault(j['dtype'], []).append(j)
        self.tables = {}
        for dtype, jobs in by_dtype.items():
            # A forward pack whose weight ALSO has a transposed pack from a channels-last master (both row-major layouts) is written by
            # the transposed job's blocks when both are wanted (one read of the fp32 master): such "covered" forward jobs come first,
            # a launch with autograd on starts behind them.
            def wkey(j):
                return (j['w'].data_ptr(), tuple(j['w'].shape), tuple(j['w'].stride()), j['cin_pad'])
            tr_of = {wkey(j): j for j in jobs if j['transposed'] and j['cl'] and not j['plane']}
            for j in jobs:
                j['dual'] = None
                j['covered'] = False
            for j in jobs:
                if not j['transposed'] and j['cl'] and not j['plane'] and wkey(j) in tr_of:
                    j['covered'] = True
                    tr_of[wkey(j)]['dual'] = j
            jobs.sort(key=lambda j: (j['transposed'], not j['covered']))     # covered forward packs, the other forward packs, transposed packs
            dt = L.S2E_BF16 if dtype == torch.bfloat16 else L.S2E_F32
            dev = jobs[0]['w'].device
            arr = (L.PackJob * len(jobs))()
            for i, j in enumerate(jobs):
                cout, cin, kh, kw = j['w'].shape
                rows = lib.s2e_conv_cout_pad(j['cin_pad'] if j['transposed'] else cout)
                kpad = lib.s2e_conv_k_pad(dt, kh * kw * (cout if j['transposed'] else j['cin_pad']))
                if j['plane']:                               # the PLANE layout (csrc/conv_plane.h): 64-row groups, no K padding
                    rows = ((j['cin_pad'] if j['transposed'] else cout) + 63) // 64 * 64
                    kpad = kh * kw * (cout if j['transposed'] else j['cin_pad'])
                if j['out'] is None:
                    # (a covered forward pack's padding rows / K tail are written by its own job only: zero once)
                    j['out'] = (torch.zeros if j['covered'] else torch.empty)(rows, kpad, dtype=dtype, device=dev)
                j['stale'] = True
                arr[i].w, arr[i].out, arr[i].sigma_index = j['w'].data_ptr(), j['out'].data_ptr(), j['sigma_index']
                arr[i].cout, arr[i].cin, arr[i].taps, arr[i].cin_pad = cout, cin, kh * kw, j['cin_pad']
                arr[i].transposed = int(j['transposed']) | (2 if j['cl'] else 0) | (4 if j['plane'] else 0)     # bit 1: the source is stored channels-last; bit 2: PLANE layout
            for i, j in enumerate(jobs):
                if j['dual'] is not None:
                    arr[i].out_fwd = j['dual']['out'].data_ptr()
            n_fwd = sum(1 for j in jobs if not j['transposed'])
            n_cov = sum(1 for j in jobs if j['covered'])
            nb_fwd = lib.s2e_pack_block_map(dt, C.byref(arr), n_fwd, None) if n_fwd else 0
            bm = np.zeros(3 * max(nb_fwd, 1), dtype=np.int32)
            if n_fwd:
                lib.s2e_pack_block_map(dt, C.byref(arr), n_fwd, bm.ctypes.data)
            # with autograd on: the jobs behind the covered ones (job indices of that map count from arr[n_cov])
            rest = (L.PackJob * (len(jobs) - n_cov)).from_buffer(arr, n_cov * C.sizeof(L.PackJob)) if len(jobs) > n_cov else None
            nb_all = lib.s2e_pack_block_map(dt, C.byref(rest), len(jobs) - n_cov, None) if rest is not None else 0
            bm_all = np.zeros(3 * max(nb_all, 1), dtype=np.int32)
            if nb_all:
                lib.s2e_pack_block_map(dt, C.byref(rest), len(jobs) - n_cov, bm_all.ctypes.data)
            jobs_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
            map_dev = (torch.from_numpy(bm).to(dev), torch.from_numpy(bm_all).to(dev), n_cov * C.sizeof(L.PackJob))
            max_taps = max(j['w'].shape[2] * j['w'].shape[3] for j in jobs)
            self.tables[dt] = (jobs_dev, map_dev, nb_fwd, nb_all, max_taps, jobs, n_fwd)
        self.dirty = False

    def run(self, sigma_base):
        """Pack everything learned so far (forward packs only when autograd is off)."""
        self.generation += 1
        if self.dirty or (self.tables is None and self.jobs):
            self._build()
        if not self.tables:
            return
        want_tr = torch.is_grad_enabled()
        self.packed_tr = want_tr
        st = torch.cuda.current_stream().cuda_stream
        for dt, (jobs_dev, map_dev, nb_fwd, nb_all, max_taps, jobs, n_fwd) in self.tables.items():
            nb = nb_all if want_tr else nb_fwd
            if nb == 0:
                continue
            map_fwd, map_all, skip = map_dev
            jobs_ptr = jobs_dev.data_ptr() + (skip if want_tr else 0)
            map_ptr = (map_all if want_tr else map_fwd).data_ptr()
            if any(j['sigma_index'] >= 0 for j in jobs) and sigma_base is None:
                raise L.Seg2EyeHipError('PackPlan: spectral-normed weights but no sigma array')
            from .ops import LaunchProfiler
            esz = 2 if dt == L.S2E_BF16 else 4
            # (algorithmic: the fp32 master read, the packed copy written; a covered forward pack costs its write only)
            nbytes = float(sum(j['w'].numel() * ((esz if (want_tr and j['covered']) else 4 + esz)) for i, j in enumerate(jobs) if want_tr or i < n_fwd))
            LaunchProfiler.run('weight_pack', 0.0, lambda: L.check(
                L.lib().s2e_pack_conv_weights(dt, jobs_ptr, map_ptr, nb, max_taps,
                                              None if sigma_base is None else sigma_base.data_ptr(), st),
                's2e_pack_conv_weights'), nbytes=nbytes)                # fp32 master read, packed copy written
            for i, j in enumerate(jobs):
                j['stale'] = not (want_tr or i < n_fwd)


class network_scope:
    """`with network_scope(net, bank):` around a top-level network's forward, after its power iteration."""

    def __init__(self, net, bank):
        plan = net.__dict__.get('_pack_plan')
        if plan is None:
            plan = PackPlan()
            net.__dict__['_pack_plan'] = plan
        self.plan = plan
        self.sigma = None if bank is None else bank.sigma

    def __enter__(self):
        global _current
        self.prev = _current
        self.plan.run(self.sigma)
        _current = self.plan
        return self.plan

    def __exit__(self, *exc):
        global _current
        _current = self.prev
        return False
