"""Data-parallel replication: one process per GPU, gradients exchanged with RCCL all-reduce over xGMI
(torch.distributed backend 'nccl' is RCCL on ROCm).  New relative to the reference, which is
single-GPU (README.md:56-58).

The path shards over the batch with ONE exchange per optimizer step: InstanceNorm is
per-sample, losses are batch means over equal shards, spectral-norm power iteration and Adam are
weight-only, so replicas stay identical given identical summed gradients.  Gradients already live in
one flat fp32 arena per optimizer (optim.FlatAdam), so a bucket is just a slice: no pack/unpack
copies.  Buckets are sized for xGMI's per-link bandwidth (7 links x ~153 GB/s, point to point): a few
large messages let RCCL's direct algorithms use all links at once."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if a launcher set them.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or _single_rank_collectives()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # S2E_DIST_BACKEND=gloo: the multi-rank control flow on a box with fewer GPUs than ranks (tests); RCCL
            # itself refuses two ranks on one device
            backend = os.environ.get('S2E_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        # generous timeout: rank 0 alone runs the periodic validation passes (train.py) while the others wait in the next
        # collective; a full validation over the OpenEDS set takes longer than the default 10 minutes
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(hours=3))
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _single_rank_collectives():
    """S2E_DIST_SINGLE=1: run every collective of the data-parallel path even in a process group of ONE rank.  It moves no
    data between GPUs but executes every call an 8-GPU run makes -- RCCL initialisation, asynchronous all-reduces launched
    from the backward hooks on RCCL's stream, the broadcasts -- on a box with a single GPU (tests/test_networks_gpu.py)."""
    return os.environ.get('S2E_DIST_SINGLE', '0') == '1'


def exchange_active():
    """Do the data-parallel collectives run?  More than one replica -- or one, when S2E_DIST_SINGLE asks for it."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _single_rank_collectives())


_solo_depth = 0


class solo:
    """`with solo():` -- the enclosed passes run on THIS rank only (rank 0's validation passes in train.py, the Tester):
    ops that exchange data between replicas as part of a forward or backward (BatchNorm SPADE's batch statistics) must not
    issue a collective there, because the other ranks are not in the same code path -- they are waiting in the next
    broadcast -- and a solo all-reduce would pair up with it (hang, or mixed payloads).  Inside, `sync_world_size()` is 1."""

    def __enter__(self):
        global _solo_depth
        _solo_depth += 1
        return self

    def __exit__(self, *exc):
        global _solo_depth
        _solo_depth -= 1
        return False


def sync_world_size():
    """The number of replicas a forward / backward op exchanges data with: world_size(), or 1 inside `solo()`."""
    return 1 if _solo_depth > 0 else world_size()


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class FlatGradSync:
    """Sum-all-reduce a flat gradient arena in `bucket_bytes` slices.  The division by world_size is
    folded into the Adam kernel (grad_scale), so the exchange is a pure sum.

    groups: optional list of (start, end) element ranges of the arena in the order their gradients become FINAL during the
    backward pass (Pix2PixModel lays the generator's parameters out so: the late blocks' range first).  `launch(i)` starts
    the all-reduces of group i right away -- asynchronously: the collective is ordered after the kernels already enqueued on
    the current stream and runs on the backend's own stream while the rest of the backward keeps the compute stream busy --
    and `all_reduce()` starts whatever was not launched yet and waits for everything (buckets launched as they
    become ready).  Without groups, or on one process, `all_reduce()` is the whole exchange.

    payload ('fp32' | 'bf16', --grad_dtype): what travels.  bf16 halves the bytes on the links (G + E: 198 instead of 396 MB per
        step): a bucket is rounded to bf16 into a staging buffer, summed across the replicas in bf16, and written back to the
        fp32 arena.  Every replica receives the same bits, so the replicas stay bit-identical (tested); the gradient itself
        carries bf16's 8 significant bits, like every activation gradient of the bf16 step already does.
    algorithm ('allreduce' | 'direct', --grad_exchange): 'allreduce' hands each bucket to the backend's all-reduce (RCCL chooses
        ring / tree and the protocol; NCCL_ALGO / NCCL_PROTO select them explicitly and are recorded in the bench line);
        'direct' spells the exchange out for the fully connected xGMI mesh of one node (a ring pushes 2 (P-1)/P S
        over ONE link, point-to-point pieces use all seven): all-to-all of the P bucket shards, each replica adds the P copies
        of ITS shard in rank order, all-gather of the sums -- one owner per element, so bit-identical replicas here too.
    noop (attribute, bench.py's `exchange_ms_exposed`): launch / all_reduce start and wait for NOTHING and still return 1 / world --
        the same step without its collectives; the replicas' gradients then differ, so only for timing.
    S2E_DEBUG_SYNC=1: `launch(i)` only records a copy of the group's slice; `all_reduce()` checks that the slice still holds those
        bits when the backward has ended -- i.e. that nothing wrote a group after it was declared final -- and then exchanges."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20, group=None, groups=None, payload='fp32', algorithm='allreduce'):
        if payload not in ('fp32', 'bf16') or algorithm not in ('allreduce', 'direct'):
            raise ValueError('FlatGradSync: payload %r / algorithm %r' % (payload, algorithm))
        self.flat = flat_grad
        self.group = group
        self.payload, self.algorithm = payload, algorithm
        self.per = max(1, bucket_bytes // flat_grad.element_size())
        n = flat_grad.numel()
        self.groups = [(int(a), int(b)) for a, b in groups] if groups else [(0, n)]
        covered = sorted(self.groups)
        if covered[0][0] != 0 or covered[-1][1] != n or any(a[1] != b[0] for a, b in zip(covered[:-1], covered[1:])):
            raise ValueError('FlatGradSync: groups must tile the arena exactly: %s vs %d elements' % (covered, n))
        self.buckets = [(s, min(b, s + self.per)) for a, b in self.groups for s in range(a, b, self.per)]
        self._launched, self._handles = set(), []
        self._pending = []                                   # second halves of buckets in flight: callables run by all_reduce()
        self._stage = {}                                     # bucket start -> staging buffers (kept: no allocation per step)
        self._poisoned = False
        self.noop = False
        self._debug = os.environ.get('S2E_DEBUG_SYNC', '0') == '1'
        self._final = {}                                     # S2E_DEBUG_SYNC: group -> copy of its slice when it was declared final

    # ---- one bucket ---------------------------------------------------------------------------------
    def _buffers(self, s, e):
        st = self._stage.get(s)
        if st is None:
            world = dist.get_world_size(self.group)
            dt = torch.bfloat16 if self.payload == 'bf16' else self.flat.dtype
            n = e - s
            st = {}
            if self.algorithm == 'direct':
                shard = ((n + world - 1) // world + 7) // 8 * 8                                  # (16-byte multiples: the owner-sum kernel's vectors)
                st['send'] = torch.zeros(shard * world, dtype=dt, device=self.flat.device)       # (zero tail: padding adds nothing)
                st['recv'] = torch.empty(shard * world, dtype=dt, device=self.flat.device)
                st['sum'] = torch.empty(shard, dtype=dt, device=self.flat.device)
                st['shard'] = shard
            elif self.payload == 'bf16':
                st['send'] = torch.empty(n, dtype=dt, device=self.flat.device)
            self._stage[s] = st
        return st

    def _start_bucket(self, s, e):
        sl = self.flat[s:e]
        st = self._buffers(s, e)
        if self.algorithm == 'allreduce':
            if self.payload == 'fp32':
                self._handles.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                return
            st['send'].copy_(sl)                             # round to bf16 on the compute stream, behind the kernels that wrote sl
            self._handles.append(dist.all_reduce(st['send'], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._pending.append(lambda: (sl.copy_(st['send']), None)[1])
            return
        # direct: shard r of every replica's bucket goes to replica r ...
        n, shard = e - s, st['shard']
        st['send'][:n].copy_(sl)
        h = dist.all_to_all_single(st['recv'], st['send'], group=self.group, async_op=True)

        def finish():
            h.wait()                                         # (device-side dependency on RCCL's stream; a host wait under gloo)
            world = dist.get_world_size(self.group)
            # ... which adds the P copies in rank order (fp32 accumulation, one rounding for a bf16 payload): one launch on the GPU ...
            if st['recv'].is_cuda:
                from . import _lib as L
                L.check(L.lib().s2e_shard_sum(L.S2E_BF16 if self.payload == 'bf16' else L.S2E_F32, st['recv'].data_ptr(), st['sum'].data_ptr(),
                                              world, shard, torch.cuda.current_stream().cuda_stream), 's2e_shard_sum')
            else:                                            # (CPU arenas: the gloo dry runs of tests/test_distributed_gloo.py)
                acc = st['recv'].view(world, shard).to(torch.float32).sum(dim=0) if self.payload == 'bf16' else \
                    st['recv'].view(world, shard).sum(dim=0)
                st['sum'].copy_(acc)
            # ... and hands the sums to everybody: asynchronously, every bucket's gather is in flight before the first is
            # waited for; the copy back into the arena is the returned closure
            h2 = dist.all_gather_into_tensor(st['send'], st['sum'], group=self.group, async_op=True)

            def done():
                h2.wait()
                sl.copy_(st['send'][:n])
            return done
        self._pending.append(finish)

    def _start(self, i):
        a, b = self.groups[i]
        if self._debug:
            self._final[i] = self.flat[a:b].clone()
            self._launched.add(i)
            return
        for s in range(a, b, self.per):
            self._start_bucket(s, min(b, s + self.per))
        self._launched.add(i)

    def _check_poison(self):
        if self._poisoned:
            raise RuntimeError('FlatGradSync: the previous step was aborted with collectives in flight (reset() dropped their '
                               'handles): they may still write this gradient arena and the backend\'s stream holds them in front '
                               'of every later collective.  Tear the process group down and restart the job.')

    def launch(self, i):
        """Group i's gradients are final: start their exchange now (no-op on one process / when already started)."""
        if self.noop:
            return
        if exchange_active() and i not in self._launched:
            self._check_poison()
            self._start(i)

    def reset(self):
        """Forget launches whose step did not complete (a backward that raised after `launch(0)`).  The handles are DROPPED, not
        waited for: when the failure is local to this rank the peers never issue the matching collective, and a wait here would
        hold the real exception back until the process group's timeout.  If any collective was in flight the sync is
        POISONED: the stale collectives still write the arena on the backend's stream and sit in front of every later
        one, so a caller that survives the exception must not run another step through it -- `launch` / `all_reduce` raise.
        Nothing in flight (the failure came before the first group completed): the object stays usable."""
        if self._handles or self._pending:
            self._poisoned = True
        self._launched, self._handles, self._pending, self._final = set(), [], [], {}

    def all_reduce(self):
        if not exchange_active():
            return 1.0
        if self.noop:
            return 1.0 / world_size()
        self._check_poison()
        if self._debug:
            stale = [i for i, snap in self._final.items() if not torch.equal(self.flat[self.groups[i][0]:self.groups[i][1]], snap)]
            self._final, self._launched = {}, set()
            if stale:
                raise AssertionError('FlatGradSync (S2E_DEBUG_SYNC): gradient group(s) %s were written after they had been '
                                     'declared final -- an early all-reduce would have exchanged stale values' % stale)
            self._debug = False
            try:
                return self.all_reduce()
            finally:
                self._debug = True
        for i in range(len(self.groups)):
            if i not in self._launched:
                self._start(i)
        for h in self._handles:
            h.wait()
        tails = [fin() for fin in self._pending]             # (second halves: a 'direct' bucket returns the closure that ends it)
        for t in tails:
            if t is not None:
                t()
        self._launched, self._handles, self._pending = set(), [], []
        return 1.0 / world_size()

    def time_groups(self):
        """bench.py (--gpus N): the exchange of each group on its own -- started on an idle device, waited for, timed by the host
        around a device synchronize -- in milliseconds.  Collective: every rank calls it.  The arena's contents are summed (garbage
        in, garbage out: call it after the timed region)."""
        import time
        out = []
        if not exchange_active():
            return out
        for i in range(len(self.groups)):
            torch.cuda.synchronize() if self.flat.is_cuda else None
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            self._start(i)
            for h in self._handles:
                h.wait()
            for t in [fin() for fin in self._pending]:
                if t is not None:
                    t()
            torch.cuda.synchronize() if self.flat.is_cuda else None
            out.append((time.perf_counter() - t0) * 1e3)
            self._launched, self._handles, self._pending = set(), [], []
        return out

    def describe(self):
        """What the bench line records about the exchange."""
        return {'payload': self.payload, 'algorithm': self.algorithm, 'bucket_MB': self.per * self.flat.element_size() / 2 ** 20,
                'groups': len(self.groups), 'NCCL_ALGO': os.environ.get('NCCL_ALGO', 'backend default'),
                'NCCL_PROTO': os.environ.get('NCCL_PROTO', 'backend default')}


def broadcast_flat(flat, src=0):
    if exchange_active():
        dist.broadcast(flat, src=src)


def all_reduce_sum_(t, group=None):
    """In-place sum over the replicas (no-op on one process); returns t."""
    if exchange_active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def replica_buffers(nets):
    """The per-network state that is NOT in the optimizer arenas and must be the same on every replica: each spectral-norm
    bank's u|v arena (the power iteration is deterministic -- integer atomics -- so equal weights keep them equal, but they
    are drawn randomly at construction) and BatchNorm SPADE's running_mean / running_var / num_batches_tracked."""
    from .spectral import ensure_bank
    out = []
    for net in nets:
        if net is None:
            continue
        bank = ensure_bank(net)
        if bank is not None and bank.n:
            bank.ensure_built()
            out.append(bank.uv_arena)
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats:
                out += [m.running_mean, m.running_var, m.num_batches_tracked]
    return out


def broadcast_buffers(nets, src=0):
    """Make the replicas' non-parameter state rank `src`'s (at start-up, and after rank 0 alone ran a train-mode validation
    pass, which advances its u, v and BatchNorm statistics like the reference's does, util/tester.py + SURVEY F7)."""
    if not exchange_active():
        return
    for t in replica_buffers(nets):
        dist.broadcast(t, src=src)


def shard_seed(base_seed):
    """Each rank draws its own synthetic shard (seeds data 1234 + rank)."""
    return base_seed + get_rank()
