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
    become ready).  Without groups, or on one process, `all_reduce()` is the whole exchange."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20, group=None, groups=None):
        self.flat = flat_grad
        self.group = group
        self.per = max(1, bucket_bytes // flat_grad.element_size())
        n = flat_grad.numel()
        self.groups = [(int(a), int(b)) for a, b in groups] if groups else [(0, n)]
        covered = sorted(self.groups)
        if covered[0][0] != 0 or covered[-1][1] != n or any(a[1] != b[0] for a, b in zip(covered[:-1], covered[1:])):
            raise ValueError('FlatGradSync: groups must tile the arena exactly: %s vs %d elements' % (covered, n))
        self.buckets = [(s, min(b, s + self.per)) for a, b in self.groups for s in range(a, b, self.per)]
        self._launched, self._handles = set(), []

    def _start(self, i):
        a, b = self.groups[i]
        for s in range(a, b, self.per):
            self._handles.append(dist.all_reduce(self.flat[s:min(b, s + self.per)], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched.add(i)

    def launch(self, i):
        """Group i's gradients are final: start their exchange now (no-op on one process / when already started)."""
        if exchange_active() and i not in self._launched:
            self._start(i)

    def reset(self):
        """Forget launches whose step did not complete (a backward that raised after `launch(0)`), so that the next step starts
        its own exchange instead of treating the groups as already launched.  The handles are DROPPED, not waited for: when the
        failure is local to this rank the peers never issue the matching collective, and a wait here would hold the real
        exception back until the process group's timeout.  A job whose ranks have diverged is over either way; the
        caller re-raises."""
        self._launched, self._handles = set(), []

    def all_reduce(self):
        if not exchange_active():
            return 1.0
        for i in range(len(self.groups)):
            if i not in self._launched:
                self._start(i)
        for h in self._handles:
            h.wait()
        self._launched, self._handles = set(), []
        return 1.0 / world_size()


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
