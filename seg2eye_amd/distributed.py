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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # S2E_DIST_BACKEND=gloo: the multi-rank control flow on a box with fewer GPUs than ranks (tests); RCCL
            # itself refuses two ranks on one device
            backend = os.environ.get('S2E_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class FlatGradSync:
    """Sum-all-reduce a flat gradient arena in `bucket_bytes` slices.  The division by world_size is
    folded into the Adam kernel (grad_scale), so the exchange is a pure sum."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20, group=None):
        self.flat = flat_grad
        self.group = group
        per = max(1, bucket_bytes // flat_grad.element_size())
        n = flat_grad.numel()
        self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]

    def all_reduce(self):
        if world_size() == 1:
            return 1.0
        handles = [dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                   for a, b in self.buckets]
        for h in handles:
            h.wait()
        return 1.0 / world_size()


def broadcast_flat(flat, src=0):
    if world_size() > 1:
        dist.broadcast(flat, src=src)


def shard_seed(base_seed):
    """Each rank draws its own synthetic shard (seeds data 1234 + rank)."""
    return base_seed + get_rank()
