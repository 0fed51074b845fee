"""The N>1 path on CPU: world_size-2 gloo processes run the flat-arena gradient exchange and must end
with identical, correctly averaged updates (the same code path RCCL takes on the GPUs)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from seg2eye_amd import distributed as sdist
    from seg2eye_amd.optim import FlatAdam
    r, w, _ = sdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and sdist.world_size() == world and sdist.get_rank() == rank
    assert sdist.shard_seed(1234) == 1234 + rank
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT on purpose
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    opt = FlatAdam(list(net.parameters()), lr=1e-2, betas=(0.0, 0.9))
    sdist.broadcast_flat(opt.flat_p)                     # ...and are made identical here
    p0 = opt.flat_p.clone()
    sync = sdist.FlatGradSync(opt.flat_g, bucket_bytes=64)         # tiny buckets -> several all-reduces
    assert len(sync.buckets) > 1
    x = torch.full((4, 6), float(rank + 1))
    opt.zero_grad()
    net(x).sum().backward()
    local = opt.flat_g.clone()
    scale = sync.all_reduce()
    assert abs(scale - 1.0 / world) < 1e-12
    # grouped exchange (the overlap path): groups in "completion order", launched early and out of arena order
    n = opt.flat_g.numel()
    cut = opt.offsets[2]
    gsync = sdist.FlatGradSync(opt.flat_g, bucket_bytes=64, groups=[(cut, n), (0, cut)])
    opt.flat_g.copy_(local)
    gsync.launch(0)                                      # the second Linear's slice is final first
    gsync.launch(0)                                      # (idempotent)
    assert abs(gsync.all_reduce() - 1.0 / world) < 1e-12 and not gsync._handles and not gsync._launched
    grouped = opt.flat_g.clone()
    try:
        sdist.FlatGradSync(opt.flat_g, groups=[(0, cut - 1), (cut, n)])
        raise AssertionError('a gap in the groups must be refused')
    except ValueError:
        pass
    # rank-local passes (rank 0's validation in train.py): ops must see ONE replica there and issue no collective
    assert sdist.sync_world_size() == world
    with sdist.solo():
        assert sdist.sync_world_size() == 1 and sdist.world_size() == world
        with sdist.solo():
            assert sdist.sync_world_size() == 1
        assert sdist.sync_world_size() == 1
    assert sdist.sync_world_size() == world
    # a step that failed after its first group was launched must not leave the sync thinking that group is in flight
    opt.flat_g.copy_(local)
    gsync.launch(0)
    in_flight = list(gsync._handles)
    gsync.reset()                                        # drops the handles WITHOUT waiting (a peer may never issue its half: ADVICE r3)
    assert not gsync._handles and not gsync._launched
    for h in in_flight:                                  # (here both ranks did launch: let the collectives finish before reusing the arena)
        h.wait()
    opt.flat_g.copy_(local)
    assert abs(gsync.all_reduce() - 1.0 / world) < 1e-12 and torch.equal(opt.flat_g, grouped)
    out[rank] = (p0, local, grouped, opt.flat_g.clone())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (p_a, l_a, g_a, s_a), (p_b, l_b, g_b, s_b) = out[0], out[1]
    assert torch.equal(g_a, s_a) and torch.equal(g_b, s_b)          # grouped / early-launched exchange == plain exchange
    assert torch.equal(p_a, p_b)                                   # broadcast made replicas identical
    assert torch.allclose(s_a, l_a + l_b) and torch.equal(s_a, s_b)  # sum all-reduce, same on both ranks
    assert not torch.equal(l_a, l_b)


def test_single_process_is_a_noop():
    from seg2eye_amd import distributed as sdist
    g = torch.ones(10)
    assert sdist.FlatGradSync(g).all_reduce() == 1.0 and sdist.world_size() == 1 and sdist.get_rank() == 0
