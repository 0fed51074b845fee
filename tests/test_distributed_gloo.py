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
    # a step that failed BEFORE anything was launched leaves the sync usable ...
    gsync.reset()
    opt.flat_g.copy_(local)
    assert abs(gsync.all_reduce() - 1.0 / world) < 1e-12 and torch.equal(opt.flat_g, grouped)
    # ... one that failed with collectives in flight POISONS it: the handles are dropped without waiting (a peer may
    # never issue its half: ADVICE r3), so the stale collectives may still write the arena -- the next step must not run through it
    psync = sdist.FlatGradSync(opt.flat_g, bucket_bytes=64, groups=[(cut, n), (0, cut)])
    opt.flat_g.copy_(local)
    psync.launch(0)
    in_flight = list(psync._handles)
    psync.reset()
    assert not psync._handles and not psync._launched and psync._poisoned
    for h in in_flight:                                  # (here both ranks did launch: let the collectives finish before reusing the arena)
        h.wait()
    for call in (lambda: psync.launch(1), psync.all_reduce):
        try:
            call()
            raise AssertionError('a poisoned FlatGradSync must refuse to exchange')
        except RuntimeError as e:
            assert 'collectives in flight' in str(e)
    # the bf16 payload and the spelled-out direct exchange: same sums (to the payload's rounding), replicas bit-identical
    variants = {}
    for payload, algo in (('bf16', 'allreduce'), ('fp32', 'direct'), ('bf16', 'direct')):
        vs = sdist.FlatGradSync(opt.flat_g, bucket_bytes=64, groups=[(cut, n), (0, cut)], payload=payload, algorithm=algo)
        for early in (False, True):                      # everything at the end / group 0 launched early (twice: staging buffers reused)
            opt.flat_g.copy_(local)
            if early:
                vs.launch(0)
            assert abs(vs.all_reduce() - 1.0 / world) < 1e-12 and not vs._pending and not vs._handles
            variants[(payload, algo, early)] = opt.flat_g.clone()
        d = vs.describe()
        assert d['payload'] == payload and d['algorithm'] == algo and d['groups'] == 2
    # S2E_DEBUG_SYNC: a group written after it was declared final is caught; an untouched one passes and is exchanged
    os.environ['S2E_DEBUG_SYNC'] = '1'
    dsync = sdist.FlatGradSync(opt.flat_g, bucket_bytes=64, groups=[(cut, n), (0, cut)])
    del os.environ['S2E_DEBUG_SYNC']
    opt.flat_g.copy_(local)
    dsync.launch(0)
    assert abs(dsync.all_reduce() - 1.0 / world) < 1e-12 and torch.equal(opt.flat_g, grouped)
    opt.flat_g.copy_(local)
    dsync.launch(0)
    opt.flat_g[cut] += 1.0                               # "a later segment" writes the group that was reported final
    try:
        dsync.all_reduce()
        raise AssertionError('S2E_DEBUG_SYNC must catch a write after launch')
    except AssertionError as e:
        assert 'declared final' in str(e)
    out[rank] = (p0, local, grouped, grouped.clone(), variants)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (p_a, l_a, g_a, s_a, var_a), (p_b, l_b, g_b, s_b, var_b) = out[0], out[1]
    assert torch.equal(g_a, s_a) and torch.equal(g_b, s_b)          # grouped / early-launched exchange == plain exchange
    assert torch.equal(p_a, p_b)                                   # broadcast made replicas identical
    assert torch.allclose(s_a, l_a + l_b) and torch.equal(s_a, s_b)  # sum all-reduce, same on both ranks
    assert not torch.equal(l_a, l_b)
    for key, got in var_a.items():
        assert torch.equal(got, var_b[key]), key                    # every replica holds the same bits, whatever travelled
        if key[0] == 'fp32':
            assert torch.allclose(got, s_a, rtol=1e-6, atol=1e-7), key   # direct: the owner adds in rank order (fp32)
        else:
            assert torch.allclose(got, s_a, rtol=2 ** -7, atol=1e-6) and not torch.equal(got, s_a), key    # bf16 payload: 8 bits
    assert torch.equal(var_a[('bf16', 'direct', False)], var_a[('bf16', 'direct', True)])


def test_single_process_is_a_noop():
    from seg2eye_amd import distributed as sdist
    g = torch.ones(10)
    assert sdist.FlatGradSync(g).all_reduce() == 1.0 and sdist.world_size() == 1 and sdist.get_rank() == 0
