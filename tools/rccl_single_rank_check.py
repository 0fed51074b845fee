#!/usr/bin/env python3
"""RCCL itself on a one-GPU box: a process group of ONE rank with backend 'nccl' (= RCCL on ROCm) and S2E_DIST_SINGLE=1, so
that every collective of the data-parallel path is really issued -- the start-up broadcasts of the parameter arenas and of the
spectral-norm / BatchNorm buffers, the asynchronous all-reduces of the generator's gradient groups launched from the backward
hooks (on RCCL's stream, while the backward continues), the wait before Adam.  No byte crosses a link, but every call the
8-GPU run makes is executed.  Three trainers from the same weights run the same G+D iterations: the overlapped exchange with
eager launches (hooks start each group's all-reduce), the overlapped exchange with hipGraphs (the G step replays as one graph
SEGMENT per gradient group, group k's all-reduce is started between segments k and k+1), and --no_overlap_allreduce (one graph,
one exchange after the backward); prints one JSON line with the largest parameter differences and the losses.

    RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 S2E_DIST_SINGLE=1 python tools/rccl_single_rank_check.py
"""
import argparse
import contextlib
import io
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--ngf', type=int, default=16)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--dtype', default='fp32')
    args = ap.parse_args()
    os.environ.setdefault('S2E_DIST_SINGLE', '1')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
    import torch.distributed as dist
    import bench
    from seg2eye_amd import distributed as sdist
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
    rank, world, local = sdist.init_from_env(backend='nccl')
    assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1 and sdist.exchange_active()
    dev = torch.device('cuda', 0)
    data = bench.make_data(args.batch, 256, 1234, dev)
    calls = {'all_reduce': 0, 'broadcast': 0}
    real_ar, real_bc = dist.all_reduce, dist.broadcast

    def counting_all_reduce(*a, **k):
        calls['all_reduce'] += 1
        return real_ar(*a, **k)

    def counting_broadcast(*a, **k):
        calls['broadcast'] += 1
        return real_bc(*a, **k)
    dist.all_reduce, dist.broadcast = counting_all_reduce, counting_broadcast
    res = {}
    for tag, kw in (('overlap', dict(hip_graphs=False)), ('overlap_graphs', dict(hip_graphs=True)),
                    ('after_backward', dict(hip_graphs=True, no_overlap_allreduce=True))):
        opt = default_opt(ngf=args.ngf, ndf=args.ngf, crop_size=256, aspect_ratio=1.0, batchSize=args.batch, compute_dtype=args.dtype,
                          gpu_ids=[0], **kw)
        before = dict(calls)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Pix2PixTrainer(opt)
        bench.fill_weights(tr.pix2pix_model)
        hooked = 'grad_ready' in tr.pix2pix_model.netG.__dict__
        launched_early = []
        if hooked:
            real_launch = tr.sync_G.launch

            def launch(i, _real=real_launch):
                launched_early.append(i)
                return _real(i)
            tr.sync_G.launch = launch
        for _ in range(args.iters):
            tr.run_generator_one_step(dict(data))
            tr.run_discriminator_one_step(dict(data))
        torch.cuda.synchronize()
        res[tag] = dict(G=tr.optimizer_G.flat_p.detach().clone(), D=tr.optimizer_D.flat_p.detach().clone(),
                        losses={k: float(v.detach().float().mean()) for k, v in tr.get_latest_losses().items()},
                        hooked=hooked, early_launches=len(launched_early), graphs=bool(tr.use_graphs and tr.graph_G is not None),
                        segments=(len(tr.graph_G.segments) if tr.use_graphs and tr.graph_G is not None else 0),
                        all_reduce_calls=calls['all_reduce'] - before['all_reduce'], broadcast_calls=calls['broadcast'] - before['broadcast'])
    a, b, c = res['overlap'], res['after_backward'], res['overlap_graphs']
    strip = lambda r: {k: v for k, v in r.items() if k not in ('G', 'D')}
    out = {'backend': dist.get_backend(), 'world': world, 'iters': args.iters,
           'max_abs_diff_G': float((a['G'] - b['G']).abs().max()), 'max_abs_diff_D': float((a['D'] - b['D']).abs().max()),
           'max_abs_diff_G_segmented': float((c['G'] - b['G']).abs().max()), 'max_abs_diff_D_segmented': float((c['D'] - b['D']).abs().max()),
           'finite': bool(torch.isfinite(a['G']).all() and torch.isfinite(b['G']).all() and torch.isfinite(c['G']).all()),
           'overlap': strip(a), 'overlap_graphs': strip(c), 'after_backward': strip(b)}
    print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
