#!/usr/bin/env python3
"""Data-parallel consistency check, run under torch.distributed.run (tests/test_networks_gpu.py launches it with two ranks
sharing one GPU over gloo; on a multi-GPU node it runs over RCCL as is).

  --mode train : every rank builds a Pix2PixTrainer from DIFFERENT initial weights / spectral-norm vectors (seed + rank; the
                 start-up broadcasts must make them rank 0's), runs --iters G+D iterations on its shard [rank*b, (rank+1)*b) of
                 one global batch, and rank 0 prints one JSON line: per-iteration losses (mean over ranks), whether the parameter
                 arenas and the u|v arenas are BIT-identical on all ranks, and checksums of rank 0's arenas.
  --mode bn    : BatchNorm-SPADE generator (--norm_G spectralspadebatch3x3): forward + backward on the shard with statistics
                 synchronised over the replicas; prints the output checksum per rank, the summed weight-gradient checksum and the
                 running buffers, to be compared with one process running the whole batch."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def fill(net, seed):
    from seg2eye_amd import synthetic as syn
    sd = syn.fill_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(torch.from_numpy(sd[k]))


def shard(batch, rank, b):
    return {k: (v[rank * b:(rank + 1) * b] if torch.is_tensor(v) else v) for k, v in batch.items()}


def global_batch(n, hw, seed):
    from seg2eye_amd import synthetic as syn
    b = syn.make_batch(n, hw, hw, seed=seed)
    return {'label': torch.from_numpy(b['label']), 'style_image': torch.from_numpy(b['style_image']), 'target': torch.from_numpy(b['target'])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='train')
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--batch', type=int, default=2, help='per rank')
    ap.add_argument('--ngf', type=int, default=8)
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--norm_G', default='spectralspadeinstance3x3')
    args = ap.parse_args()
    import torch.distributed as dist
    from seg2eye_amd import distributed as sdist
    from seg2eye_amd.options import default_opt
    rank, world, local = sdist.init_from_env()
    dev_index = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    opt = default_opt(ngf=args.ngf, ndf=args.ngf, crop_size=256, aspect_ratio=1.0, batchSize=args.batch, compute_dtype=args.dtype,
                      gpu_ids=[dev_index], norm_G=args.norm_G, hip_graphs=False)
    import contextlib, io
    gb = global_batch(args.batch * world, 256, 77)
    data = shard(gb, rank, args.batch)

    def gather(obj):
        out = [None] * world
        if world > 1:
            dist.all_gather_object(out, obj)
        else:
            out = [obj]
        return out

    if args.mode == 'train':
        from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
        from seg2eye_amd.spectral import ensure_bank
        # weights are filled BEFORE the trainer's start-up broadcast can see them only if that broadcast runs after; the
        # trainer broadcasts in its constructor, so: construct, fill with rank-dependent values, broadcast again explicitly
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Pix2PixTrainer(opt)
        m = tr.pix2pix_model
        for net, s in ((m.netG, 11), (m.netD, 12), (m.netE, 13)):
            fill(net, s + 100 * rank)
        sdist.broadcast_flat(tr.optimizer_G.flat_p)
        sdist.broadcast_flat(tr.optimizer_D.flat_p)
        tr.sync_replica_buffers()
        losses = []
        for _ in range(args.iters):
            tr.run_generator_one_step(dict(data))
            tr.run_discriminator_one_step(dict(data))
            losses.append({k: float(v.detach().float().mean()) for k, v in tr.get_latest_losses().items()})
        torch.cuda.synchronize()
        uv = torch.cat([ensure_bank(n).uv_arena for n in (m.netG, m.netD, m.netE)])
        mine = {'G': digest(tr.optimizer_G.flat_p), 'D': digest(tr.optimizer_D.flat_p), 'uv': digest(uv), 'losses': losses}
        allr = gather(mine)
        if rank == 0:
            mean_losses = [{k: float(np.mean([r['losses'][i][k] for r in allr])) for k in losses[i]} for i in range(args.iters)]
            print(json.dumps({'world': world, 'identical_G': len({r['G'] for r in allr}) == 1, 'identical_D': len({r['D'] for r in allr}) == 1,
                              'identical_uv': len({r['uv'] for r in allr}) == 1, 'losses': mean_losses,
                              'G_sum': float(tr.optimizer_G.flat_p.double().sum()), 'G_abs': float(tr.optimizer_G.flat_p.double().abs().sum()),
                              'D_sum': float(tr.optimizer_D.flat_p.double().sum()), 'D_abs': float(tr.optimizer_D.flat_p.double().abs().sum())}), flush=True)
    else:
        from seg2eye_amd import networks, synthetic as syn
        with contextlib.redirect_stdout(io.StringIO()):
            G = networks.define_G(opt)
        fill(G, 31)
        G.train()
        n = args.batch * world
        w_all = torch.from_numpy(syn.hash_normal('latent_w', (n, 16), seed=31)).cuda()
        proj_all = torch.from_numpy(syn.hash_uniform('g_proj', (n, 1, 256, 256), seed=9)).cuda()
        sl = slice(rank * args.batch, (rank + 1) * args.batch)
        y = G(data['label'].cuda(), w_all[sl])
        (y.float() * proj_all[sl]).sum().backward()
        gsum = torch.cat([p.grad.detach().float().flatten() for p in G.parameters()])
        sdist.all_reduce_sum_(gsum)                                   # sum over the replicas = gradient of the whole-batch loss
        bn = [mod for mod in G.modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        mine = {'y_sum': float(y.double().sum()), 'y_abs': float(y.double().abs().sum()), 'rm': digest(torch.cat([b.running_mean for b in bn])),
                'rv': digest(torch.cat([b.running_var for b in bn]))}
        allr = gather(mine)
        if rank == 0:
            print(json.dumps({'world': world, 'y_sum': [r['y_sum'] for r in allr], 'y_abs': [r['y_abs'] for r in allr],
                              'identical_running': len({r['rm'] for r in allr}) == 1 and len({r['rv'] for r in allr}) == 1,
                              'g_sum': float(gsum.double().sum()), 'g_abs': float(gsum.double().abs().sum()),
                              'rm_abs': float(torch.cat([b.running_mean for b in bn]).double().abs().sum()),
                              'rv_sum': float(torch.cat([b.running_var for b in bn]).double().sum())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
