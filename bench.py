#!/usr/bin/env python3
"""Headline benchmark: images/s for one G+D train step (run_generator_one_step +
run_discriminator_one_step, optimizer steps included) at 256x256, batch 8 per GPU, bf16 MFMA /
fp32 accumulate, synthetic 4-class label maps + styles (BASELINE.json metric; configs[2]).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Besides the driver contract it carries
  roofline     : the dominant kernel (by GPU time: the patch-resident 3x3 conv kernel, MFMA-bound): algorithmic FLOPs
                 of its launches / their HIP-event durations, measured live; `kernels` lists every conv kernel the same way;
  roofline_hbm : the dominant HBM-bound entry point the same way, against 8 TB/s;
  cpu_baseline : the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded
                 sample (G+D steps at batch 8 of the same 256x256 ngf=64 workload), rank 0, N=1 only;
  dense_labels : the same timed steps on iid random label maps (no label-uniform rectangle: the label-sparse SPADE
                 launches skip nothing), `eager` the same steps as individual launches instead of hipGraph replays (the mode
                 the overlapped multi-GPU exchange runs in), `secondary` SURVEY 8(d)'s secondary metric: G-only forward
                 images/s of config 2 (fp32, eval mode) with its fraction of the fp32 MFMA peak, and the bf16 inference
                 rate (netE + netG + post-processing).  Rank 0, N=1 only; none of them is inside the timed region.
"""
import argparse
import json
import os
import sys
import subprocess
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16 = 2500.0     # TFLOP/s dense (MI355X_MICROARCH.md)
MFMA_PEAK_F32 = 157.3
HBM_PEAK_GBS = 8000.0        # HBM3E (MI355X_MICROARCH.md)


def make_data(n, hw, seed, dev):
    from seg2eye_amd import synthetic as syn
    b = syn.make_batch(n, hw, hw, seed=seed)
    return {'label': torch.from_numpy(b['label']).to(dev), 'style_image': torch.from_numpy(b['style_image']).to(dev),
            'target': torch.from_numpy(b['target']).to(dev), 'filename': b['filename']}


def fill_weights(model):
    from seg2eye_amd import synthetic as syn
    for net in (model.netG, model.netD, model.netE):
        sd = net.state_dict()
        filled = syn.fill_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed=0, settle=False)
        with torch.no_grad():
            for k, v in sd.items():
                v.copy_(torch.from_numpy(filled[k]))


def _physical_cores():
    """Physical cores of the host (`n = all physical cores`): distinct (socket, core) pairs of /proc/cpuinfo,
    falling back to os.cpu_count()."""
    try:
        seen, phys, core = set(), None, None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_step_seconds(opt_kwargs, hw, batch, threads, warm, timed, budget_s):
    """Seconds per G+D step of the CPU oracle (oracle/seg2eye_oracle.py, the pinned restatement of the reference's
    Pix2PixTrainer step) at this batch and thread count: `warm` untimed + up to `timed` timed iterations, stopping early
    once `budget_s` of wall time is spent.  -> (median seconds, iterations timed, warm-ups done)"""
    from oracle import seg2eye_oracle as O
    from seg2eye_amd import networks, synthetic as syn
    from seg2eye_amd.options import default_opt, latent_size
    opt = default_opt(**{**opt_kwargs, 'gpu_ids': [], 'batchSize': batch})
    sds = []
    for cls in (networks.SPADESTYLEGenerator, networks.MultiscaleDiscriminator, networks.ConvEncoder):
        net = cls(opt)
        man = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        sds.append({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(man, seed=0, settle=False).items()})
    sw, sh = latent_size(opt)
    m = O.OracleModel(sds[0], sds[1], sds[2], opt, sh, sw)
    b = syn.make_batch(batch, hw, hw, seed=1234)
    data = {'label': torch.from_numpy(b['label'].astype(np.int64)), 'style_image': torch.from_numpy(b['style_image']),
            'target': torch.from_numpy(b['target'])}
    torch.set_num_threads(threads)
    t_start, times, warmed = time.time(), [], 0
    for it in range(warm + timed):
        t0 = time.time()
        m.run_generator_one_step(data)
        m.run_discriminator_one_step(data)
        dt = time.time() - t0
        if it < warm:
            warmed += 1
        else:
            times.append(dt)
        if time.time() - t_start > budget_s and (times or it + 1 >= warm + timed):
            break
    if not times:                                   # the budget ran out inside the warm-up: report that iteration, say so
        times, warmed = [dt], warmed - 1
    return float(np.median(times)), len(times), warmed


def cpu_baseline(opt_kwargs, hw, batch, budget_s=100.0, extras=False):
    """SURVEY 8(d) / BASELINE.md 4: the CPU restatement runs the identical G+D step (same shapes -- batch 8 -- fp32, same synthetic
    inputs) on this box's host cores: 1 warm-up + 3 timed iterations (median; ~18 s each on the GPU box), and the 8-thread figure
    SURVEY 6's survey numbers were taken at (one un-warmed iteration) -- both in the DEFAULT run since round 5 (VERDICT r4 #7:
    the 30-s cap of round 4 left a single timed iteration).  Thread count: measured on the GPU box's 2 x 64-core EPYC 9575F
    (profiles/r02/cpu_baseline_sweep.txt: 54 / 28 / 19.0 / 18.5 / 21 s per step at 128 / 64 / 32 / 16 / 8 threads) the step is
    fastest at 16-32 threads and 3x slower on all 128 physical cores (oneDNN across two sockets), so `cores` =
    min(physical, 32) -- the host's best, stated.  extras (--cpu-baseline-extras): also all physical cores, one un-warmed
    iteration (~55 s more).  budget_s caps the main leg's wall time on a slow host (fewer timed iterations, stated)."""
    phys = _physical_cores()
    cores = max(1, min(phys, 32))
    sec, n, warmed = cpu_step_seconds(opt_kwargs, hw, batch, cores, 1, 3, budget_s)
    out = {'value': batch / sec, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
           'sample': 'G+D train step of oracle/seg2eye_oracle.py (torch %s CPU fp32) at batch %d, %dx%d, ngf=ndf=%d: %d warm-up + '
                     'median of %d timed iteration(s), %.1f s per step, %d threads of %d physical cores'
                     % (torch.__version__, batch, hw, hw, opt_kwargs['ngf'], warmed, n, sec, cores, phys)}
    if cores > 8:
        sec8, n8, w8 = cpu_step_seconds(opt_kwargs, hw, batch, 8, 0, 1, 1.0)
        out['at_8_threads'] = {'value': batch / sec8, 'cores': 8, 'seconds_per_step': sec8, 'sample': '%d un-warmed iteration(s)' % n8}
    if extras and phys > cores:
        # SURVEY 8(d) says "all physical cores": that figure too, from ONE un-warmed iteration (it is the slow one: ~54 s)
        seca, na, wa = cpu_step_seconds(opt_kwargs, hw, batch, phys, 0, 1, 1.0)
        out['all_cores'] = {'value': batch / seca, 'cores': phys, 'seconds_per_step': seca, 'sample': '%d un-warmed iteration(s)' % na}
    return out


def make_dense_label_data(data, seed):
    """The same batch with iid uniform random labels: every rectangle of every resolution crosses a label boundary."""
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, 4, tuple(data['label'].shape), generator=g, dtype=torch.int64).to(data['label'].dtype)
    return dict(data, label=lab.to(data['label'].device))


def timed_steps(step, steps, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


G_FWD_GFLOP_PER_SAMPLE_256 = 273.15      # netG forward at 256x256, ngf 64


def secondary_metrics(args, dev_index, data):
    """SURVEY 8(d) "Secondary": G-only forward images/s at config 2 (256x256, batch 8, fp32, eval mode, given style codes)
    against the fp32 MFMA peak, and the bf16 inference path (netE + netG + resize to 400x640 + 0..255)."""
    import contextlib, io
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_model import Pix2PixModel
    from seg2eye_amd.postprocessor import ImageProcessor
    out = {}
    for dt in ('fp32', 'bf16'):
        opt = default_opt(ngf=args.ngf, ndf=args.ngf, crop_size=args.size, aspect_ratio=1.0, batchSize=args.batch,
                          compute_dtype=dt, gpu_ids=[dev_index])
        with contextlib.redirect_stdout(io.StringIO()):
            model = Pix2PixModel(opt)
        fill_weights(model)
        model.eval()
        with torch.no_grad():
            seg, style, _ = model.preprocess_input(dict(data))
            w, _ = model.encode_w(style)

            def g_only():
                return model.generate_fake_from_stylecode(seg, w)

            def inference():
                return ImageProcessor.to_255resized_imagebatch(model.forward(dict(data), mode='inference'))
            sec = timed_steps(g_only, 20, 5)
            ent = {'images_per_s': args.batch / sec, 'ms_per_batch': sec * 1e3, 'batch': args.batch}
            if args.size == 256 and args.ngf == 64:
                # the fraction of the MFMA peak is taken on DENSE label maps (iid classes: no rectangle is label-uniform, every
                # multiply-add of SURVEY 8(d)'s count is executed); on the bench's ellipse maps the label-sparse launches skip
                # work, so the same count over that time is an algorithmic rate, not a fraction of the peak
                dense = make_dense_label_data(data, 4321)
                seg_d, _, _ = model.preprocess_input(dict(dense))
                sec_d = timed_steps(lambda: model.generate_fake_from_stylecode(seg_d, w), 20, 5)
                peak = MFMA_PEAK_F32 if dt == 'fp32' else MFMA_PEAK_BF16
                tf_d = G_FWD_GFLOP_PER_SAMPLE_256 * args.batch / sec_d / 1e3
                ent.update({'algorithmic_tflops': G_FWD_GFLOP_PER_SAMPLE_256 * args.batch / sec / 1e3, 'peak': peak,
                            'dense_labels': {'images_per_s': args.batch / sec_d, 'ms_per_batch': sec_d * 1e3, 'executed_tflops': tf_d},
                            'frac': tf_d / peak})
            out['g_forward_' + dt] = ent
            if dt == 'bf16':
                sec = timed_steps(inference, 20, 5)
                out['inference_bf16'] = {'images_per_s': args.batch / sec, 'ms_per_batch': sec * 1e3, 'batch': args.batch,
                                         'what': 'netE on 4 style images + netG (eval) + resize to 400x640 + 0..255, inputs resident'}
        del model
    out.update(secondary_train_steps(args, dev_index))
    out['what'] = ('config 2 of BASELINE.json: netG forward only, eval mode, %dx%d batch %d, style codes given; algorithmic FLOPs '
                   '= SURVEY 8(d) (%.2f GFLOP per sample); frac = the rate on dense (iid) label maps / the dense MFMA peak of the dtype'
                   % (args.size, args.size, args.batch, G_FWD_GFLOP_PER_SAMPLE_256))
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) as
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` in a CHILD process (subprocess.run) and hand
    its exit status back.  The parent only ever SPAWNS a child -- never `os.exec*` from here: `torch.cuda.device_count()` may
    initialise the HIP runtime in this process (it falls back to hipGetDeviceCount when amdsmi is not importable), and a
    process that has touched the GPU must not be replaced by another GPU program on this pool.  More ranks than GPUs are
    refused unless S2E_DIST_BACKEND=gloo asks for the dry run in which the ranks share a device (RCCL refuses two ranks on one
    GPU).  The rendezvous port is probed by bind-then-close: on a shared box another process can take it before the ranks
    bind it (the launch then fails with EADDRINUSE and is simply re-run; MASTER_PORT in the environment overrides the probe)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if n > have and os.environ.get('S2E_DIST_BACKEND') != 'gloo':
        print('bench.py: --gpus %d but %d GPU(s) visible (S2E_DIST_BACKEND=gloo runs the ranks on shared devices as a dry run)'
              % (n, have), file=sys.stderr)
        return 2
    if os.environ.get('MASTER_PORT'):
        port = int(os.environ['MASTER_PORT'])
    else:
        with socket.socket() as sk:                     # a free rendezvous port on the loopback interface
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: what this pool's driver supports (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode     # the ranks inherit stdout: rank 0's JSON line passes through


def _git_head():
    try:
        return subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, timeout=5).stdout.strip()
    except Exception:                                    # noqa: BLE001 -- no git on the GPU box: S2E_GIT_HEAD names the commit there
        return ''


BATCH_SEEDS = (1234, 77, 2024, 5)          # the timed loop's batches (rank r: + r)
TRAIN_GFLOP_PER_SAMPLE = {(256, 256): 1239.4, (640, 384): 4528.9}     # G step + D step, ngf = ndf = 64


def secondary_train_steps(args, dev_index):
    """Two more G+D train-step rates beside the headline, hipGraph replays, inputs resident:
    train_step_fp32 -- the reference's own arithmetic (fp32 storage, exact-fp32 MFMA) on the headline's workload;
    cfg5_bs4 -- config 5's per-GPU workload: 640x384 (--crop_size 384 --aspect_ratio 0.6), batch 4, bf16."""
    import contextlib, io
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
    from seg2eye_amd import synthetic as syn
    out = {}
    dev = torch.device('cuda', dev_index)
    for key, kw, (h, w), batch, steps in (('train_step_fp32', dict(crop_size=args.size, aspect_ratio=1.0, compute_dtype='fp32'), (args.size, args.size), args.batch, 5),
                                          ('cfg5_bs4', dict(crop_size=384, aspect_ratio=0.6, compute_dtype='bf16'), (640, 384), 4, 10)):
        opt = default_opt(ngf=args.ngf, ndf=args.ngf, batchSize=batch, gpu_ids=[dev_index], hip_graphs=True, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Pix2PixTrainer(opt)
        fill_weights(tr.pix2pix_model)
        b = syn.make_batch(batch, h, w, seed=1234)
        data = {'label': torch.from_numpy(b['label']).to(dev), 'style_image': torch.from_numpy(b['style_image']).to(dev),
                'target': torch.from_numpy(b['target']).to(dev), 'filename': b['filename']}

        def step():
            tr.run_generator_one_step(dict(data))
            tr.run_discriminator_one_step(dict(data))
        sec = timed_steps(step, steps, 3)
        losses = {k: float(v.detach().float().mean()) for k, v in tr.get_latest_losses().items()}
        ent = {'images_per_s': batch / sec, 'ms_per_step': sec * 1e3, 'batch': batch, 'size': [h, w], 'dtype': kw['compute_dtype'],
               'hip_graphs': bool(tr.use_graphs and tr.graph_G is not None), 'finite': bool(all(np.isfinite(list(losses.values()))))}
        gf = TRAIN_GFLOP_PER_SAMPLE.get((h, w)) if args.ngf == 64 else None
        if gf:
            peak = MFMA_PEAK_F32 if kw['compute_dtype'] == 'fp32' else MFMA_PEAK_BF16
            ent.update({'algorithmic_tflops': gf * batch / sec / 1e3, 'peak': peak, 'algorithmic_frac': gf * batch / sec / 1e3 / peak})
        out[key] = ent
        del tr
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='per-GPU batch (BASELINE: 8)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--ngf', type=int, default=64)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-extras', action='store_true', help='also time the CPU oracle on 8 threads and on all physical cores')
    ap.add_argument('--no-kernel-events', action='store_true')
    ap.add_argument('--no-graphs', action='store_true', help='launch eagerly instead of replaying hipGraphs')
    ap.add_argument('--no-extras', action='store_true', help='skip the dense_labels / eager / secondary measurements')
    ap.add_argument('--exchange', default='overlap', choices=['after_backward', 'overlap'],
                    help='--gpus > 1: all-reduce each gradient group of the G arena as soon as the backward has finished it (the '
                         'G step replays as one hipGraph segment per group, the collective of group k runs beside segment k+1; '
                         'the default) or all-reduce the whole arena after the backward (one graph, one exposed exchange)')
    ap.add_argument('--grad-dtype', default='fp32', choices=['fp32', 'bf16'], help='--gpus > 1: payload of the gradient exchange')
    ap.add_argument('--grad-exchange', default='allreduce', choices=['allreduce', 'direct'],
                    help="--gpus > 1: the backend's all-reduce, or all-to-all + owner sum + all-gather (distributed.FlatGradSync)")
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    from seg2eye_amd import distributed as sdist, ops
    from seg2eye_amd.options import default_opt
    from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
    rank, world, local_rank = sdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dev_index = local_rank % max(torch.cuda.device_count(), 1)     # (== local_rank on a full node; lets a 2-rank gloo
    torch.cuda.set_device(dev_index)                               #  dry run share the single GPU of a test box)
    dev = torch.device('cuda', dev_index)

    opt_kwargs = dict(ngf=args.ngf, ndf=args.ngf, crop_size=args.size, aspect_ratio=1.0, batchSize=args.batch,
                      compute_dtype=args.dtype, gpu_ids=[dev_index], hip_graphs=not args.no_graphs,
                      no_overlap_allreduce=(args.exchange == 'after_backward'),
                      grad_dtype=args.grad_dtype, grad_exchange=args.grad_exchange)
    opt = default_opt(**opt_kwargs)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        trainer = Pix2PixTrainer(opt)
    fill_weights(trainer.pix2pix_model)
    # FOUR batches, resident in HBM before timing, taken in rotation (what train.py runs -- the label-sparse launches'
    # rectangle lists, class tables and uniform fractions differ from batch to batch; the first one alone is `single_batch` below)
    datas = [make_data(args.batch, args.size, seed + rank, dev) for seed in BATCH_SEEDS]
    data = datas[0]
    turn = [0]

    exchange_desc = trainer.sync_G.describe()

    def step():
        d = datas[turn[0] % len(datas)]
        turn[0] += 1
        trainer.run_generator_one_step(dict(d))
        trainer.run_discriminator_one_step(dict(d))

    for _ in range(args.warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    # did the timed steps really run as hipGraph replays? (a failed capture falls back to eager launches and clears the option)
    graphs_ran = bool(trainer.use_graphs and trainer.graph_G is not None and trainer.graph_D is not None)
    # SURVEY 8(d) quotes the steady-state MEDIAN: a second, per-step-synchronised pass over the same number of steps gives
    # the distribution (the contract's `value` stays the whole-region mean above; the per-step syncs cost a launch gap each)
    per_step = []
    for _ in range(args.steps):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per_step.append((time.perf_counter() - t1) * 1e3)
    # Beside the headline (rank 0's line only, one GPU only, never inside the timed region above):
    extras = {}
    if world == 1 and not args.no_extras:
        def step_single():
            trainer.run_generator_one_step(dict(data))
            trainer.run_discriminator_one_step(dict(data))
        sec = timed_steps(step_single, args.steps, 3)
        extras['single_batch'] = {'value': args.batch / sec, 'unit': 'images/s', 'ms_per_step': sec * 1e3, 'hip_graphs': graphs_ran,
                                  'what': 'the same G+D steps replayed on ONE batch (seed %d: what rounds 1-5 timed)' % BATCH_SEEDS[0]}
        dense = make_dense_label_data(data, 4321)

        def step_dense():
            trainer.run_generator_one_step(dict(dense))
            trainer.run_discriminator_one_step(dict(dense))
        sec = timed_steps(step_dense, args.steps, 3)
        extras['dense_labels'] = {'value': args.batch / sec, 'unit': 'images/s', 'ms_per_step': sec * 1e3, 'hip_graphs': graphs_ran,
                                  'what': 'the same G+D steps on iid uniform random 4-class label maps: no label-uniform '
                                          'rectangle exists, the label-sparse SPADE launches compute every rectangle'}
        was = trainer.opt.hip_graphs
        trainer.opt.hip_graphs = False
        sec = timed_steps(step, args.steps, 2)
        extras['eager'] = {'value': args.batch / sec, 'unit': 'images/s', 'ms_per_step': sec * 1e3,
                           'what': 'the same steps as individual launches (no hipGraph replay): the mode of the overlapped '
                                   'multi-GPU gradient exchange'}
        trainer.opt.hip_graphs = was
    # Per-launch HIP events for the roofline.  A graph replay cannot carry per-launch events, so the
    # same step (same kernels, shapes, data) is re-run eagerly right after the timed region with an
    # event pair around every C-ABI call, on the launch stream.
    # Every rank runs these extra steps (a step contains the gradient all-reduce: rank 0 alone would wait for ever);
    # only rank 0 records events.
    prof_steps = 0
    prof = ops.LaunchProfiler()
    if not args.no_kernel_events:
        trainer.opt.hip_graphs = False
        step()
        torch.cuda.synchronize()
        if rank == 0:
            ops.LaunchProfiler.install(prof)
        # FOUR eager steps, each summed per family on its own; the family's time is the MEDIAN of the four, counted twice (the summary
        # keeps the "two profiled steps" form below): an eager step now and then contains one launch that sits 30-60 ms behind a
        # runtime stall (seen twice in round 6: a family at ten times its time), which a mean over two steps hands to the roofline
        prof_steps = 2
        fam_steps = []
        for _ in range(4):
            prof.reset()
            step()
            torch.cuda.synchronize()
            if rank == 0:
                fam_steps.append(prof.summary())
        ops.LaunchProfiler.install(None)
        if fam_steps:
            fams = fam_steps[0]
            for fam, d in fams.items():
                ms = sorted(ps[fam]['ms'] for ps in fam_steps if fam in ps)
                med = 0.5 * (ms[(len(ms) - 1) // 2] + ms[len(ms) // 2])
                for k in ('launches', 'flops', 'executed_flops', 'bytes'):
                    d[k] *= prof_steps
                d['ms'] = med * prof_steps
            prof.summary = lambda: fams
    losses = {k: float(v.detach().float().mean()) for k, v in trainer.get_latest_losses().items()}
    if not all(np.isfinite(list(losses.values()))):
        raise SystemExit('non-finite losses: %s' % losses)
    #  what the gradient exchange costs the step: the same K steps with the collectives switched off (FlatGradSync.noop:
    # the replicas' weights drift apart from here on -- nothing below depends on them), and each group's exchange on its own
    exchange = None
    if world > 1 and sdist.exchange_active():
        trainer.opt.hip_graphs = not args.no_graphs
        trainer.sync_G.noop = trainer.sync_D.noop = True
        for _ in range(2):
            step()
        torch.distributed.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        trainer.sync_G.noop = trainer.sync_D.noop = False
        ms_noex = float(t.item()) / args.steps * 1e3
        exchange = {'ms_per_step_without_exchange': ms_noex, 'exchange_ms_exposed': elapsed / args.steps * 1e3 - ms_noex,
                    'allreduce_ms_per_group_standalone': {'G': trainer.sync_G.time_groups(), 'D': trainer.sync_D.time_groups()},
                    'what': 'exposed = ms_per_step minus the same steps with the collectives off (max over ranks); per group: the '
                            'exchange of that arena slice alone on an idle device, host-timed on rank 0'}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        global_batch = args.batch * world
        out = {
            'metric': 'images/sec per G+D train step, 256x256 bs=8', 'value': global_batch / (ms / 1e3),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'hip_graphs': graphs_ran,
            'config': {'workload': 'Seg2Eye G+D hinge-GAN train step (G step + D step, TTUR Adam, GAN + GAN_Feat), '
                                   '%dx%d, batch %d per GPU, ngf=ndf=%d, 4 style images, synthetic ellipse labels, 4 batches in rotation'
                                   % (args.size, args.size, args.batch, args.ngf),
                       'global_batch': global_batch, 'parallelism': 'dp%d' % world,
                       'gradient_exchange': (args.exchange if sdist.exchange_active() else 'none'),
                       'graph_segments_G': (len(trainer.graph_G.segments) if graphs_ran else 0)},
            'losses': losses,
        }
        out['ms_per_step_median'] = float(np.median(per_step))
        prof = prof.summary()
        if prof:
            peak = MFMA_PEAK_BF16 if args.dtype == 'bf16' else MFMA_PEAK_F32
            # the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_traffic.py)
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]', 'pmc', 'hbm_traffic.json')))
            pmc_rel, pmc = None, {}
            if cands:
                pmc_rel = os.path.relpath(cands[-1], ROOT)
                pmc = json.load(open(cands[-1]))

            def traffic_of(fam):
                k = pmc.get('kernels', {}).get(fam)
                return (k['hbm_bytes_per_launch'], '%s @ %s' % (pmc_rel, pmc.get('git_head', '?'))) if k else (None, None)
            mfma = {k: v for k, v in prof.items() if v['flops'] > 0}
            hbm = {k: v for k, v in prof.items() if v['flops'] == 0}
            fam = max(mfma, key=lambda k: mfma[k]['ms'])
            d = mfma[fam]
            ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
            exe = d['executed_flops'] / (d['ms'] * 1e-3) / 1e12
            traffic, traffic_src = traffic_of(fam)
            # `achieved` / `frac`: the multiply-adds the matrix pipe really EXECUTED (the label-sparse SPADE launches
            # skip the rectangles they serve from the class table, so the algorithmic count overstates hardware utilisation on
            # these ellipse maps); the SURVEY 8(d) algorithmic figure -- 2*Cin*Cout*k^2*pixels whatever was skipped -- is kept
            # beside it as `algorithmic_tflops` / `algorithmic_frac`.  On dense labels (`dense_labels`) the two coincide.
            out['roofline'] = {'kernel': fam, 'bound': 'mfma', 'achieved': exe, 'peak': peak, 'unit': 'TFLOP/s',
                               'frac': exe / peak, 'algorithmic_tflops': ach, 'algorithmic_frac': ach / peak,
                               'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (PMC, separate passes)',
                               'traffic_source': traffic_src,
                               'algorithmic_bytes_per_launch': d['bytes'] / d['launches'],
                               'algorithmic_gflop_per_launch': d['flops'] / d['launches'] / 1e9,
                               'launches_per_step': d['launches'] / prof_steps, 'ms_per_step': d['ms'] / prof_steps,
                               'gflop_per_step': d['flops'] / prof_steps / 1e9,
                               'executed_gflop_per_step': d['executed_flops'] / prof_steps / 1e9,
                               'measured': 'HIP events around every launch, eager re-run of the timed step'}
            #  the same family's GPU time per step from the newest committed rocprofv3 --kernel-trace --stats summary
            # of this command (tools/rocprof_family_ms.py): kernel durations without the ~12 us of dispatch an eager launch's events carry
            kms = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]', 'kernel_ms_per_step.json')))
            if kms:
                kj = json.load(open(kms[-1]))
                rp = kj.get('families', {}).get(fam, {}).get('ms_per_step')
                if rp:
                    #  the summary is of a committed run -- another commit and another box unless its git_head is the code
                    # being measured now (S2E_GIT_HEAD on the GPU box, git HEAD here): say so instead of letting the figure pass as live
                    # ... judged by the digest of the sources the profile was taken from (tools/pmc_traffic.py source_digest: there is no
                    # .git on a GPU box), else by the commit
                    here = os.environ.get('S2E_GIT_HEAD') or _git_head()
                    same = bool(here) and str(kj.get('git_head', '?')).startswith(here[:7])
                    if kj.get('source_digest'):
                        sys.path.insert(0, os.path.join(ROOT, 'tools'))
                        from pmc_traffic import source_digest
                        same = source_digest() == kj['source_digest']
                    out['roofline'].update({'rocprof_ms_per_step': rp, 'rocprof_frac': d['executed_flops'] / prof_steps / (rp * 1e-3) / 1e12 / peak,
                                            'rocprof_source': '%s @ %s' % (os.path.relpath(kms[-1], ROOT), kj.get('git_head', '?')),
                                            'rocprof_stale': not same})
            if (args.size, args.size) in TRAIN_GFLOP_PER_SAMPLE and args.ngf == 64:
                # the whole step against the dense MFMA peak: SURVEY 8(d)'s algorithmic FLOPs of a G+D step / the timed step
                step_tf = TRAIN_GFLOP_PER_SAMPLE[(args.size, args.size)] * args.batch / 1e3
                out['roofline']['whole_step'] = {'algorithmic_tflop': step_tf, 'tflops': step_tf / (ms * 1e-3), 'frac': step_tf / (ms * 1e-3) / peak,
                                                 'what': 'SURVEY 8(d) algorithmic FLOPs of one G+D step / ms_per_step / the dense MFMA peak'}
            if hbm:
                # the HBM-bound class (north_star: "HBM GB/s against the roofline"): its dominant family by time, algorithmic
                # bytes per SURVEY 8(d) / DESIGN 3.5 over the same HIP-event durations, against 8 TB/s
                hf = max(hbm, key=lambda k: hbm[k]['ms'])
                h = hbm[hf]
                gbs = h['bytes'] / (h['ms'] * 1e-3) / 1e9
                tr_h, tr_src = traffic_of(hf)
                out['roofline_hbm'] = {'kernel': hf, 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                       'frac': gbs / HBM_PEAK_GBS, 'traffic': tr_h, 'traffic_unit': 'HBM bytes per launch (PMC, separate passes)',
                                       'traffic_source': tr_src, 'algorithmic_bytes_per_launch': h['bytes'] / h['launches'],
                                       'launches_per_step': h['launches'] / prof_steps, 'ms_per_step': h['ms'] / prof_steps,
                                       'measured': 'HIP events around every C-ABI call (all kernels of the call), eager re-run of the timed step'}
            out['kernels'] = {k: dict({'launches_per_step': v['launches'] / prof_steps, 'ms_per_step': v['ms'] / prof_steps,
                                       'algorithmic_bytes_per_launch': v['bytes'] / max(v['launches'], 1),
                                       'pmc_bytes_per_launch': traffic_of(k)[0]},
                                      **({'tflops': v['executed_flops'] / (v['ms'] * 1e-3) / 1e12,
                                          'algorithmic_tflops': v['flops'] / (v['ms'] * 1e-3) / 1e12} if v['flops'] > 0 else
                                         {'gbs': v['bytes'] / (v['ms'] * 1e-3) / 1e9,
                                          # the rate of the bytes the PMC passes MEASURED (algorithmic counts overstate a kernel that is
                                          # latency-bound or served from cache: label_conv, spectral_norm -- VERDICT r3 #6)
                                          'gbs_pmc': (traffic_of(k)[0] * v['launches'] / (v['ms'] * 1e-3) / 1e9) if traffic_of(k)[0] else None}))
                              for k, v in prof.items()}
        out.update(extras)
        if world == 1 and not args.no_extras:
            del trainer
            torch.cuda.empty_cache()
            out['secondary'] = secondary_metrics(args, dev_index, data)
        if world > 1:
            out['rccl_ranks'] = world
        if sdist.exchange_active():
            out['config']['exchange'] = exchange_desc        # payload, algorithm, bucket size, NCCL_ALGO / NCCL_PROTO as set
        if exchange is not None:
            out['exchange'] = exchange
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(opt_kwargs, args.size, args.batch, extras=args.cpu_baseline_extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
