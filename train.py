#!/usr/bin/env python3
"""Training entry point, flag-compatible with the reference's train.py (train.py:23-116): same option names and defaults
(seg2eye_amd/options.py), same loop -- G step when `i % D_steps_per_G == 0`, then the D step -- same LR schedule,
same checkpoint files (`<checkpoints_dir>/<name>/<epoch>_net_{G,D,E}.pth`, reference state-dict keys) and `iter.txt`
resume record, same validation passes -- every `--display_freq` samples a quick one (`--validation_limit` samples), every
`--full_val_freq` samples a full one, on the train and validation splits, scored with the OpenEDS metric on the device
(seg2eye_amd/tester.py).  Not carried over (out of scope): the visualizer / TF logging, source-tree copy.
Data: `--dataset_mode synthetic` (default) or `openeds` (an H5 file at `--dataroot`; needs h5py).

    python train.py --name run1 --batchSize 8 --aspect_ratio 1.0 --niter 1 --niter_decay 0
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 train.py ...
"""
import sys
import traceback

import torch

from seg2eye_amd import distributed as dist
from seg2eye_amd.data import create_dataloader
from seg2eye_amd.iter_counter import IterationCounter
from seg2eye_amd.options import parse
from seg2eye_amd.pix2pix_trainer import Pix2PixTrainer
from seg2eye_amd.tester import Tester


class TrainingRun:
    """The objects of one run and what is due around every iteration.  `fit()` walks the epochs; the per-iteration
    duties (progress line, quick / full validation, `latest` checkpoint) are a table of (is it due?, do it) pairs
    checked in the reference's order."""

    def __init__(self, opt, rank=0, world=1):
        self.opt, self.rank = opt, rank
        self.dataloader = create_dataloader(opt, rank, world)
        self.trainer = Pix2PixTrainer(opt)
        # samples per epoch AS THIS RANK COUNTS THEM (each rank sees 1/world of an epoch and counts its own samples, so the
        # cadences --print_freq / --save_latest_freq / ... are per-rank sample counts).  The reference passes len(dataloader) --
        # BATCHES -- while its counter advances in samples (train.py:33, util/iter_counter.py:13-31): the same number at its
        # batchSize 1; counting samples throughout keeps `iter.txt` resumes exact at any batch size.
        self.counter = IterationCounter(opt, len(self.dataloader) * opt.batchSize)
        self.world = world
        # validation runs on rank 0 only (it has no collectives); the reference keeps one tester per split
        self.testers = [Tester(opt, dataset_key=split) for split in ('train', 'validation')] if rank == 0 else []
        c = self.counter
        self.duties = ((c.needs_printing, self.report), (c.needs_displaying, self.quick_validation),
                       (c.needs_saving, self.save_latest), (c.needs_full_validation, self.full_validation))
        self.epoch = c.current_epoch

    # ---- duties
    def report(self):
        if self.rank:
            return
        c = self.counter
        losses = self.trainer.get_latest_losses(include_log_losses=True)
        head = '(epoch: %d, iters: %d, time: %.3f) ' % (self.epoch, c.total_steps_so_far, c.time_per_iter)
        print(head + ' '.join('%s: %.3f' % (name, float(v.float().mean())) for name, v in losses.items()), flush=True)

    def quick_validation(self):
        # (the reference validates the model as it is: train mode.  solo: only rank 0 is here -- no per-layer exchange)
        with torch.no_grad(), dist.solo():
            for t in self.testers:
                t.run_partial_modes(model=self.trainer.pix2pix_model, epoch=self.epoch, n_steps=self.counter.total_steps_so_far,
                                    log=True, visualize_images=False, limit=self.opt.validation_limit)
        self.trainer.sync_replica_buffers()                      # rank 0's train-mode pass advanced its u, v / BN statistics

    def full_validation(self):
        with torch.no_grad(), dist.solo():
            for t in self.testers:
                t.run(self.trainer.pix2pix_model, mode='full', epoch=self.epoch, n_steps=self.counter.total_steps_so_far,
                      log=True, write_error_log=self.opt.write_error_log)
        self.trainer.sync_replica_buffers()

    def save_latest(self):
        if self.rank:
            return
        print('saving the latest model (epoch %d, total_steps %d)' % (self.epoch, self.counter.total_steps_so_far))
        self.trainer.save('latest')
        self.counter.record_current_iter()

    # ---- the loop
    def one_epoch(self, epoch):
        c, trainer = self.counter, self.trainer
        self.epoch = epoch
        if c.current_epoch != epoch:                             # equal only at the very start and right after a resume
            c.record_epoch_start(epoch)
        sampler = getattr(self.dataloader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)                             # (DistributedSampler: a new permutation per epoch)
        for i, batch in enumerate(self.dataloader, start=c.epoch_iter):
            c.record_one_iteration()
            if i % self.opt.D_steps_per_G == 0:
                trainer.run_generator_one_step(batch)
            trainer.run_discriminator_one_step(batch)
            for due, act in self.duties:
                if due():
                    act()
        trainer.update_learning_rate(epoch)
        c.record_epoch_end(write=self.rank == 0)
        if self.rank == 0 and (epoch % self.opt.save_epoch_freq == 0 or epoch == c.total_epochs):
            print('saving the model at the end of epoch %d, iters %d' % (epoch, c.total_steps_so_far))
            trainer.save('latest')
            trainer.save(epoch)

    def fit(self):
        try:
            for epoch in self.counter.training_epochs():
                self.one_epoch(epoch)
            print('Training was successfully finished.')
        except (KeyboardInterrupt, SystemExit):
            print('KeyboardInterrupt. Shutting down.')
            print(traceback.format_exc())
        finally:
            if self.rank == 0:
                print('saving the model before quitting')
                self.trainer.save('latest')
                self.counter.record_current_iter()
        return self.trainer


def main(argv=None):
    opt = parse(argv, is_train=True)
    rank, world, local = dist.init_from_env()
    if torch.cuda.is_available():
        dev = local % max(torch.cuda.device_count(), 1)          # (== local on a full node; a 2-rank gloo run may share one GPU)
        opt.gpu_ids = [dev]
        torch.cuda.set_device(dev)
    return TrainingRun(opt, rank, world).fit()


if __name__ == '__main__':
    main(sys.argv[1:])
