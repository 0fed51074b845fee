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


def main(argv=None):
    opt = parse(argv, is_train=True)
    rank, world, local = dist.init_from_env()
    if torch.cuda.is_available():
        opt.gpu_ids = [local]
        torch.cuda.set_device(local)
    dataloader = create_dataloader(opt, rank, world)
    trainer = Pix2PixTrainer(opt)
    iter_counter = IterationCounter(opt, len(dataloader) * opt.batchSize)
    testers = [Tester(opt, dataset_key=k) for k in ('train', 'validation')] if rank == 0 else []
    try:
        for epoch in iter_counter.training_epochs():
            if iter_counter.current_epoch != epoch:
                iter_counter.record_epoch_start(epoch)
            for i, data_i in enumerate(dataloader, start=iter_counter.epoch_iter):
                iter_counter.record_one_iteration()
                if i % opt.D_steps_per_G == 0:
                    trainer.run_generator_one_step(data_i)
                trainer.run_discriminator_one_step(data_i)
                if iter_counter.needs_printing() and rank == 0:
                    losses = trainer.get_latest_losses(include_log_losses=True)
                    msg = '(epoch: %d, iters: %d, time: %.3f) ' % (epoch, iter_counter.total_steps_so_far, iter_counter.time_per_iter)
                    print(msg + ' '.join('%s: %.3f' % (k, float(v.float().mean())) for k, v in losses.items()), flush=True)
                if iter_counter.needs_displaying():
                    with torch.no_grad():                        # (the reference validates the model as it is: train mode)
                        for t in testers:
                            t.run_partial_modes(model=trainer.pix2pix_model, epoch=epoch, n_steps=iter_counter.total_steps_so_far,
                                                log=True, visualize_images=False, limit=opt.validation_limit)
                if iter_counter.needs_full_validation():
                    with torch.no_grad():
                        for t in testers:
                            t.run(trainer.pix2pix_model, mode='full', epoch=epoch, n_steps=iter_counter.total_steps_so_far,
                                  log=True, write_error_log=opt.write_error_log)
                if iter_counter.needs_saving() and rank == 0:
                    print('saving the latest model (epoch %d, total_steps %d)' % (epoch, iter_counter.total_steps_so_far))
                    trainer.save('latest')
                    iter_counter.record_current_iter()
            trainer.update_learning_rate(epoch)
            iter_counter.record_epoch_end()
            if rank == 0 and (epoch % opt.save_epoch_freq == 0 or epoch == iter_counter.total_epochs):
                print('saving the model at the end of epoch %d, iters %d' % (epoch, iter_counter.total_steps_so_far))
                trainer.save('latest')
                trainer.save(epoch)
        print('Training was successfully finished.')
    except (KeyboardInterrupt, SystemExit):
        print('KeyboardInterrupt. Shutting down.')
        print(traceback.format_exc())
    finally:
        if rank == 0:
            print('saving the model before quitting')
            trainer.save('latest')
            iter_counter.record_current_iter()
    return trainer


if __name__ == '__main__':
    main(sys.argv[1:])
