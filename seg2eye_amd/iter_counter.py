"""IterationCounter (reference util/iter_counter.py:11-83): epoch / step bookkeeping, the `iter.txt` resume record and
the print / save cadence, with the reference's arithmetic (steps are counted in SAMPLES: += batchSize per iteration)."""
import os
import time

import numpy as np


class IterationCounter:
    def __init__(self, opt, dataset_size):
        self.opt, self.dataset_size = opt, dataset_size
        self.first_epoch, self.epoch_iter = 1, 0
        self.total_epochs = opt.niter + opt.niter_decay
        self.current_epoch = self.first_epoch
        self.iter_record_path = os.path.join(opt.checkpoints_dir, opt.name, 'iter.txt')
        if opt.isTrain and opt.continue_train:
            try:
                self.first_epoch, self.epoch_iter = (int(v) for v in np.loadtxt(self.iter_record_path, delimiter=',', dtype=int))
                print('Resuming from epoch %d at iteration %d' % (self.first_epoch, self.epoch_iter))
            except Exception:
                print('Could not load iteration record at %s. Starting from beginning.' % self.iter_record_path)
        self.total_steps_so_far = (self.first_epoch - 1) * dataset_size + self.epoch_iter
        self.last_iter_time = self.epoch_start_time = time.time()
        self.time_per_iter = 0.0

    def training_epochs(self):
        return range(self.first_epoch, self.total_epochs + 1)

    def record_epoch_start(self, epoch):
        self.epoch_start_time = self.last_iter_time = time.time()
        self.epoch_iter = 0
        self.current_epoch = epoch

    def record_one_iteration(self):
        now = time.time()
        self.time_per_iter = (now - self.last_iter_time) / self.opt.batchSize
        self.last_iter_time = now
        self.total_steps_so_far += self.opt.batchSize
        self.epoch_iter += self.opt.batchSize

    def record_epoch_end(self):
        print('End of epoch %d / %d \t Time Taken: %d sec' % (self.current_epoch, self.total_epochs, time.time() - self.epoch_start_time))
        if self.current_epoch % self.opt.save_epoch_freq == 0:
            self._write(self.current_epoch + 1, 0)

    def record_current_iter(self):
        self._write(self.current_epoch, self.epoch_iter)

    def _write(self, epoch, it):
        os.makedirs(os.path.dirname(self.iter_record_path), exist_ok=True)
        np.savetxt(self.iter_record_path, (epoch, it), delimiter=',', fmt='%d')
        print('Saved current iteration count at %s.' % self.iter_record_path)

    def needs_saving(self):
        return (self.total_steps_so_far % self.opt.save_latest_freq) < self.opt.batchSize

    def needs_printing(self):
        return (self.total_steps_so_far % self.opt.print_freq) < self.opt.batchSize

    def needs_displaying(self):                                  # util/iter_counter.py:76-77
        return (self.total_steps_so_far % self.opt.display_freq) < self.opt.batchSize

    def needs_full_validation(self):                             # util/iter_counter.py:79-80
        return (self.total_steps_so_far % self.opt.full_val_freq) < self.opt.batchSize
