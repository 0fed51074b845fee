"""Where a training run stands, and what is due at this step.

Drop-in for the reference's `IterationCounter` (util/iter_counter.py:11-83) as train.py uses it: same constructor, attribute
and method names, same `iter.txt` (two integers, one per line: epoch to resume in, samples already seen in it) and the same
arithmetic -- progress is counted in SAMPLES, `batchSize` per iteration, and an action with period P is due at the first
step whose sample count has crossed a multiple of P.

Built around two small pieces instead of inline bookkeeping: `_Record` (the resume file) and `_due` (the cadence rule)."""
import os
import time


def _due(samples_seen, period, batch):
    """True on the iteration during which `samples_seen` passed a multiple of `period`."""
    return samples_seen % period < batch


class _Record:
    """<checkpoints_dir>/<name>/iter.txt: what np.savetxt((epoch, samples), fmt='%d') writes / np.loadtxt reads."""

    def __init__(self, path):
        self.path = path

    def load(self):
        with open(self.path) as f:
            fields = f.read().replace(',', ' ').split()
        epoch, samples = (int(float(v)) for v in fields[:2])
        return epoch, samples

    def store(self, epoch, samples):
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        with open(self.path, 'w') as f:
            f.write('%d\n%d\n' % (epoch, samples))
        print('Saved current iteration count at %s.' % self.path)


class IterationCounter:
    def __init__(self, opt, dataset_size):
        self.opt = opt
        self.dataset_size = dataset_size
        self.total_epochs = opt.niter + opt.niter_decay
        self.iter_record_path = os.path.join(opt.checkpoints_dir, opt.name, 'iter.txt')
        self._record = _Record(self.iter_record_path)
        start = (1, 0)
        if opt.isTrain and opt.continue_train:
            try:
                start = self._record.load()
                print('Resuming from epoch %d at iteration %d' % start)
            except (OSError, ValueError):
                print('Could not load iteration record at %s. Starting from beginning.' % self.iter_record_path)
        self.first_epoch, self.epoch_iter = start              # epoch_iter: samples seen within the current epoch
        self.current_epoch = self.first_epoch
        self.total_steps_so_far = (self.first_epoch - 1) * dataset_size + self.epoch_iter
        self.time_per_iter = 0.0                                # seconds per SAMPLE of the latest iteration
        self.time_per_epoch = 0.0
        self.epoch_start_time = self.last_iter_time = time.time()

    # ---- the epoch loop
    def training_epochs(self):
        return range(self.first_epoch, self.total_epochs + 1)

    def record_epoch_start(self, epoch):
        self.current_epoch, self.epoch_iter = epoch, 0
        self.epoch_start_time = self.last_iter_time = time.time()

    def record_one_iteration(self):
        batch = self.opt.batchSize                              # the loader drops the last partial batch
        now = time.time()
        self.time_per_iter, self.last_iter_time = (now - self.last_iter_time) / batch, now
        self.epoch_iter += batch
        self.total_steps_so_far += batch

    def record_epoch_end(self, write=True):
        """write=False: a data-parallel rank other than 0 (one writer for iter.txt)."""
        self.time_per_epoch = time.time() - self.epoch_start_time
        if write:
            print('End of epoch %d / %d \t Time Taken: %d sec' % (self.current_epoch, self.total_epochs, self.time_per_epoch))
        if write and self.current_epoch % self.opt.save_epoch_freq == 0:
            self._record.store(self.current_epoch + 1, 0)       # a resume starts the next epoch from its beginning

    def record_current_iter(self):
        self._record.store(self.current_epoch, self.epoch_iter)

    # ---- what is due now
    def _every(self, period):
        return _due(self.total_steps_so_far, period, self.opt.batchSize)

    def needs_saving(self):
        return self._every(self.opt.save_latest_freq)

    def needs_printing(self):
        return self._every(self.opt.print_freq)

    def needs_displaying(self):
        return self._every(self.opt.display_freq)

    def needs_full_validation(self):
        return self._every(self.opt.full_val_freq)
