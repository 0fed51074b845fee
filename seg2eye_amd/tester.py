"""Validation / inference harness with the reference Tester's interface (util/tester.py:15-233), metric on the device.

    tester = Tester(opt, dataset_key='validation')
    tester.run(model, mode='full')            # per-image OpenEDS error over the dataset + 'mse/<key>/<mode>/relative'
    tester.run_test(model)                    # uint8 .npy predictions of shape (1, 640, 400) + pred_npy_list.txt

What differs from the reference: the generated batch never leaves the GPU before it is scored (resize to 400 x 640,
0..255 truncation and sqrt(sum d^2)/(H W) are `s2e_resize_to255` + `s2e_openeds_error_u8`); the error log carries error / user /
filename (h5 with h5py, else npz) without the visualiser's side-by-side images; the visdom / TF visualisations are not built
(out of scope); the dataset is whatever `data.create_dataloader`
yields (synthetic, or the OpenEDS H5 dataset) -- batches must carry `target_original` (N, 1, 640, 400) for validation."""
import os
import re
from copy import deepcopy

import numpy as np
import torch

from . import data as data_mod
from .networks.loss import MSECalculator
from .postprocessor import ImageProcessor


class Tester:
    def __init__(self, opt, dataset_key='test', visualizer=None):
        self.opt = deepcopy(opt)
        self.opt.serial_batches = True
        self.opt.no_flip = True
        self.opt.isTrain = False
        self.opt.dataset_key = dataset_key
        if not hasattr(self.opt, 'results_dir'):
            self.opt.results_dir = 'results/'
        self.dataloader = data_mod.create_dataloader(self.opt)
        self.is_validation = self.opt.dataset_key in ['validation', 'train']
        self.N = getattr(self.dataloader, 'N', len(self.dataloader) * self.opt.batchSize)
        self.results_dir = os.path.join(opt.checkpoints_dir, self.opt.name, self.opt.results_dir, self.opt.dataset_key)
        os.makedirs(self.results_dir, exist_ok=True)

    def forward(self, model, data_i):
        """tester.py:44-47: (fake in [-1,1], fake resized to 640 x 400 as 0..255); both stay on the device."""
        with torch.no_grad():
            fake = model.forward(data_i, mode='inference').detach()
        return fake, ImageProcessor.to_255resized_imagebatch(fake)

    def get_iterator(self, dataloader, indices=None):
        """tester.py:49-65: the whole dataset, or the listed samples one by one (`dataset.get_particular`)."""
        if indices is None:
            for data_i in dataloader:
                yield data_i
        else:
            ds = getattr(dataloader, 'dataset', None)
            for i_val in indices:
                yield ds.get_particular(int(i_val)) if ds is not None else dataloader.batch(int(i_val))

    def _get_validation_indices(self, mode, limit):
        """tester.py:152-163: 'rand*' -> random samples, 'fix*' -> first/last sample of every person, 'full' -> all."""
        ds = getattr(self.dataloader, 'dataset', None)
        if 'full' in mode:
            return None
        if ds is None or not hasattr(ds, 'get_validation_indices'):
            raise NotImplementedError("validation mode '%s' needs a dataset with index lists (--dataset_mode openeds)" % mode)
        if 'rand' in mode:
            return ds.get_random_indices(limit)
        if 'fix' in mode:
            return ds.get_validation_indices()[:limit]
        raise ValueError('Invalid mode: %s' % mode)

    def run_batch(self, data_i, model):
        """tester.py:93-97."""
        fake, fake_resized = self.forward(model, data_i)
        target = ImageProcessor.as_batch(data_i['target_original']).to(fake_resized.device).to(torch.uint8)
        errors = MSECalculator.calculate_mse_for_images(fake_resized, target).cpu().numpy()
        return errors, fake, fake_resized, target

    def _prepare_error_log(self):
        """tester.py:67-74: `error_log_<dataset_key>.h5` with one row per sample -- datasets `error` (float64), `user` (S4),
        `filename` (S13).  Written with h5py when it is installed; this image has none, so the same three arrays go to
        `error_log_<dataset_key>.npz` instead (np.load gives the same names).  The reference's fourth dataset, `visualisation`
        (side-by-side images rendered by util/visualizer.py with cv2), belongs to the visualiser: not built."""
        return {'error': np.zeros((self.N,), dtype=np.float64), 'user': np.zeros((self.N,), dtype='S4'),
                'filename': np.zeros((self.N,), dtype='S13')}

    def _write_error_log_batch(self, error_log, data_i, i, errors):
        """tester.py:76-91 without the visualisation."""
        a = i * self.opt.batchSize
        b = min(a + len(errors), self.N)
        error_log['user'][a:b] = np.array(list(data_i['user']), dtype='S4')[:b - a]
        error_log['filename'][a:b] = np.array(list(data_i['filename']), dtype='S13')[:b - a]
        error_log['error'][a:b] = np.asarray(errors, dtype=np.float64)[:b - a]
        return error_log

    def _close_error_log(self, error_log):
        base = os.path.join(self.results_dir, 'error_log_%s' % self.opt.dataset_key)
        try:
            import h5py
        except ImportError:
            np.savez(base + '.npz', **error_log)
            return base + '.npz'
        with h5py.File(base + '.h5', 'w') as f:
            for k, v in error_log.items():
                f.create_dataset(k, data=v)
        return base + '.h5'

    def run_validation(self, model, generator, limit=-1, write_error_log=False):
        """tester.py:99-121."""
        assert self.is_validation, 'Must be in validation mode'
        print('write error log: %s' % write_error_log)
        error_log = self._prepare_error_log() if write_error_log else None
        all_errors, counter = [], 0
        for i, data_i in enumerate(generator):
            counter += data_i['label'].shape[0]
            if counter > limit:
                break
            if i % 10 == 9:
                print('Processing batch %d' % i)
                print('Error so far: %s' % (np.sum(all_errors) / len(all_errors) * 1471))
            errors, _, _, _ = self.run_batch(data_i, model)
            all_errors += list(errors)
            if error_log is not None:
                self._write_error_log_batch(error_log, data_i, i, errors)
        if error_log is not None:
            print('error log: %s' % self._close_error_log(error_log))
        return all_errors

    def print_results(self, all_errors, errors_dict, epoch='n.a.', n_steps='n.a.'):
        print('Validation Results')
        print('------------------')
        print('Error calculated on %d / %d samples' % (len(all_errors), self.N))
        for k in sorted(errors_dict):
            print('  %s, %.2f' % (k, errors_dict[k]))
        print('  dataset_key: %s, model: %s, epoch: %s, n_steps: %s' % (self.opt.dataset_key, self.opt.name, epoch, n_steps))

    def run(self, model, mode, epoch=None, n_steps=None, limit=-1, write_error_log=False, log=False):
        """tester.py:165-176."""
        print("Running validation for mode '%s'..." % mode)
        limit = limit if limit > 0 else self.N
        generator = self.get_iterator(self.dataloader, indices=self._get_validation_indices(mode, limit))
        all_errors = self.run_validation(model, generator, limit=limit, write_error_log=write_error_log)
        errors_dict = MSECalculator.calculate_error_statistics(all_errors, mode=mode, dataset_key=self.opt.dataset_key)
        self.print_results(all_errors, errors_dict, epoch, n_steps)
        return all_errors, errors_dict

    def run_partial_modes(self, model, epoch, n_steps, log=False, visualize_images=False, limit=-1):
        """tester.py:221-233: a quick validation on `limit` random samples ('rand'); a dataset without index lists (the
        synthetic one) is walked from the start for `limit` samples instead.  (No image visualisation: not built.)"""
        try:
            return self.run(model=model, mode='rand', epoch=epoch, n_steps=n_steps, log=log, limit=limit)
        except NotImplementedError:
            return self.run(model=model, mode='full', epoch=epoch, n_steps=n_steps, log=log, limit=limit)

    def run_test(self, model, limit=-1):
        """tester.py:193-219: one uint8 .npy of shape (1, 640, 400) per sample + the list of written paths."""
        filepaths = []
        for i, data_i in enumerate(self.dataloader):
            if limit > 0 and i * self.opt.batchSize >= limit:
                break
            if i % 10 == 0:
                print('Processing batch %d (processed %d images)' % (i, self.opt.batchSize * i))
            names = [re.sub(r'\.', '', f) for f in data_i['filename']]        # test file names carry a dot to remove
            _, fake_resized = self.forward(model, data_i)
            imgs = fake_resized.cpu().numpy()
            for b, name in enumerate(names):
                path = os.path.join(self.results_dir, name + '.npy')
                np.save(path, imgs[b].astype(np.uint8))
                filepaths.append(path)
        list_path = os.path.join(self.results_dir, 'pred_npy_list.txt')
        with open(list_path, 'w') as f:
            for line in filepaths:
                f.write(line)
                f.write(os.linesep)
        print('Written %d files. Filepath: %s' % (len(filepaths), list_path))
        return filepaths
