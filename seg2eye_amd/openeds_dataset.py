"""OpenEDS H5 dataset with the reference's semantics (data/openeds_dataset.py:16-226, data/base_dataset.py:20-147),
SURVEY 8 f4.  The data are licence-gated and h5py is not part of every image, so the class works on any H5-LIKE STORE:

    store[dataset_key][user][name]  ->  array-like with .shape and integer indexing
        names: images_ss, labels_ss, images_gen, images_seq, labels_gen   uint8 (n, 640, 400)
               images_ss_filenames, labels_gen_filenames                  bytes (S13)

`opt.dataroot` is opened with h5py when it is a path (ImportError with a pointer here if h5py is missing); tests and
users without h5py pass a nested dict of numpy arrays (`OpenEDSDataset(opt, store=...)`).

Built: the reference's training recipe -- `--preprocess_mode fixed` (resize to crop_size x round(crop_size / aspect_ratio):
cv2.INTER_NEAREST for masks, PIL bicubic for images), ToTensor + Normalize(0.5, 0.5) -> [-1, 1], horizontal flip
augmentation, the train/validation vs test key sets, style sampling `random` / `first` / `ref` / `ref_random<N>`,
`target_original`, `get_particular`, `get_validation_indices`, `get_random_indices`.  Other preprocess modes raise.
cv2 is not in this image: its nearest-neighbour resize is restated by its rule (source index = floor(dst * src / dst))."""
import re

import numpy as np
import torch
from PIL import Image


def get_params(opt, size, rng=None):
    """base_dataset.py:20-48 for the modes that do not crop ('fixed', 'none'): only the flip decision is random."""
    rng = rng or np.random
    flip = False if getattr(opt, 'no_flip', False) else bool(rng.random() > 0.5)
    return {'crop_pos': (0, 0), 'flip': flip}


def resize_nearest(img, w, h):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST) for a 2-D array: src = min(floor(dst * scale), size - 1)."""
    ih, iw = img.shape[:2]
    ys = np.minimum((np.arange(h) * (ih / h)).astype(np.int64), ih - 1)
    xs = np.minimum((np.arange(w) * (iw / w)).astype(np.int64), iw - 1)
    return img[ys[:, None], xs[None, :]]


def flip(img, do_flip):
    """base_dataset.py:137-146."""
    if not do_flip:
        return img
    if isinstance(img, np.ndarray):
        return np.array(np.flip(img, axis=img.ndim - 1))
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def get_transform(opt, params, mask=False):
    """base_dataset.py:51-80 for preprocess_mode 'fixed'.  mask=True: the label-map variant (nearest, no normalisation,
    numpy in / numpy out); else PIL in -> float tensor (1, H, W) in [-1, 1]."""
    if opt.preprocess_mode != 'fixed':
        # the reference's dataset hands its mask transform numpy arrays (openeds_dataset.py:89-90): the resize / scale / crop
        # branches of base_dataset.get_transform need PIL images and fail there, and 'none' transposes a non-square label map
        raise NotImplementedError("preprocess_mode '%s': only 'fixed' works with the reference's OpenEDS dataset, and only it is built"
                                  % opt.preprocess_mode)
    w = opt.crop_size
    h = round(opt.crop_size / opt.aspect_ratio)
    do_flip = bool(opt.isTrain and not opt.no_flip and params['flip'])

    def tf_mask(m):
        return flip(resize_nearest(np.asarray(m), w, h), do_flip)

    def tf_image(img):
        img = flip(img.resize((w, h), Image.BICUBIC), do_flip)
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).float().div(255.0).unsqueeze(0)     # ToTensor
        return (t - 0.5) / 0.5                                                                            # Normalize((0.5,), (0.5,))
    return tf_mask if mask else tf_image


class OpenEDSDataset(torch.utils.data.Dataset):
    def __init__(self, opt, store=None, style_refs=None, rng=None):
        self.opt = opt
        self.dataset_key = opt.dataset_key
        # the keys of style images, labels and file names differ for the test set (openeds_dataset.py:44-52)
        self.key_style_images = 'images_ss' if self.dataset_key == 'test' else 'images_gen'
        self.label_key = 'labels_ss' if self.dataset_key != 'test' else 'labels_gen'
        self.key_filenames = 'labels_gen_filenames' if self.dataset_key == 'test' else 'images_ss_filenames'
        self._store, self._path = store, None
        if store is None:
            self._path = opt.dataroot
        self.style_image_refs = style_refs
        self.rng = rng or np.random
        h5 = self._h5()
        self.user_ids = list(h5.keys())
        self.N, self.N_start = 0, []
        for user in self.user_ids:
            self.N_start.append(self.N)
            if self.key_filenames in h5[user]:
                self.N += h5[user][self.key_filenames].shape[0]

    def _h5(self):
        if self._store is None:
            try:
                import h5py
            except ImportError as e:
                raise ImportError('reading %s needs h5py, which is not installed here; pass an in-memory store '
                                  '(OpenEDSDataset(opt, store=...), see seg2eye_amd/openeds_dataset.py)' % self._path) from e
            self._store = h5py.File(self._path, 'r')
            if 'ref' in self.opt.style_sample_method and self.style_image_refs is None:
                assert self.opt.style_ref != '', 'You need to provide a h5 file for style references.'
                self.style_image_refs = h5py.File(self.opt.style_ref, 'r')
        return self._store[self.dataset_key]

    def _get_tuple_identifier_from_index(self, index):
        """openeds_dataset.py:67-80: (user id, index within the user) of a flat index."""
        idx_user = 0
        for i in range(len(self.user_ids)):
            if index >= self.N_start[i]:
                idx_user = i
            else:
                break
        return self.user_ids[idx_user], index - self.N_start[idx_user]

    def __len__(self):
        return self.N

    def __getitem__(self, index):
        h5 = self._h5()
        user, idx = self._get_tuple_identifier_from_index(index)
        mask = np.asarray(h5[user][self.label_key][idx])
        params = get_params(self.opt, mask.shape, self.rng)
        mask_tensor = torch.from_numpy(np.ascontiguousarray(get_transform(self.opt, params, mask=True)(mask)))
        filename = h5[user][self.key_filenames][idx]
        filename = filename.decode('utf-8') if isinstance(filename, (bytes, np.bytes_)) else str(filename)
        filename = re.sub(r'\.', '', filename)                  # some file names carry an additional dot
        transform_image = get_transform(self.opt, params)
        style, _, _ = self.get_style_images(user, self.opt.input_ns, transform_image, filename)
        out = {'label': mask_tensor, 'filename': filename, 'user': user, 'style_image': style}
        if self.dataset_key != 'test':                           # ground truth exists
            target = np.array(h5[user]['images_ss'][idx])
            out['target'] = transform_image(Image.fromarray(target, mode='L'))
            # only the ORIGINAL is flipped here; the transformed one was flipped by the transform (openeds_dataset.py:113)
            out['target_original'] = torch.from_numpy(np.expand_dims(flip(target, params['flip']), axis=0).copy()).int()
        return out

    # ---------------------------------------------------------------- style images (openeds_dataset.py:152-210)
    def _sample_style_idx(self, n_images, n, user_id=None, filename=None):
        method, subsets = self.opt.style_sample_method, None
        if method == 'random':
            indices = self.rng.choice(list(range(n_images)), n)
        elif method == 'first':
            indices = list(range(min(n, n_images)))
        elif 'ref' in method:
            entry = self.style_image_refs[self.opt.dataset_key][user_id][filename]
            use_seq = 'subset' in list(entry.keys())
            all_indices = entry['index']
            all_subsets = entry['subset'] if use_seq else None
            if 'random' in method:                               # "ref_random40": sample among the 40 most similar
                reduced_n = re.sub(r'[^\d]', '', method)
                reduced_n = int(reduced_n) if reduced_n else 40
                to_select = self.rng.choice(list(range(reduced_n)), n)
                indices = [all_indices[to_select[i]] for i in range(n)]
                if use_seq:
                    subsets = [all_subsets[to_select[i]] for i in range(n)]
            else:
                indices = all_indices[:n]
                if use_seq:
                    subsets = all_subsets[:n]
        else:
            raise ValueError('Invalid style sampling method: %s' % method)
        return indices, subsets

    def get_style_images(self, user_id, n, transform_image, filename=None):
        h5 = self._h5()
        n_images = h5[user_id][self.key_style_images].shape[0]
        selected_idx, subsets = self._sample_style_idx(n_images, n, user_id=user_id, filename=filename)
        selected_idx = list(selected_idx)
        subset_keys = {b'g': self.key_style_images, b's': 'images_seq'}
        imgs = []
        for i, sel in enumerate(selected_idx):
            key = subset_keys[bytes(subsets[i])] if subsets is not None else self.key_style_images
            if key == 'images_seq':                              # sequence indices were appended behind the generative ones
                sel = sel - n_images
                selected_idx[i] = sel
            imgs.append(np.asarray(h5[user_id][key][int(sel)]))
        tensors = [transform_image(Image.fromarray(im, mode='L')) for im in imgs]
        return torch.stack(tensors), selected_idx, subsets

    # ---------------------------------------------------------------- helpers the Tester uses
    def unsqueeze_batch(self, batch):
        for key in ('style_image', 'target', 'target_original', 'label'):
            if key in batch:
                batch[key] = batch[key].unsqueeze(0)
        return batch

    def get_particular(self, idx):
        b = self.unsqueeze_batch(self[idx])
        b['filename'], b['user'] = [b['filename']], [b['user']]
        return b

    def get_validation_indices(self):
        """openeds_dataset.py:139-143: first and last sample of every person."""
        return list(self.N_start) + [i - 1 for i in self.N_start[1:]] + [self.N - 1]

    def get_random_indices(self, n):
        return self.rng.choice(list(range(self.N)), n)


def create_dataloader(opt, store=None, style_refs=None, rank=0, world=1):
    """data/__init__.py:43-59: batch_size, shuffle = not serial_batches, nThreads workers, drop_last = isTrain.
    world > 1 (data parallel, new relative to the reference): every rank walks its own 1/world of each epoch's permutation
    (torch DistributedSampler; call `loader.sampler.set_epoch(epoch)` per epoch -- train.py does), so an epoch is still ONE
    pass over the dataset and the per-epoch LR decay / save_epoch_freq / iter.txt mean what they do on one GPU."""
    ds = OpenEDSDataset(opt, store=store, style_refs=style_refs)
    print('dataset [%s] of size %d was created' % (type(ds).__name__, len(ds)))
    workers = int(opt.nThreads) if store is None else 0
    if world > 1 and opt.isTrain:
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=not opt.serial_batches,
                                                                  drop_last=True)
        dl = torch.utils.data.DataLoader(ds, batch_size=opt.batchSize, sampler=sampler, num_workers=workers, drop_last=True)
    else:
        dl = torch.utils.data.DataLoader(ds, batch_size=opt.batchSize, shuffle=not opt.serial_batches, num_workers=workers,
                                         drop_last=opt.isTrain)
    dl.N = ds.N
    return dl
