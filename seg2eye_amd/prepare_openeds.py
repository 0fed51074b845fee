"""OpenEDS folder tree -> the per-user store the dataset reads (reference data/prepare_openeds.py:16-138), SURVEY 8 f4.

    OpenEDSPreparator(base_path, limit=-1, verbose=False, n_jobs=8, out_filename='openeds.h5').run()

Input (the unzipped challenge data): `<base>/OpenEDS_{train,validation,test}_userID_mapping_to_images.json` -- a list of
{"id", "semantic_segmenation_images", "generative_images", "sequence_images"} -- and the three folders
Semantic_Segmentation_Dataset/{subset}/{images,labels}, Generative_Dataset/{subset}[/labels], Sequence_Dataset/{subset}.
Output layout (what seg2eye_amd/openeds_dataset.py documents): /{subset}/{user}/{images_ss, labels_ss, images_gen,
images_seq | labels_gen} uint8 (n, 640, 400) + `<name>_filenames` S13; image file names lose their 4-character extension,
label entries keep the image name they belong to (prepare_openeds.py:51, 69).

Differences from the reference, both forced by this image: images are read with PIL instead of imageio (same pixel values
for the 8-bit PNGs; multi-channel files are averaged over channels like prepare_openeds.py:46-47), and the store is an HDF5
file only when h5py is installed -- otherwise the same tree goes to `<out_filename>.npz` with '/'-joined keys, which
`load_store` turns back into the nested mapping OpenEDSDataset(opt, store=...) takes."""
import json
import os

import numpy as np
from PIL import Image


class OpenEDSPreparator:
    FOLDER_SEMANTIC_SEGMENTATION = 'Semantic_Segmentation_Dataset'
    FOLDER_GENERATIVE = 'Generative_Dataset'
    FOLDER_SEQUENTIAL = 'Sequence_Dataset'

    def __init__(self, base_path, limit=-1, verbose=False, n_jobs=8, out_filename='openeds.h5'):
        self.base_path = base_path
        self.limit = limit - 1 if limit > 0 else np.inf        # prepare_openeds.py:23
        self.verbose = verbose
        self.n_jobs = n_jobs
        self.path_out = os.path.join(base_path, out_filename)

    # ---- readers (prepare_openeds.py:29-73)
    def load_and_preprocess(self, filename, path):
        path_image = os.path.join(path, filename)
        try:
            img = np.asarray(Image.open(path_image))
        except (OSError, ValueError):
            print('Could not read file from %s' % path_image)
            return None
        if img.ndim > 2:
            img = np.mean(img, axis=2)
        return img, filename[:-4]                               # no extension in the stored name

    def parallel_load_and_preprocess(self, img_ids, path_images):
        if self.n_jobs and self.n_jobs > 1 and len(img_ids) > 64:
            from joblib import Parallel, delayed
            result = Parallel(n_jobs=self.n_jobs, verbose=3 if self.verbose else 0)(
                delayed(self.load_and_preprocess)(i, path_images) for i in img_ids)
        else:
            result = [self.load_and_preprocess(i, path_images) for i in img_ids]
        ok = [r for r in result if r is not None]
        images, filenames = zip(*ok) if ok else ((), ())
        return images, filenames, len(result) - len(ok)

    def create_dataset_images(self, path, img_ids, group_user, ds_name):
        images, filenames, n_errors = self.parallel_load_and_preprocess(img_ids, path)
        group_user[ds_name] = np.array(images).astype(np.uint8)
        group_user[ds_name + '_filenames'] = np.array(filenames).astype('S13')
        print("Dataset '%s' with %d images created." % (ds_name, len(images)))
        if n_errors > 0:
            print('%d skipped images when creating dataset' % n_errors)
        return group_user

    def create_dataset_labels(self, path, img_ids, group_user, ds_name):
        labels = np.array([np.load(os.path.join(path, i[:-3] + 'npy')) for i in img_ids])
        group_user[ds_name] = labels.astype(np.uint8)           # values 0..3
        group_user[ds_name + '_filenames'] = np.array(img_ids).astype('S13')
        print("Dataset '%s' with %d labels created." % (ds_name, len(labels)))
        return group_user

    # ---- the walk (prepare_openeds.py:75-138)
    def build(self):
        """-> nested dict {subset: {user: {dataset name: array}}}."""
        store = {}
        for subset in ('validation', 'train', 'test'):
            print("Processing '%s'..." % subset)
            g_subset = store.setdefault(subset, {})
            with open(os.path.join(self.base_path, 'OpenEDS_%s_userID_mapping_to_images.json' % subset)) as f:
                user_ids = json.load(f)
            n = min(len(user_ids) - 1, self.limit)
            for i, user in enumerate(user_ids):
                print('Processing user %d / %s' % (i, n))
                g = g_subset.setdefault(user['id'], {})
                ss = os.path.join(self.base_path, self.FOLDER_SEMANTIC_SEGMENTATION, subset)
                self.create_dataset_images(os.path.join(ss, 'images'), user['semantic_segmenation_images'], g, 'images_ss')
                if subset != 'test':
                    self.create_dataset_labels(os.path.join(ss, 'labels'), user['semantic_segmenation_images'], g, 'labels_ss')
                    self.create_dataset_images(os.path.join(self.base_path, self.FOLDER_GENERATIVE, subset), user['generative_images'], g, 'images_gen')
                else:                                            # the test split carries labels for the images to generate
                    self.create_dataset_labels(os.path.join(self.base_path, self.FOLDER_GENERATIVE, subset, 'labels'),
                                               user['generative_images'], g, 'labels_gen')
                self.create_dataset_images(os.path.join(self.base_path, self.FOLDER_SEQUENTIAL, subset), user['sequence_images'], g, 'images_seq')
                if i > self.limit:
                    break
        return store

    def run(self):
        print('Processing data from folder %s and saving data to %s' % (self.base_path, self.path_out))
        store = self.build()
        return save_store(store, self.path_out)


def save_store(store, path):
    """HDF5 at `path` (one chunk per image like the reference) when h5py is installed, else `path + '.npz'`."""
    try:
        import h5py
    except ImportError:
        flat = {'%s/%s/%s' % (s, u, k): v for s, users in store.items() for u, d in users.items() for k, v in d.items()}
        np.savez(path + '.npz', **flat)
        return path + '.npz'
    with h5py.File(path, 'w') as f:
        for s, users in store.items():
            gs = f.create_group(s)
            for u, d in users.items():
                g = gs.create_group(u)
                for k, v in d.items():
                    g.create_dataset(k, data=v, chunks=(1,) + v.shape[1:] if v.ndim == 3 else True)
    return path


def load_store(path):
    """The nested mapping OpenEDSDataset(opt, store=...) takes, from the .npz save_store wrote."""
    z = np.load(path)
    store = {}
    for key in z.files:
        s, u, k = key.split('/')
        store.setdefault(s, {}).setdefault(u, {})[k] = z[key]
    return store
