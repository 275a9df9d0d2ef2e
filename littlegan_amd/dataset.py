"""Mirror of /root/reference/dataset.py:7-49 (interface only: `.batches`, `.label`,
`.get_new_iterator().get_next() -> (image[B,H,W,3] in [-1,1], cond[B,cond_dim])`).  The JPEG pipeline is
host I/O outside the hot path; `synthetic=True` (or a missing image_path) yields
CelebA-shaped random batches resident on the device, which is what the metric is quoted on."""
import os
from glob import glob

import numpy as np
import torch

from .utils import data_rescale, soft


class _Iterator:
    def __init__(self, ds):
        self.ds, self.i = ds, 0
        self.order = ds._order()

    def get_next(self):
        if self.i >= len(self.order):
            raise StopIteration  # tf.errors.OutOfRangeError
        b = self.order[self.i]
        self.i += 1
        return self.ds._batch(b)


class CelebA:
    def __init__(self, args):
        print(" - Initializing Dataset...")
        self.args = args
        self.device = torch.device(getattr(args, "device", "cuda"))
        files = glob(os.path.join(args.image_path, "*." + args.image_ext)) if os.path.isdir(str(args.image_path)) else []
        self.synthetic = bool(getattr(args, "synthetic", False)) or not files
        self.label = list(args.attr)
        if self.synthetic:
            self.n = int(getattr(args, "synthetic_images", 64 * args.batch_size))
            self._image_list, self._attributes_list = None, None
        else:
            self._image_list = files
            self._attributes_list = self._get_attr_list(args.attr_path, args.attr)
            self.n = len(files)
        # data parallel: the (seeded, rank-independent) batch order is dealt round-robin to the ranks, every rank takes
        # the same number of batches per epoch (the gradient all-reduce needs matching step counts)
        d = torch.distributed
        self.rank, self.world = (d.get_rank(), d.get_world_size()) if (d.is_available() and d.is_initialized()) else (0, 1)
        self.total_batches = self.n // args.batch_size
        self.batches = self.total_batches // self.world
        self._gen = torch.Generator().manual_seed(int(getattr(args, "seed", 0)) + 17)

    @staticmethod
    def _get_attr_list(attr_file, attr_filter):  # dataset.py:36-46
        with open(attr_file) as f:
            raw = f.read().splitlines()
        out = []
        for item in raw:
            a = item.split()[1:]
            out.append(a if attr_filter is None else [a[x] for x in attr_filter])
        return out

    def _order(self):
        # dataset.py:21-22: batch THEN shuffle with a `prefetch`-sized buffer; a full permutation of batches here
        order = torch.randperm(self.total_batches, generator=self._gen).tolist()
        return order[self.rank::self.world][:self.batches]

    def _batch(self, b):
        a = self.args
        B, H = a.batch_size, a.image_dim
        if self.synthetic:
            g = torch.Generator().manual_seed(1000003 * b + 7)
            img = torch.rand(B, H, H, a.image_channel, generator=g) * 2 - 1
            cond = soft(2.0 * torch.randint(0, 2, (B, len(a.attr)), generator=g).float() - 1.0)
            return img.to(self.device), cond.to(self.device)
        from PIL import Image
        idx = range(b * B, (b + 1) * B)
        imgs = np.stack([np.asarray(Image.open(self._image_list[i]).convert("RGB"), np.float32) for i in idx])
        cond = np.asarray([[float(v) for v in self._attributes_list[i]] for i in idx], np.float32)
        return data_rescale(torch.from_numpy(imgs)).to(self.device), soft(torch.from_numpy(cond)).to(self.device)

    def get_new_iterator(self):
        return _Iterator(self)
