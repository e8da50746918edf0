"""``TaskWrapper`` / ``Glover`` serving half / ``torchize`` -- the surface of
/root/reference/code/utils.py:18-76,185-195,220-224,248-254 on GPU-resident tensors.

The offline preprocessing half of the reference's utils.py (Butterworth filter, moving RMS,
Welford statistics: code/utils.py:79-183,197-246) is out of scope.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .constants import GLOVE_DIM

_DEVICE = "cuda"


def torchize(X):
    """code/utils.py:18-19"""
    return torch.from_numpy(np.array(X)).to(torch.device(_DEVICE))


class Glover:
    """Serving half of code/utils.py:185-254: a resident ``GLOVE (41, D_g, 20)`` tensor, re-sliced per
    mode by ``load_valid`` and indexed by flat row.  (In contrastive mode the model never reads the
    values -- only the shape, SURVEY.md section 0 item 3 -- but the dataset API delivers them.)"""

    def __init__(self):
        self.device = torch.device(_DEVICE)
        self.GLOVE = None

    def load_stored(self, path_dir: str):
        p = os.path.join(path_dir, "data", "glove.pt")
        self.GLOVE = torch.load(p, map_location=self.device, weights_only=True)
        return self.GLOVE

    def load_valid(self, tasks_mask):
        tensor = self.GLOVE[tasks_mask]
        self.D = self.GLOVE.shape[1]
        self.GLOVE_use = tensor.reshape(-1, GLOVE_DIM)

    def __getitem__(self, idx):
        return self.GLOVE_use[idx]


class TaskWrapper:
    """code/utils.py:21-76.  ``__getitem__`` keeps the reference's per-item contract (one group of 41
    windows, one per class); ``batch(perm)`` is the fused path: the B items of a DataLoader batch and
    their default_collate in ONE cp_gather_groups launch."""

    def __init__(self, dataset):
        self.dataset = dataset
        self.device = torch.device(_DEVICE)

    def return_rand(self, D):
        # code/utils.py:34-36: per-class random permutation of that class's row range
        b = torch.arange(self.dataset.TASKS, device=self.device, dtype=torch.long).reshape(self.dataset.TASKS, 1) * D
        return torch.rand((self.dataset.TASKS, D), device=self.device).argsort(dim=-1) + b

    def reset(self):
        self.emg_rand = self.return_rand(self.dataset.D).contiguous()
        self.glove_rand = self.return_rand(self.dataset.glover.D).contiguous()
        # (the reference also draws an unused randperm(T*D), code/utils.py:41; not reproduced)

    def __getattr__(self, name):
        return getattr(self.dataset, name)

    def __len__(self):
        return self.dataset.D

    def __getitem__(self, idx):
        tensor_emg = self.dataset[self.emg_rand[:, idx]]
        tensor_glove = self.dataset.glover[self.glove_rand[:, idx % self.dataset.glover.D]]
        label = torch.arange(self.dataset.TASKS, device=self.device, dtype=torch.long)
        return tensor_emg.to(torch.float32), tensor_glove.to(torch.float32), label

    def batch(self, perm: torch.Tensor, engine=None, with_glove: bool = True):
        """== default_collate([self[i] for i in perm]) (code/train.py:86,95).
        EMG (B,41,V,1,12) f32, GLOVE (B,41,20) f32, label (B,41) int64."""
        from .engine import gather_groups
        ds = self.dataset
        B = perm.numel()
        V = 1 if ds.train else ds.OUTPUT_DIM
        EMG = gather_groups(ds.EMG_use, self.emg_rand, perm, V).reshape(B, ds.TASKS, V, 1, 12)
        GLOVE = None
        if with_glove:
            gidx = self.glove_rand[:, perm % ds.glover.D].t()               # (B,41)
            GLOVE = ds.glover.GLOVE_use[gidx].to(torch.float32)
        label = torch.arange(ds.TASKS, device=self.device, dtype=torch.long).repeat(B, 1)
        return EMG, GLOVE, label

    def set_train(self):
        self.dataset.set_train()
        self.reset()

    def set_val(self):
        self.dataset.set_val()
        self.reset()

    def set_test(self):
        self.dataset.set_test()
        self.reset()


class GroupLoader:
    """Stand-in for ``data.DataLoader(dataset, batch_size=B, shuffle=True)`` of code/train.py:86 for a
    TaskWrapper whose data already lives on the GPU: a random permutation of range(len(dataset)) cut into
    batches (the last one may be short, as with drop_last=False), each produced by TaskWrapper.batch."""

    def __init__(self, dataset: TaskWrapper, batch_size: int, shuffle: bool = True, engine=None, rank: int = 0,
                 world: int = 1, generator: torch.Generator = None):
        self.dataset, self.batch_size, self.shuffle, self.engine = dataset, batch_size, shuffle, engine
        self.rank, self.world, self.generator = rank, world, generator

    def __len__(self):
        n = len(self.dataset) // self.world
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator) if self.shuffle else torch.arange(n)
        per = n // self.world                       # contiguous shard of the shuffled order per rank
        order = order[self.rank * per:(self.rank + 1) * per].to(self.dataset.device)
        for i in range(0, per, self.batch_size):
            yield self.dataset.batch(order[i:i + self.batch_size], self.engine)
