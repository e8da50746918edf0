"""``DB23`` -- serving half of /root/reference/code/load.py:23-73,157-273: a GPU-resident
``EMG (41,46,6,100,12)`` tensor, the task/people/repetition masks and the mode-dependent re-slice.

The raw-``.mat`` preprocessing half (code/load.py:75-155) needs the 10 GB Ninapro download and
scipy filtering; it is out of scope.  ``load_stored`` reads the
reference's ``emg.pt`` / ``glove.pt`` if present; ``load_synthetic`` builds Ninapro-shaped tensors
 for benchmarks and tests.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.utils.data as data

from .constants import (AMT_PREDICTION_WINDOWS, EMG_DIM, GLOVE_DIM, MAX_PEOPLE, MAX_TASKS, PEOPLE_IDXS,
                        PREDICTION_WINDOW_SIZE, REPS, REPS_TEST, REPS_TRAIN, TASKS, TEST_PEOPLE_IDXS, TEST_TASKS,
                        TRAIN_PEOPLE_IDXS, TRAIN_TASKS, VOTE, WINDOW_OUTPUT_DIM, d2_idxs, d3_idxs)
from .utils import Glover, torchize

PATH_DIR = os.environ.get("CP_DB23_DIR", "/home/breezy/hci/prosthetics/db23/")   # code/constants.py:56


class DB23(data.Dataset):
    def __init__(self, db2=False, train=True, val=False):
        self.device = torch.device("cuda")
        self.train = train
        self.val = val
        self.raw = False
        self.db2 = db2
        self.tasks_train = torchize(TRAIN_TASKS)
        self.tasks_test = torchize(TEST_TASKS)
        self.tasks = torchize(TASKS)
        self.people_train = torchize(TRAIN_PEOPLE_IDXS)
        self.people_test = torchize(TEST_PEOPLE_IDXS)
        self.people = torchize(PEOPLE_IDXS)
        train_reps, test_reps, reps = torchize(REPS_TRAIN), torchize(REPS_TEST), torchize(REPS)
        self.rep_train = train_reps[:-1] - 1          # code/load.py:43-46
        self.rep_val = train_reps[-1:] - 1
        self.rep_test = test_reps - 1
        self.reps = reps - 1
        self.glover = Glover()

    # -- mode switches (code/load.py:51-64) -------------------------------------------------------------
    def set_train(self):
        self.train, self.val = True, False
        self.load_valid()

    def set_val(self):
        self.train, self.val = False, True
        self.load_valid()

    def set_test(self):
        self.train, self.val = False, False
        self.load_valid()

    # -- residency -----------------------------------------------------------------------------------------
    def load_stored(self, path_dir: str = None):
        """code/load.py:66-73: emg.pt is (people, tasks, reps, 100, 12); kept task-major."""
        path_dir = path_dir or PATH_DIR
        p = os.path.join(path_dir, "data", "emg.pt")
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} not found (the Ninapro tensors are not distributed with the reference); "
                                    "use load_synthetic() / train.py --synthetic")
        self.EMG = torch.load(p, map_location=self.device, weights_only=True).transpose(0, 1)
        self.GLOVE = self.glover.load_stored(path_dir)

    def load_synthetic(self, seed: int = 1234, glove_d: int = 5850):
        """Seeded Ninapro-shaped tensors: class mean + subject offset + unit noise, standardised per channel."""
        g = torch.Generator().manual_seed(seed)
        mu = torch.randn(MAX_TASKS, EMG_DIM, generator=g)
        nu = torch.randn(MAX_PEOPLE, EMG_DIM, generator=g)
        emg = (mu[:, None, None, None, :] + 0.5 * nu[None, :, None, None, :]
               + torch.randn(MAX_TASKS, MAX_PEOPLE, len(REPS), WINDOW_OUTPUT_DIM, EMG_DIM, generator=g))
        flat = emg.reshape(-1, EMG_DIM)
        emg = (emg - flat.mean(0)) / flat.std(0)
        gl = torch.randn(MAX_TASKS, 1, GLOVE_DIM, generator=g) + 0.3 * torch.randn(MAX_TASKS, glove_d, GLOVE_DIM, generator=g)
        self.EMG = emg.contiguous().to(self.device)
        self.GLOVE = self.glover.GLOVE = gl.contiguous().to(self.device)

    # -- masks (code/load.py:157-203) -------------------------------------------------------------------------
    @property
    def tasks_mask(self):
        return torch.cat((self.tasks, torchize([0]))).to(torch.long)      # 40 shuffled grasps, rest last

    @property
    def people_mask(self):
        return torchize(d2_idxs if self.db2 else d3_idxs + len(d2_idxs)).to(torch.long)

    @property
    def rep_mask(self):
        if self.train:
            return torch.cat((self.rep_train, self.rep_test)) if self.db2 else self.rep_train
        if self.val:
            return self.rep_val
        return self.rep_val if self.db2 else self.rep_test

    @property
    def PEOPLE(self):
        return len(self.people_mask)

    @property
    def TASKS(self):
        return len(self.tasks_mask)

    @property
    def REPS(self):
        return len(self.rep_mask)

    @property
    def D(self):
        if self.train:
            return self.PEOPLE * self.REPS * self.OUTPUT_DIM
        return self.PEOPLE * self.REPS * (AMT_PREDICTION_WINDOWS if VOTE else self.OUTPUT_DIM)

    @property
    def OUTPUT_DIM(self):
        if self.train:
            return WINDOW_OUTPUT_DIM
        return WINDOW_OUTPUT_DIM if not VOTE else PREDICTION_WINDOW_SIZE

    # -- re-slice (code/load.py:233-251) ------------------------------------------------------------------------
    def load_valid(self):
        tensor = self.EMG[self.tasks_mask][:, self.people_mask][:, :, self.rep_mask]
        tensor_ = tensor[:, :, :, :WINDOW_OUTPUT_DIM]
        self.EMG_use = tensor_.reshape(-1, EMG_DIM).to(torch.float32).contiguous()
        self.tensor = self.EMG_use.view(-1, self.OUTPUT_DIM, EMG_DIM)      # same memory, vote view
        if self.train or not VOTE:
            assert torch.equal(self.EMG_use[self.D * 2 + 1], tensor_[2].reshape(-1, EMG_DIM)[1]), "indexing is not correct"
        else:
            assert torch.equal(self.tensor[self.D * 2 + 1], tensor_[2].reshape(-1, self.OUTPUT_DIM, EMG_DIM)[1]), \
                "indexing is not correct"
        self.glover.load_valid(self.tasks_mask)

    def __len__(self):
        return self.TASKS * self.D

    def slice_batch(self, idx):
        return self.EMG_use[idx].reshape(-1, 1, 1, EMG_DIM)

    def __getitem__(self, idx):
        if self.raw:
            return self.EMG
        if not self.train and VOTE:
            return self.tensor[idx, :, :].unsqueeze(2)
        return self.slice_batch(idx)
