"""CPU oracle for the contrastive sEMG training path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, pure-PyTorch (CPU, fp32) restatement of the one hot
path of FibonacciDude/ContrastiveProsthetics that this repository accelerates.
It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``contrastiveprosthetics_amd``) never imports anything from ``oracle/``.

Parity status: PINNED against the reference itself.  ``tools/make_golden.py``
imports the reference's ``code/models.py`` / ``code/utils.py`` / ``code/load.py``
on CPU in the build container and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those
vectors.  Versus the reference author's original (unpinned) PyTorch environment
the numerics are "parity unpinned"; parity is defined against torch 2.10 CPU
fp32.

Each function cites the reference lines (under /root/reference/) it restates.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# constants (values of code/constants.py:1-97, recomputed, not copied)
# ----------------------------------------------------------------------------
EMG_DIM = 12                  # code/constants.py:97
GLOVE_DIM = 20                # code/constants.py:96   (22 sensors minus 2)
N_TASKS = 41                  # code/constants.py:45-48 (17 + 23 + rest)
WINDOW_OUTPUT_DIM = 100       # code/constants.py:71-72,90 (1 s at 100 Hz)
VOTE_SAMPLES = 25             # code/constants.py:76-77 (250 ms at 100 Hz)
VOTE_RUNS = WINDOW_OUTPUT_DIM // VOTE_SAMPLES   # code/constants.py:78  (4)
PREDICTION_WINDOW = 250       # code/constants.py:76 (used as a loop bound, models.py:153)
BN_EPS = 1e-5
N_D2, N_D3 = 40, 6


def split_constants() -> Dict[str, np.ndarray]:
    """Seed-0 permutations of code/constants.py:3-46.

    The reference draws, in this order, from numpy's legacy global generator
    seeded with 0: permutation(40), permutation(6), shuffle(tasks 1..17),
    shuffle(tasks 18..40).
    """
    rs = np.random.RandomState(0)
    d2_idxs = rs.permutation(N_D2)
    d3_idxs = rs.permutation(N_D3)
    tasks_a = np.arange(1, 18, dtype=np.uint8)
    tasks_b = np.arange(18, 41, dtype=np.uint8)
    rs.shuffle(tasks_a)
    rs.shuffle(tasks_b)
    tasks = np.concatenate((tasks_a, tasks_b))
    reps = np.array([1, 3, 4, 6, 2, 5])               # code/constants.py:50
    return dict(d2_idxs=d2_idxs, d3_idxs=d3_idxs, tasks=tasks,
                reps_train=reps[:-2], reps_test=reps[-2:])


# ----------------------------------------------------------------------------
# parameters: state_dict layout of code/models.py
# ----------------------------------------------------------------------------
LINEAR_IDX = (0, 3, 6, 9, 13, 17, 21)        # code/models.py:266-298
LINEAR_BN_IDX = (2, 5, 8, 11, 15, 19, 23)
DROPOUT_AFTER = (False, False, False, True, True, True, True)  # code/models.py:282-297


def _bn_prefix(base: str, adabn: bool) -> str:
    # AdaBatchNorm wraps nn.BatchNorm in an attribute called ``bn`` (models.py:22,32)
    return base + (".bn" if adabn else "")


def param_specs(d_e: int = 16, adabn: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered (key -> shape) of the reference ``Model.state_dict()`` parameters."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["logit_scale"] = ()
    s["emg_net.conv_emg.0.weight"] = (64, 1, 3, 3)
    s["emg_net.conv_emg.0.bias"] = (64,)
    s[_bn_prefix("emg_net.conv_emg.2", adabn) + ".weight"] = (64,)
    s[_bn_prefix("emg_net.conv_emg.2", adabn) + ".bias"] = (64,)
    s["emg_net.conv_emg.3.weight"] = (64, 64, 3, 3)
    s["emg_net.conv_emg.3.bias"] = (64,)
    s[_bn_prefix("emg_net.conv_emg.5", adabn) + ".weight"] = (64,)
    s[_bn_prefix("emg_net.conv_emg.5", adabn) + ".bias"] = (64,)
    fan_in = EMG_DIM * 64
    for li, bi in zip(LINEAR_IDX, LINEAR_BN_IDX):
        s[f"emg_net.linear.{li}.weight"] = (512, fan_in)
        s[f"emg_net.linear.{li}.bias"] = (512,)
        s[_bn_prefix(f"emg_net.linear.{bi}", adabn) + ".weight"] = (512,)
        s[_bn_prefix(f"emg_net.linear.{bi}", adabn) + ".bias"] = (512,)
        fan_in = 512
    s["emg_net.last.0.weight"] = (d_e, 512)
    s["glove_net.easy.0.weight"] = (d_e, N_TASKS)
    s["glove_net.easy.0.bias"] = (d_e,)
    s["glove_net.last.0.weight"] = (d_e, 256)
    return s


# ---- glove-angle class encoder (BASELINE config 3) -- PARITY UNPINNED ------------------
# The reference holds this encoder only as commented-out lines: `nn.Linear(GLOVE_DIM, 512//2, bias=False)`,
# `self.bn1d_func(512//2)`, `nn.ReLU()` inside GLOVENet.linear (code/models.py:388-390, after the Flatten at
# index 0) and `out=self.last(self.linear(out))` in GLOVENet.forward (code/models.py:461), with `self.last` =
# `nn.Linear(512//2, d_e, bias=False)` (code/models.py:425-428), which IS built and sits unused in every
# state_dict.  Nothing in the reference can run it, so there are no vectors to pin it to: this restatement
# un-comments those lines and is the definition the HIP path is checked against.
GLOVE_HIDDEN = 256
GLOVE_LINEAR_KEY = "glove_net.linear.1.weight"          # Sequential(Flatten, Linear, BN, ReLU): Linear is index 1


def glove_bn_base(adabn: bool) -> str:
    return _bn_prefix("glove_net.linear.2", adabn)


def add_glove_encoder(sd: "OrderedDict[str, torch.Tensor]", seed: int, adabn: bool) -> "OrderedDict[str, torch.Tensor]":
    """state_dict of the one-hot model -> state_dict with the glove encoder's tensors inserted where
    nn.Module.state_dict() would put them (glove_net.linear.* before glove_net.easy.*)."""
    g = torch.Generator().manual_seed(seed)
    bound = 1.0 / math.sqrt(GLOVE_DIM)
    extra: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    extra[GLOVE_LINEAR_KEY] = (torch.rand((GLOVE_HIDDEN, GLOVE_DIM), generator=g) * 2 - 1) * bound
    b = glove_bn_base(adabn)
    extra[b + ".weight"] = torch.ones(GLOVE_HIDDEN)
    extra[b + ".bias"] = torch.zeros(GLOVE_HIDDEN)
    if not adabn:
        extra[b + ".running_mean"] = torch.zeros(GLOVE_HIDDEN)
        extra[b + ".running_var"] = torch.ones(GLOVE_HIDDEN)
        extra[b + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        if k == "glove_net.easy.0.weight":
            out.update(extra)
        out[k] = v
    return out


def bn_bases(adabn: bool) -> List[str]:
    return ([_bn_prefix("emg_net.conv_emg.2", adabn), _bn_prefix("emg_net.conv_emg.5", adabn)]
            + [_bn_prefix(f"emg_net.linear.{bi}", adabn) for bi in LINEAR_BN_IDX])


def init_state_dict(seed: int, d_e: int = 16, adabn: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Seeded PyTorch-default-style init (kaiming-uniform a=sqrt(5) == U(+-1/sqrt(fan_in))).

    Mirrors what ``Model(...)`` produces at code/models.py:67-85 (same
    distributions, not the same random stream)."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    specs = param_specs(d_e, adabn)
    bns = set(bn_bases(adabn))
    for k, shp in specs.items():
        base = k.rsplit(".", 1)[0]
        if k == "logit_scale":
            sd[k] = torch.zeros(())          # ones * log(1) / 0.07  (models.py:81)
        elif base in bns:
            sd[k] = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        else:
            if k.endswith("weight"):
                fan_in = int(np.prod(shp[1:]))
            else:
                fan_in = int(np.prod(specs[base + ".weight"][1:]))
            bound = 1.0 / math.sqrt(fan_in)
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    if not adabn:
        for b in bn_bases(adabn):
            n = specs[b + ".weight"][0]
            sd[b + ".running_mean"] = torch.zeros(n)
            sd[b + ".running_var"] = torch.ones(n)
            sd[b + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    # re-order to the reference's state_dict order (buffers follow their BN's params)
    ordered: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k in specs:
        ordered[k] = sd[k]
        if not adabn and k.endswith(".bias") and k.rsplit(".", 1)[0] in bns:
            b = k.rsplit(".", 1)[0]
            for suffix in ("running_mean", "running_var", "num_batches_tracked"):
                ordered[b + "." + suffix] = sd[b + "." + suffix]
    return ordered


def trainable_keys(sd: Dict[str, torch.Tensor]) -> Tuple[List[str], List[str]]:
    """(emg_net keys, glove_net keys) handed to the two Adam optimisers (train.py:72-73).
    ``logit_scale`` belongs to neither."""
    emg = [k for k, v in sd.items() if k.startswith("emg_net.") and v.dtype.is_floating_point
           and not k.endswith(("running_mean", "running_var"))]
    glove = [k for k, v in sd.items() if k.startswith("glove_net.") and v.dtype.is_floating_point
             and not k.endswith(("running_mean", "running_var"))]
    return emg, glove


# ----------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------
class OracleModel:
    """Functional restatement of ``Model`` / ``EMGNet`` / ``GLOVENet``
    (code/models.py:66-228, 230-349, 352-472), contrastive mode only."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], params: Dict[str, float],
                 adabn: bool = False, requires_grad: bool = False, class_encoder: str = "onehot"):
        self.adabn = adabn
        self.class_encoder = class_encoder           # "onehot" (the reference as it runs) | "glove" (row f2, unpinned)
        self.params = dict(params)
        self.sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, v in state_dict.items():
            t = v.detach().clone()
            if requires_grad and t.dtype.is_floating_point and not k.endswith(
                    ("running_mean", "running_var")):
                t.requires_grad_(True)
            self.sd[k] = t
        self.training = True
        self.reset()

    # -- mode switches (models.py:87-104) -------------------------------------
    def set_train(self):
        self.training = True
        self.reset()

    def set_test(self):
        self.training = False
        self.reset()

    set_val = set_test

    def reset(self):
        self.corrects: List[float] = []
        self.voting: List[List[float]] = []
        self.y_pred: List[np.ndarray] = []
        self.y_true: List[np.ndarray] = []

    # -- batch norm (models.py:17-35 AdaBN; models.py:238-243 stock) ---------
    def _bn(self, x: torch.Tensor, base: str) -> torch.Tensor:
        w, b = self.sd[base + ".weight"], self.sd[base + ".bias"]
        if self.adabn:
            # momentum=0, track_running_stats=False: batch statistics always
            return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
        rm, rv = self.sd[base + ".running_mean"], self.sd[base + ".running_var"]
        if self.training:
            self.sd[base + ".num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, w, b, self.training, 0.1, BN_EPS)

    # -- EMG encoder (models.py:248-264, 266-298, 310-315, 319-342) ----------
    def encode_emg(self, EMG: torch.Tensor, taps: Optional[dict] = None,
                   dropout_masks: Optional[dict] = None, relu_masks: Optional[dict] = None) -> torch.Tensor:
        """``taps`` (test aid) receives r0..r8 = post-ReLU pre-BN outputs and the post-BN
        tensors; ``dropout_masks`` (test aid) {layer 5..8: keep-mask already scaled by
        1/(1-p)} replaces F.dropout so a device-generated mask can be replayed;
        ``relu_masks`` (test aid) {layer 0..8: bool mask in this tensor's layout} replaces
        relu(y) by y*mask, so that a pre-activation that two fp32 implementations round to
        opposite sides of zero does not make their gradients incomparable."""
        sd = self.sd

        def relu(y, layer):
            if taps is not None:
                taps[f"pre{layer}"] = y                      # pre-activation (test aid: how close to zero a flipped ReLU was)
            if relu_masks is not None and layer in relu_masks:
                return y * relu_masks[layer].to(y.dtype)
            return F.relu(y)

        self.shape = tuple(EMG.shape)
        bases = bn_bases(self.adabn)
        x = EMG.reshape(-1, 1, 1, EMG_DIM)
        x = relu(F.conv2d(x, sd["emg_net.conv_emg.0.weight"], sd["emg_net.conv_emg.0.bias"], padding=1), 0)
        if taps is not None:
            taps["r0"] = x
        x = self._bn(x, bases[0])
        if taps is not None:
            taps["bn1"] = x
        x = relu(F.conv2d(x, sd["emg_net.conv_emg.3.weight"], sd["emg_net.conv_emg.3.bias"], padding=1), 1)
        if taps is not None:
            taps["r1"] = x
        x = self._bn(x, bases[1])
        x = x.flatten(1)
        if taps is not None:
            taps["bn2"] = x
        dp = float(self.params.get("dp_emg", 0.0))
        for n, (li, drop) in enumerate(zip(LINEAR_IDX, DROPOUT_AFTER)):
            x = relu(F.linear(x, sd[f"emg_net.linear.{li}.weight"], sd[f"emg_net.linear.{li}.bias"]), n + 2)
            if taps is not None:
                taps[f"r{n + 2}"] = x
            x = self._bn(x, bases[2 + n])
            if drop:
                if dropout_masks is not None:
                    x = x * dropout_masks[n + 2]
                else:
                    x = F.dropout(x, dp, self.training)
            if taps is not None:
                taps[f"fc{n + 1}"] = x
        z = F.linear(x, sd["emg_net.last.0.weight"])
        if taps is not None:
            taps["z"] = z
        shape = self.shape
        d_e = z.shape[-1]
        out = z.reshape(shape[0], shape[1], shape[2], d_e).transpose(1, 2)
        return out.reshape(-1, shape[1], d_e)

    # -- class encoder, contrastive branch (models.py:447-465, 412-414) ------
    def encode_class(self, GLOVE: torch.Tensor, labels: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        sd = self.sd
        if self.class_encoder == "glove":
            # code/models.py:386-391,461 un-commented: Flatten -> Linear(20,256,no bias) -> BN -> ReLU -> last
            x = GLOVE.reshape(-1, GLOVE_DIM).to(sd[GLOVE_LINEAR_KEY].dtype)
            h = F.linear(x, sd[GLOVE_LINEAR_KEY])
            a = F.relu(self._bn(h, glove_bn_base(self.adabn)))
            out = F.linear(a, sd["glove_net.last.0.weight"])
            if taps is not None:
                taps["glove_h"], taps["glove_a"], taps["zg"] = h, a, out
        else:
            hot = F.one_hot(labels).to(sd["glove_net.easy.0.weight"].dtype)
            out = F.linear(hot, sd["glove_net.easy.0.weight"], sd["glove_net.easy.0.bias"])
        B, T = GLOVE.shape[0], GLOVE.shape[1]
        d_e = out.shape[-1]
        out = out.reshape(B, -1, d_e)
        if not self.training:
            out = out.reshape(B, 1, T, d_e).expand(-1, VOTE_SAMPLES, -1, -1).reshape(-1, T, d_e)
        return out

    # -- Model.forward (models.py:112-130) ------------------------------------
    def forward(self, EMG: torch.Tensor, GLOVE: torch.Tensor, labels: torch.Tensor,
                taps: Optional[dict] = None, dropout_masks: Optional[dict] = None,
                relu_masks: Optional[dict] = None) -> torch.Tensor:
        ze = self.encode_emg(EMG, taps, dropout_masks, relu_masks)
        ze = ze / ze.norm(dim=-1, keepdim=True)
        zc = self.encode_class(GLOVE, labels, taps)
        zc = zc / zc.norm(dim=-1, keepdim=True)
        return torch.bmm(ze, zc.transpose(1, 2))

    # -- loss, reference-faithful "loopy" form (models.py:132-173, 198-208) --
    def _loopy(self, logits: torch.Tensor, labels: torch.Tensor, acc: bool) -> torch.Tensor:
        loss = torch.zeros(1)
        correct = 0.0
        shape = self.shape
        vote = not self.training
        if vote:
            logits = logits.reshape(shape[0], shape[2], shape[1], shape[1])
            times = shape[2]
        else:
            times = 1
        bs, tasks = logits.shape[0], logits.shape[-1]
        target = torch.cat([labels[:tasks]] * times)
        for log in logits:
            loss = loss + F.cross_entropy(log.reshape(-1, tasks), target)
            if acc:
                pred = F.softmax(log, dim=-1).argmax(-1)
                if vote:
                    curve = []
                    for win in range(1, PREDICTION_WINDOW):
                        pred_ = pred[:win].mode(0)[0]
                        curve.append(float((pred_ == labels[:tasks]).numpy().mean()))
                    self.voting.append(curve)
                    self.y_pred.append(pred_.numpy())
                    self.y_true.append(labels[:tasks].numpy())
                    equal = curve[-1]
                else:
                    equal = float((pred == labels[:tasks]).numpy().mean())
                correct += equal
        loss = loss / bs
        if acc:
            self.corrects.append(float(correct / bs))
        return loss

    def loss(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        le = self._loopy(logits, labels, acc=True)
        lg = self._loopy(logits.transpose(1, 2), labels, acc=False)
        return (le + lg) / 2

    def loss_vectorized(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """Same value as ``loss`` in one shot."""
        T = logits.shape[-1]
        tgt = labels[:T].repeat(logits.shape[0])
        row = F.cross_entropy(logits.reshape(-1, T), tgt)
        col = F.cross_entropy(logits.transpose(1, 2).reshape(-1, T), tgt)
        return ((row + col) / 2).reshape(1)

    # -- global negatives: an EXTENSION, PARITY UNPINNED -- the reference has no such code.  It changes only the
    #    class->EMG direction of Model.loss (the second contrastive_loopy_loss call, models.py:203-206, whose softmax ranges over
    #    the 41 windows of one group): the column of class k of group b sees its positive window and every window of another
    #    class in the WHOLE batch handed in (all groups; under data parallelism: all ranks' groups).  With one group it
    #    reduces to the reference's loss, which is what pins it (tests/test_oracle_golden.py).
    def loss_global_negatives(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """logits (Bg,41,41) of the global batch in training mode (rows = windows, columns = class embeddings)."""
        T = logits.shape[-1]
        lab = labels[:T]
        tgt = lab.repeat(logits.shape[0])
        row = F.cross_entropy(logits.reshape(-1, T), tgt)
        pos_of_class = torch.empty(T, dtype=torch.long)
        pos_of_class[lab] = torch.arange(T)
        k = torch.arange(T)
        pos = logits[:, pos_of_class, k]                                    # (Bg, T): s[b, position of class k, k]
        neg = (lab.reshape(T, 1) != k.reshape(1, T)).to(logits.dtype)       # [window position i][class k]
        G = (logits.exp() * neg).sum(dim=(0, 1))                            # (T,)
        col = (-pos + torch.log(pos.exp() + G)).mean()
        return ((row + col) / 2).reshape(1)

    # -- L2 regulariser (models.py:225-228, 344-349, 467-472) ----------------
    def l2_keys(self) -> Tuple[List[str], List[str]]:
        def ok(k):
            return ("bn" not in k) and ("bias" not in k) and not k.endswith(
                ("running_mean", "running_var", "num_batches_tracked"))
        emg = [k for k in self.sd if k.startswith("emg_net.") and ok(k[len("emg_net."):])]
        glove = [k for k in self.sd if k.startswith("glove_net.") and ok(k[len("glove_net."):])]
        return emg, glove

    def l2(self) -> torch.Tensor:
        emg, glove = self.l2_keys()
        r_e = sum(torch.norm(self.sd[k]) for k in emg)
        r_g = sum(torch.norm(self.sd[k]) for k in glove)
        return r_g * self.params["reg_glove"] + r_e * self.params["reg_emg"]

    def correct(self) -> float:
        return float(np.array(self.corrects).mean())

    # -- one optimisation step (train.py:95-108) ------------------------------
    def make_optimizers(self):
        emg, glove = trainable_keys(self.sd)
        opt_e = torch.optim.Adam([self.sd[k] for k in emg], lr=self.params["lr_emg"], weight_decay=0)
        opt_g = torch.optim.Adam([self.sd[k] for k in glove], lr=self.params["lr_glove"], weight_decay=0)
        return opt_e, opt_g

    def train_step(self, EMG, GLOVE, labels, opts) -> float:
        logits = self.forward(EMG, GLOVE, labels)
        loss = self.loss(logits, labels)
        value = loss.item()
        loss = loss + self.l2()
        for o in opts:
            o.zero_grad(set_to_none=True)
        loss.backward()
        for o in opts:
            o.step()
        return value


# ----------------------------------------------------------------------------
# dataset half: DB23 / Glover / TaskWrapper index math on CPU tensors
# ----------------------------------------------------------------------------
class OracleDB23:
    """code/load.py:23-73,157-273 + Glover serving half (code/utils.py:248-254),
    on a resident CPU tensor ``EMG (41,46,6,100,12)`` (already transposed as at
    load.py:71) and ``GLOVE (41, D_g, 20)``."""

    def __init__(self, EMG: torch.Tensor, GLOVE: torch.Tensor, db2: bool = False):
        c = split_constants()
        self.EMG, self.GLOVE, self.db2 = EMG, GLOVE, db2
        self.tasks_mask = torch.from_numpy(np.concatenate((c["tasks"], [0])).astype(np.int64))
        self.people_mask = torch.from_numpy(
            (c["d2_idxs"] if db2 else c["d3_idxs"] + N_D2).astype(np.int64))
        tr, te = c["reps_train"], c["reps_test"]
        self.rep_train = torch.from_numpy(tr[:-1] - 1)
        self.rep_val = torch.from_numpy(tr[-1:] - 1)
        self.rep_test = torch.from_numpy(te - 1)
        self.train, self.val = True, False

    @property
    def rep_mask(self):
        if self.train:
            return torch.cat((self.rep_train, self.rep_test)) if self.db2 else self.rep_train
        if self.val:
            return self.rep_val
        return self.rep_val if self.db2 else self.rep_test

    TASKS = N_TASKS

    @property
    def PEOPLE(self):
        return len(self.people_mask)

    @property
    def REPS(self):
        return len(self.rep_mask)

    @property
    def OUTPUT_DIM(self):
        return WINDOW_OUTPUT_DIM if self.train else VOTE_SAMPLES

    @property
    def D(self):
        return self.PEOPLE * self.REPS * (WINDOW_OUTPUT_DIM if self.train else VOTE_RUNS)

    def set_mode(self, mode: str):
        self.train, self.val = mode == "train", mode == "val"
        t = self.EMG[self.tasks_mask][:, self.people_mask][:, :, self.rep_mask][:, :, :, :WINDOW_OUTPUT_DIM]
        self.EMG_use = t.reshape(-1, EMG_DIM)
        self.tensor = t.reshape(-1, self.OUTPUT_DIM, EMG_DIM)
        g = self.GLOVE[self.tasks_mask]
        self.D_g = self.GLOVE.shape[1]
        self.GLOVE_use = g.reshape(-1, GLOVE_DIM)

    def emg_item(self, idx: torch.Tensor) -> torch.Tensor:
        if not self.train:
            return self.tensor[idx].unsqueeze(2)                 # (41,25,1,12)
        return self.EMG_use[idx].reshape(-1, 1, 1, EMG_DIM)      # (41,1,1,12)


def make_rand_table(keys: torch.Tensor) -> torch.Tensor:
    """code/utils.py:34-36: per-class argsort of uniform keys, offset by class*D."""
    T, D = keys.shape
    return keys.argsort(dim=-1) + torch.arange(T, dtype=torch.long).reshape(T, 1) * D


def group_item(db: OracleDB23, emg_rand: torch.Tensor, glove_rand: torch.Tensor, idx: int):
    """code/utils.py:51-64."""
    emg = db.emg_item(emg_rand[:, idx]).to(torch.float32)
    glove = db.GLOVE_use[glove_rand[:, idx % db.D_g]].to(torch.float32)
    label = torch.arange(db.TASKS, dtype=torch.long)
    return emg, glove, label


def collate(db: OracleDB23, emg_rand, glove_rand, idxs) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """default_collate of B items (train.py:86,95)."""
    items = [group_item(db, emg_rand, glove_rand, int(i)) for i in idxs]
    return (torch.stack([i[0] for i in items]), torch.stack([i[1] for i in items]),
            torch.stack([i[2] for i in items]))


# ----------------------------------------------------------------------------
# synthetic Ninapro-shaped data
# ----------------------------------------------------------------------------
def synthetic_resident(seed: int = 1234, people: int = 46, glove_d: int = 5850):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(N_TASKS, EMG_DIM, generator=g)
    nu = torch.randn(people, EMG_DIM, generator=g)
    emg = (mu[:, None, None, None, :] + 0.5 * nu[None, :, None, None, :]
           + torch.randn(N_TASKS, people, 6, WINDOW_OUTPUT_DIM, EMG_DIM, generator=g))
    flat = emg.reshape(-1, EMG_DIM)
    emg = (emg - flat.mean(0)) / flat.std(0)
    gl = torch.randn(N_TASKS, 1, GLOVE_DIM, generator=g) + 0.3 * torch.randn(
        N_TASKS, glove_d, GLOVE_DIM, generator=g)
    return emg.contiguous(), gl.contiguous()
