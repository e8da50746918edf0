"""``Model`` -- the class surface of /root/reference/code/models.py:66-228 over the HIP engine.

Same constructor, methods, attributes and ``state_dict`` keys/shapes as the reference
(SURVEY.md 8b), contrastive mode only (``prediction=True`` is the reference's separate
softmax-classifier variant and is out of scope: it raises).  Parameters are ``nn.Parameter``
views into the engine's flat buffer, so ``state_dict()/load_state_dict()``, ``.to(torch.float32)``
and ``model.emg_net.parameters()`` behave as in the reference while the kernels see one buffer.

Two ways to run an optimisation step:

* reference-style (drop-in for code/train.py:95-108)::

      logits = model.forward(EMG, GLOVE, label); loss = model.loss(logits, label)
      loss = loss + model.l2(); opt.zero_grad(); loss.backward(); opt.step()   # torch.optim.Adam

  ``loss`` and ``l2`` are autograd nodes whose backward runs the HIP backward kernels.

* fused (what this package's train.py / bench.py use)::

      logits = model.forward(EMG, GLOVE, label); loss = model.loss(logits, label)
      model.backward(); model.optimizer_step()      # regulariser + both Adams in one launch set
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .constants import EMG_DIM, GLOVE_DIM, MAX_TASKS_TRAIN, PREDICTION_WINDOW, PREDICTION_WINDOW_SIZE, VOTE
from .engine import Engine, bn_bases, l2_member

__all__ = ["Model", "EMGNet", "GLOVENet"]


class _Node(nn.Module):
    """Anonymous container: the reference's nn.Sequential slots survive only as key names."""

    def child(self, name: str) -> "_Node":
        if name not in self._modules:
            self.add_module(name, _Node())
        return self._modules[name]


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        node = node.child(p) if isinstance(node, _Node) else _Node.child(node, p)
    if buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=True))


class EMGNet(_Node):
    """code/models.py:230-349.  ``forward`` returns (B*V, 41, d_e) like the reference; inside Model
    the regroup is folded into the head kernel's index map."""

    def l2(self):
        raise RuntimeError("use Model.l2(): the regulariser is one fused launch over both sub-nets")


class GLOVENet(_Node):
    """code/models.py:352-472, contrastive branch: Linear(41 -> d_e) on one-hot labels (`easy`);
    the unused `last` Linear(256 -> d_e) is kept because it is in the state_dict, the L2 term and Adam."""


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, loss_value, *params):
        ctx.model = model
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        m._run_backward()
        g = m.engine.grads
        scale = grad_out.reshape(())
        return (None, None) + tuple(g.views[k] * scale for k in m.engine.specs)


class _L2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, *params):
        ctx.model = model
        return model.engine.l2(model.params).reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        out = []
        for k in m.engine.specs:
            if l2_member(k):
                w = m.engine.values.views[k]
                reg = m.params["reg_glove"] if k.startswith("glove_net.") else m.params["reg_emg"]
                out.append(grad_out * reg * w / w.norm())
            else:
                out.append(None)
        return (None,) + tuple(out)


class Model(nn.Module):
    def __init__(self, params, adabn=True, train_model=True, prediction=False, glove=False, device="cuda",
                 dtype: str = "f32", seed: int = 42, class_encoder: str = "onehot", dropout_seed: int = None,
                 global_negatives: bool = False, sync_bn: bool = False):
        """class_encoder="glove" (additive): class embeddings come from the glove-angle rows that
        TaskWrapper already delivers, through the layers GLOVENet keeps as comments (code/models.py:386-391, 461),
        instead of the one-hot table.  (The reference's own `glove` flag belongs to its --prediction classifier.)
        seed initialises the parameters; dropout_seed (default: seed) keys the dropout stream -- data-parallel ranks
        pass seed + rank there so that their shards do not share one mask (parameters are broadcast anyway).
        global_negatives / sync_bn (additive, SURVEY 8e, both off by default = the reference's semantics at the local
        batch): training batches take the class->EMG direction of the loss over the z embeddings of the GLOBAL batch
        (global_negatives=True or "gather": one RCCL all-gather of z per step, cp_global_negatives / cp_head_gneg;
        "reduce": the same table from per-rank partial sums and two 64-float all-reduces, cp_global_negatives_g / _h, no z
        moves) / every BatchNorm of the sEMG encoder uses the statistics of the global batch (cp_config.stats_allreduce)."""
        super().__init__()
        if prediction or glove:
            raise NotImplementedError("only the contrastive mode (prediction=False, glove=False) is accelerated; "
                                      "the reference's --prediction / --glove branches are out of scope")
        self.params = params
        self.train_model = train_model
        self.adabn = adabn
        self.prediction = prediction
        self.glove = glove
        self.device = torch.device(device)
        self.class_encoder = class_encoder
        self.engine = Engine(adabn=adabn, dtype=dtype, dp_emg=float(params.get("dp_emg", 0.0)), device=device,
                             d_e=int(params["d_e"]), seed=seed if dropout_seed is None else dropout_seed,
                             class_encoder=class_encoder)
        self.engine.init_parameters(seed)
        self.global_negatives = bool(global_negatives)
        self.global_negatives_mode = global_negatives if isinstance(global_negatives, str) else "gather"
        if self.global_negatives_mode not in ("gather", "reduce"):
            raise ValueError("global_negatives: False, True, 'gather' or 'reduce'")
        self.sync_bn = bool(sync_bn)
        if self.global_negatives and class_encoder != "onehot":
            raise NotImplementedError("global negatives need the shared one-hot class table")
        from . import dist as cpdist
        if self.sync_bn and cpdist.world_size() > 1:
            self.engine.set_sync_bn(cpdist.all_reduce_sum_, cpdist.world_size())      # (per engine: carried in its cp_config)
        self.emg_net = EMGNet()
        self.glove_net = GLOVENet()
        for k in self.engine.specs:
            _attach(self, k, self.engine.values.views[k])
        for k, v in self.engine.running.items():
            _attach(self, k, v, buffer=True)
        # code/models.py:81: created, stored in the state_dict, never used, in no optimiser
        # (own parameters precede child modules in state_dict(), so it is the first key as in the reference)
        self.logit_scale = nn.Parameter(torch.zeros((), device=self.device))
        self.lr_scale = [1.0, 1.0]
        self.reset()
        self._last_logits = None
        self._pending = None

    def _apply(self, fn, recurse=True):
        # .to(torch.float32) (code/train.py:66) is a no-op here: parameters ARE f32 views of the flat buffer
        # and must stay views (a re-allocating cast would detach them from the kernels).
        probe = fn(torch.zeros(1, device=self.device))
        if probe.dtype != torch.float32 or probe.device.type != "cuda":
            raise RuntimeError("Model lives on the MI355X in float32 master weights; use dtype='bf16' for bf16 compute")
        return self

    # -- mode switches (code/models.py:87-104) ----------------------------------------------------
    def set_train(self):
        self.train_model = True
        self.train()
        self.reset()

    def set_test(self):
        self.train_model = False
        self.eval()
        self.reset()

    def set_val(self):
        self.set_test()

    def reset(self):
        self.corrects: List[torch.Tensor] = []
        self.voting: List[torch.Tensor] = []
        self.y_pred: List[torch.Tensor] = []
        self.y_true: List[torch.Tensor] = []

    # -- state ----------------------------------------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        self.engine.running_state()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                self.engine.num_batches_tracked = int(v)
        return out

    # -- forward (code/models.py:112-130) -----------------------------------------------------------
    def forward(self, EMG, GLOVE, labels):
        shape = tuple(EMG.shape)                       # (B,41,1,1,12) train | (B,41,25,1,12) eval-vote
        B, T, V = shape[0], shape[1], shape[2]
        x = EMG.reshape(-1, EMG_DIM)
        if x.dtype != torch.float32:
            x = x.to(torch.float32)
        x = x.contiguous()
        labels = labels.reshape(-1).to(torch.long)
        want_grad = self.training and torch.is_grad_enabled()
        z = self.engine.encoder_forward(x, training=self.training)
        if self.class_encoder == "glove":
            if GLOVE is None or tuple(GLOVE.shape[:2]) != (B, T):
                raise ValueError("class_encoder='glove' needs the (B,41,20) glove tensor TaskWrapper delivers")
            zg = self.engine.glove_forward(GLOVE, training=self.training)
            out, pred, logits = self.engine.head_glove(z, zg, labels, V, want_grad=want_grad, want_logits=True)
        else:
            gh = None
            if self.global_negatives and self.training and V == 1:
                # the global-batch z matrix: every rank's rows, rank-major (world 1: this rank's own z, no collective)
                from . import dist as cpdist
                if self.global_negatives_mode == "reduce":
                    gh = self.engine.global_negatives(z, labels, all_reduce=cpdist.all_reduce_sum_)
                else:
                    gh = self.engine.global_negatives(cpdist.all_gather_rows(z), labels)
            out, pred, logits = self.engine.head(z, labels, V, want_grad=want_grad, want_logits=True, gneg=gh)
        self._pending = dict(x=x, out=out, pred=pred, labels=labels, B=B, V=V, T=T, want_grad=want_grad, done=False)
        self._last_logits = logits
        return logits

    # -- loss (code/models.py:132-173, 198-208) -----------------------------------------------------
    def loss(self, logits, labels):
        if logits is not self._last_logits or self._pending is None:
            raise NotImplementedError("Model.loss expects the logits tensor returned by the last Model.forward call "
                                      "(the fused head kernel computed loss, argmax and gradients with them)")
        st = self._pending
        out, pred, B, V, T = st["out"], st["pred"], st["B"], st["V"], st["T"]
        vote = (not self.training) and VOTE
        if vote:
            curve, y_pred = self.engine.vote(pred, st["labels"], B, V)
            # the reference's range(1, PREDICTION_WINDOW) yields 249 columns; entries >= V repeat the last
            pad = curve[:, -1:].expand(-1, PREDICTION_WINDOW - 1 - V) if PREDICTION_WINDOW - 1 > V else curve[:, :0]
            self.voting.append(torch.cat((curve, pad), dim=1)[:, :PREDICTION_WINDOW - 1])
            self.y_pred.append(y_pred)
            self.y_true.append(st["labels"][:T].unsqueeze(0).expand(B, -1))
            self.corrects.append(curve[:, -1].mean())
        else:
            self.corrects.append(out[1] / float(B * V * T))
        value = out[0:1]
        if st["want_grad"]:
            return _LossFn.apply(self, value, *[self._parameters_by_key(k) for k in self.engine.specs])
        return value.clone()

    def _parameters_by_key(self, key: str) -> nn.Parameter:
        node = self
        parts = key.split(".")
        for p in parts[:-1]:
            node = node._modules[p]
        return node._parameters[parts[-1]]

    def _run_backward(self):
        st = self._pending
        if st is None or not st["want_grad"]:
            raise RuntimeError("backward without a training-mode forward")
        if not st["done"]:
            # the glove-angle class encoder first: data-parallel runs sum everything but the conv stack's gradients behind
            # an event that encoder_backward records (dist.GradAllReduce), and its gradients are in that bucket
            if self.class_encoder == "glove":
                self.engine.glove_backward()
            self.engine.encoder_backward(st["x"])
            st["done"] = True

    # -- fused step API ------------------------------------------------------------------------------
    def backward(self):
        """== (loss + self.l2()).backward() of code/train.py:101-105, with the regulariser's gradient deferred to
        optimizer_step() where it is applied inside the Adam kernel."""
        self._run_backward()

    def optimizer_step(self, grad_scale: float = 1.0):
        """optimizer_emg.step(); optimizer_glove.step() (code/train.py:107-108) + Model.l2()'s gradient."""
        return self.engine.adam_step(self.params, grad_scale=grad_scale, lr_scale=tuple(self.lr_scale))

    # -- regulariser (code/models.py:225-228) ----------------------------------------------------------
    def l2(self):
        if torch.is_grad_enabled():
            return _L2Fn.apply(self, *[self._parameters_by_key(k) for k in self.engine.specs])
        return self.engine.l2(self.params).reshape(())

    # -- metrics (code/models.py:210-223): one host read per call, none per step -------------------------
    def correct(self):
        return float(torch.stack([c.reshape(()) for c in self.corrects]).mean().item()) if self.corrects else float("nan")

    def correct_raw(self):
        return torch.stack([c.reshape(()) for c in self.corrects]).cpu().numpy()

    def voting_raw(self):
        return torch.cat(self.voting, 0).cpu().numpy()

    def y_pred_raw(self):
        return torch.cat(self.y_pred, 0).cpu().numpy()

    def y_true_raw(self):
        return torch.cat(self.y_true, 0).cpu().numpy()
