"""Host-side driver of the HIP kernels: owns the flat parameter / gradient / Adam
buffers and the workspace (as torch tensors -- PyTorch is only the allocator and
the stream provider here) and issues the C-ABI calls of include/cpnative.h.

Layout of the flat buffers: every trainable tensor of the reference ``Model``
(code/models.py, state_dict order, ``logit_scale`` excluded because neither
optimiser owns it, code/train.py:72-73), each aligned to 64 floats.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import CP_BF16, CP_D_E, CP_F32, CP_FP8, CP_N_BN, CP_N_FC, CP_TASKS

LINEAR_IDX = (0, 3, 6, 9, 13, 17, 21)           # code/models.py:266-298
LINEAR_BN_IDX = (2, 5, 8, 11, 15, 19, 23)


def bn_bases(adabn: bool):
    sfx = ".bn" if adabn else ""                # AdaBatchNorm wraps the BN as `.bn` (models.py:22,32)
    return (["emg_net.conv_emg.2" + sfx, "emg_net.conv_emg.5" + sfx]
            + [f"emg_net.linear.{i}{sfx}" for i in LINEAR_BN_IDX])


GLOVE_DIM, GLOVE_HIDDEN = 20, 256
GLOVE_LINEAR_KEY = "glove_net.linear.1.weight"        # Sequential(Flatten, Linear, BN, ReLU): the Linear is index 1


def glove_bn_base(adabn: bool) -> str:
    return "glove_net.linear.2" + (".bn" if adabn else "")


def param_specs(adabn: bool, d_e: int = CP_D_E, class_encoder: str = "onehot") -> "OrderedDict[str, Tuple[int, ...]]":
    """Trainable tensors in reference state_dict order, without logit_scale.
    class_encoder="glove" adds the layers GLOVENet keeps as comments (code/models.py:386-391), where
    nn.Module.state_dict() would list them: glove_net.linear.* before glove_net.easy.*."""
    bn = bn_bases(adabn)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["emg_net.conv_emg.0.weight"] = (64, 1, 3, 3)
    s["emg_net.conv_emg.0.bias"] = (64,)
    s[bn[0] + ".weight"] = (64,)
    s[bn[0] + ".bias"] = (64,)
    s["emg_net.conv_emg.3.weight"] = (64, 64, 3, 3)
    s["emg_net.conv_emg.3.bias"] = (64,)
    s[bn[1] + ".weight"] = (64,)
    s[bn[1] + ".bias"] = (64,)
    k = 768
    for n, li in enumerate(LINEAR_IDX):
        s[f"emg_net.linear.{li}.weight"] = (512, k)
        s[f"emg_net.linear.{li}.bias"] = (512,)
        s[bn[2 + n] + ".weight"] = (512,)
        s[bn[2 + n] + ".bias"] = (512,)
        k = 512
    s["emg_net.last.0.weight"] = (d_e, 512)
    if class_encoder == "glove":
        s[GLOVE_LINEAR_KEY] = (GLOVE_HIDDEN, GLOVE_DIM)
        s[glove_bn_base(adabn) + ".weight"] = (GLOVE_HIDDEN,)
        s[glove_bn_base(adabn) + ".bias"] = (GLOVE_HIDDEN,)
    s["glove_net.easy.0.weight"] = (d_e, CP_TASKS)
    s["glove_net.easy.0.bias"] = (d_e,)
    s["glove_net.last.0.weight"] = (d_e, 256)
    return s


def l2_member(name: str) -> bool:
    """code/models.py:344-349, 467-472: name (inside its sub-net) has neither 'bn' nor 'bias'."""
    local = name.split(".", 1)[1]
    return ("bn" not in local) and ("bias" not in local)


class FlatStore:
    """One contiguous f32 buffer with named views."""

    def __init__(self, specs, device, fill: Optional[float] = 0.0):
        self.offsets: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        off = 0
        for k, shp in specs.items():
            n = int(np.prod(shp))
            self.offsets[k] = (off, n)
            off += (n + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        if fill:
            self.flat.fill_(fill)
        self.views: Dict[str, torch.Tensor] = {
            k: self.flat[o:o + n].view(specs[k]) for k, (o, n) in self.offsets.items()}

    def ptr(self, name: str) -> int:
        return self.flat.data_ptr() + 4 * self.offsets[name][0]


def _params_struct(store: FlatStore, adabn: bool) -> _lib.cp_params:
    bn = bn_bases(adabn)
    p = _lib.cp_params()
    p.conv1_w = store.ptr("emg_net.conv_emg.0.weight")
    p.conv1_b = store.ptr("emg_net.conv_emg.0.bias")
    p.conv2_w = store.ptr("emg_net.conv_emg.3.weight")
    p.conv2_b = store.ptr("emg_net.conv_emg.3.bias")
    for n, li in enumerate(LINEAR_IDX):
        p.fc_w[n] = store.ptr(f"emg_net.linear.{li}.weight")
        p.fc_b[n] = store.ptr(f"emg_net.linear.{li}.bias")
    for n, b in enumerate(bn):
        p.bn_g[n] = store.ptr(b + ".weight")
        p.bn_b[n] = store.ptr(b + ".bias")
    p.last_w = store.ptr("emg_net.last.0.weight")
    p.easy_w = store.ptr("glove_net.easy.0.weight")
    p.easy_b = store.ptr("glove_net.easy.0.bias")
    return p


def gather_groups(table: torch.Tensor, emg_rand: torch.Tensor, perm: torch.Tensor, V: int) -> torch.Tensor:
    """cp_gather_groups: (B,41,V,12) f32 = table[(emg_rand[t, perm[b]] * V + v)]  (code/utils.py:51-64)."""
    lib = _lib.load()
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[1] == 12
    assert emg_rand.dtype == torch.int64 and emg_rand.is_contiguous() and perm.dtype == torch.int64
    B = perm.numel()
    out = torch.empty(B, CP_TASKS, V, 12, dtype=torch.float32, device=table.device)
    _lib.check(lib.cp_gather_groups(table.data_ptr(), table.shape[0], emg_rand.data_ptr(), emg_rand.shape[1],
                                    perm.contiguous().data_ptr(), B, V, out.data_ptr(),
                                    torch.cuda.current_stream(table.device).cuda_stream), "cp_gather_groups")
    return out


def gather_oob_count(device="cuda", reset: bool = True) -> int:
    """Source rows cp_gather_groups found outside the table since the last reset (0 in a healthy run); one host sync."""
    lib = _lib.load()
    dev = torch.device(device)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.cp_gather_oob_count(out.data_ptr(), 1 if reset else 0, torch.cuda.current_stream(dev).cuda_stream),
               "cp_gather_oob_count")
    return int(out.item()) & 0xFFFFFFFF


def subset_vote(logits: torch.Tensor, labels: torch.Tensor, B: int, V: int, masks: torch.Tensor, want_pred: bool = False):
    """cp_subset_vote: logits (B*V,41,41) f32, labels (41) int64, masks (n,41) uint8 ->
    correct (n,V) int64 [, y_pred (n,B,41) int32]  (README.md:11-19, code/models.py:146-163)."""
    lib = _lib.load()
    if logits.device.type != "cuda":
        raise _lib.CpNativeError("contrastiveprosthetics_amd runs on an MI355X (device 'cuda') only; no CPU path")
    assert logits.dtype == torch.float32 and logits.is_contiguous() and tuple(logits.shape) == (B * V, CP_TASKS, CP_TASKS)
    assert labels.dtype == torch.int64 and labels.numel() >= CP_TASKS and labels.is_contiguous()
    masks = masks.to(device=logits.device, dtype=torch.uint8).reshape(-1, CP_TASKS).contiguous()
    n = masks.shape[0]
    correct = torch.empty(n, V, dtype=torch.int64, device=logits.device)
    y_pred = torch.empty(n, B, CP_TASKS, dtype=torch.int32, device=logits.device) if want_pred else None
    _lib.check(lib.cp_subset_vote(logits.data_ptr(), labels.data_ptr(), B, V, masks.data_ptr(), n, correct.data_ptr(),
                                  y_pred.data_ptr() if want_pred else None,
                                  torch.cuda.current_stream(logits.device).cuda_stream), "cp_subset_vote")
    return (correct, y_pred) if want_pred else correct


def confusion(y_pred: torch.Tensor, labels: torch.Tensor, counts: torch.Tensor = None) -> torch.Tensor:
    """cp_confusion: counts (41,41) int64 += [labels[i % 41]][y_pred[i]]  (code/results.py:58)."""
    lib = _lib.load()
    if y_pred.device.type != "cuda":
        raise _lib.CpNativeError("contrastiveprosthetics_amd runs on an MI355X (device 'cuda') only; no CPU path")
    assert y_pred.dtype == torch.int32 and y_pred.is_contiguous() and y_pred.numel() % CP_TASKS == 0
    assert labels.dtype == torch.int64 and labels.numel() >= CP_TASKS and labels.is_contiguous()
    if counts is None:
        counts = torch.zeros(CP_TASKS, CP_TASKS, dtype=torch.int64, device=y_pred.device)
    _lib.check(lib.cp_confusion(y_pred.data_ptr(), labels.data_ptr(), y_pred.numel() // CP_TASKS, counts.data_ptr(),
                                torch.cuda.current_stream(y_pred.device).cuda_stream), "cp_confusion")
    return counts


class Engine:
    def __init__(self, adabn: bool = True, dtype: str = "bf16", dp_emg: float = 0.0, device="cuda",
                 d_e: int = CP_D_E, seed: int = 0, class_encoder: str = "onehot"):
        if class_encoder not in ("onehot", "glove"):
            raise ValueError("class_encoder must be 'onehot' or 'glove'")
        self.class_encoder = class_encoder
        if d_e != CP_D_E:
            raise ValueError(f"the HIP head kernel is built for d_e={CP_D_E} (code/train.py:183), got {d_e}")
        if dtype not in ("f32", "bf16", "fp8"):
            raise ValueError("dtype must be 'f32', 'bf16' or 'fp8'")
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.CpNativeError("contrastiveprosthetics_amd runs on an MI355X (device 'cuda') only; no CPU path")
        self.adabn = bool(adabn)
        self.dtype = {"f32": CP_F32, "bf16": CP_BF16, "fp8": CP_FP8}[dtype]
        self.dp_emg = float(dp_emg)
        self.seed = int(seed)
        self.step_count = 0              # forward passes in train mode (dropout stream)
        self.specs = param_specs(self.adabn, d_e, class_encoder)
        self.values = FlatStore(self.specs, self.device)
        self.grads = FlatStore(self.specs, self.device)
        self.exp_avg = torch.zeros_like(self.values.flat)
        self.exp_avg_sq = torch.zeros_like(self.values.flat)
        self.adam_steps = 0
        self.num_batches_tracked = 0
        self.running: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        if not self.adabn:
            for b in bn_bases(False) + ([glove_bn_base(False)] if class_encoder == "glove" else []):
                n = self.specs[b + ".weight"][0]
                self.running[b + ".running_mean"] = torch.zeros(n, device=self.device)
                self.running[b + ".running_var"] = torch.ones(n, device=self.device)
                self.running[b + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=self.device)
        self._p = _params_struct(self.values, self.adabn)
        self._g = _params_struct(self.grads, self.adabn)
        self._bn = _lib.cp_bn_buffers()
        if not self.adabn:
            for n, b in enumerate(bn_bases(False)):
                self._bn.running_mean[n] = self.running[b + ".running_mean"].data_ptr()
                self._bn.running_var[n] = self.running[b + ".running_var"].data_ptr()
        self._gp = self._gg = None
        if class_encoder == "glove":
            self._gp, self._gg = _lib.cp_glove_params(), _lib.cp_glove_params()
            gb = glove_bn_base(self.adabn)
            for st, store in ((self._gp, self.values), (self._gg, self.grads)):
                st.w1 = store.views[GLOVE_LINEAR_KEY].data_ptr()
                st.bn_g = store.views[gb + ".weight"].data_ptr()
                st.bn_b = store.views[gb + ".bias"].data_ptr()
                st.last_w = store.views["glove_net.last.0.weight"].data_ptr()
            if not self.adabn:
                self._gp.running_mean = self.running[gb + ".running_mean"].data_ptr()
                self._gp.running_var = self.running[gb + ".running_var"].data_ptr()
        self._gws: Optional[torch.Tensor] = None
        self._gws_rows = 0
        self._graph_state: Optional[torch.Tensor] = None      # device cp_step_state while a GraphStep owns the engine
        # per-engine call state, carried in every cp_config (the library has no process-wide switches)
        self.options: Dict[str, int] = {}                     # _lib.OPTIONS names -> 0/1 (tests, measurements)
        self.tile_schedule = _lib.CP_TILES_STATIC             # dist.use_dynamic_tiles() flips the default for packed sweeps
        from . import dist as _cpdist
        if _cpdist.default_tile_schedule() == "dynamic":
            self.tile_schedule = _lib.CP_TILES_DYNAMIC
        self._sync_cb = None
        self._sync_world = 1
        self._fp8_seen_forward = False
        # second stream for the weight gradients that nothing in the step waits for (cp_config.aux_stream; include/cpnative.h): created on
        # first use; None/False = one stream.  OFF by default since the projection's and conv2's weight gradients moved onto the critical
        # path (their products carry BatchNorm-backward sums): one stream measured 0.01-0.04 ms per step faster than two on the same box
        # (DESIGN 7i); $CPNATIVE_AUX_STREAM=1 or engine.aux_stream_enabled = True switches it on.
        import os as _os
        self.aux_stream_enabled = _os.environ.get("CPNATIVE_AUX_STREAM", "0") == "1"
        self._aux = None
        self.grad_tap: Optional[torch.Tensor] = None          # test aid (cp_config.grad_tap)
        self._ws: Optional[torch.Tensor] = None
        self._ws_windows = 0
        names = list(self.specs)
        self._tab_n = len(names)
        self._tab_off = (C.c_int64 * self._tab_n)(*[self.values.offsets[k][0] for k in names])
        self._tab_numel = (C.c_int64 * self._tab_n)(*[self.values.offsets[k][1] for k in names])
        self._tab_group = (C.c_int32 * self._tab_n)(*[1 if k.startswith("glove_net.") else 0 for k in names])
        self._tab_l2 = (C.c_int32 * self._tab_n)(*[1 if l2_member(k) else 0 for k in names])
        nscratch = self.lib.cp_optimizer_scratch_floats(self._tab_numel, self._tab_n)
        self._opt_scratch = torch.zeros(nscratch, device=self.device)
        self._l2_out = torch.zeros(1, device=self.device)

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _cfg(self, n_windows: int, training: bool) -> _lib.cp_config:
        c = _lib.cp_config()
        if self._graph_state is not None:          # capturing / replaying a step graph: per-step values come from device memory
            addr = self._graph_state.data_ptr()
            c.step_state_lo, c.step_state_hi = addr & 0xFFFFFFFF, addr >> 32
        c.n_windows = n_windows
        c.dtype = self.dtype
        c.adabn = 1 if self.adabn else 0
        c.training = 1 if training else 0
        c.dp_emg = self.dp_emg
        c.bn_momentum = 0.1
        c.bn_eps = 1e-5
        c.seed = self.seed
        c.step = self.step_count
        bits = 0
        for k, v in self.options.items():
            if v:
                bits |= _lib.OPTIONS[k]                      # KeyError = unknown option
        c.options = bits
        c.tile_schedule = self.tile_schedule
        if self._sync_cb is not None:
            c.stats_allreduce = C.cast(self._sync_cb, C.c_void_p)
            c.stats_world = self._sync_world
        if self.grad_tap is not None:
            c.grad_tap = self.grad_tap.data_ptr()
            c.grad_tap_bytes = self.grad_tap.numel() * self.grad_tap.element_size()
        # (not under data parallelism: the fc gradients' all-reduce starts behind an event that must follow every fc weight gradient, so the
        #  floating launches would be joined in the middle of the pass -- measured with RCCL in the loop at world size 1: 4.04 against 3.97 ms)
        if (self.aux_stream_enabled and training and self._graph_state is None and self.dp_emg > 0.0 and self.dtype != CP_F32
                and getattr(self, "fc_grads_ready", None) is None):
            if self._aux is None:
                with torch.cuda.device(self.device):
                    self._aux = (torch.cuda.Stream(device=self.device), torch.cuda.Event(), torch.cuda.Event())
                    for ev in self._aux[1:]:
                        ev.record()                          # (torch creates the hipEvent_t lazily, at the first record)
            c.aux_stream = self._aux[0].cuda_stream
            c.aux_fork = self._aux[1].cuda_event
            c.aux_join = self._aux[2].cuda_event
        return c

    def workspace(self, n_windows: int) -> torch.Tensor:
        """Scratch for up to n_windows rows; grows on demand.  A captured step graph has the address of the buffer it
        was captured with baked into every kernel node: GraphStep keeps its own reference to that tensor, so growing the
        engine's workspace later (an evaluation batch of 25x the rows) allocates a second buffer and leaves the
        graph's alone instead of handing it back to the caching allocator."""
        if self._ws is None or n_windows > self._ws_windows:
            nbytes = self.lib.cp_workspace_bytes(n_windows, self.dtype, self.dp_emg)
            # CP_FP8: the tensors' scales live in the first KiB of the workspace across steps (csrc/fp8.cuh, Fp8State).  RULE: the
            # table belongs to the engine, not to a buffer -- a grown workspace (the first validate() after training: 25x the
            # rows) inherits it, so evaluation starts from the scales training arrived at; only the engine's first workspace
            # starts from the defaults (zero-filled).  tests/test_gpu_fp8_product.py::test_scale_table_survives_reallocation.
            keep = self._ws[:_lib.FP8_STATE_BYTES].clone() if (self.dtype == CP_FP8 and self._ws is not None) else None
            self._ws = None                                   # (frees the old block first unless a GraphStep holds it)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            if self.dtype == CP_FP8:
                if keep is not None:
                    self._ws[:_lib.FP8_STATE_BYTES].copy_(keep)
                else:
                    self._ws[:_lib.FP8_STATE_BYTES].zero_()
            self._ws_windows = n_windows
        return self._ws

    def _ws_args(self, n_windows):
        # the carve depends on n_windows, so a call always passes the size for ITS n_windows
        ws = self.workspace(n_windows)
        return ws.data_ptr(), ws.numel()

    # ------------------------------------------------------------------ stages
    def gather(self, table: torch.Tensor, emg_rand: torch.Tensor, perm: torch.Tensor, V: int) -> torch.Tensor:
        B = perm.numel()
        out = torch.empty(B, CP_TASKS, V, 12, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.cp_gather_groups(table.data_ptr(), table.shape[0], emg_rand.data_ptr(), emg_rand.shape[1],
                                             perm.data_ptr(), B, V, out.data_ptr(), self._stream()), "cp_gather_groups")
        return out

    def encoder_forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        x = x.reshape(-1, 12)
        assert x.dtype == torch.float32 and x.is_contiguous() and x.device.type == "cuda"
        n = x.shape[0]
        if training:
            self.step_count += 1
        cfg = self._cfg(n, training)
        z = torch.empty(n, CP_D_E, dtype=torch.float32, device=self.device)
        ws, nb = self._ws_args(n)
        bn = C.byref(self._bn) if not self.adabn else None
        if self.dtype == CP_FP8 and not training and not self._fp8_seen_forward:
            # CP_FP8 scales are "delayed": a forward pass stores with the scales the PREVIOUS pass measured.  An engine whose first
            # pass ever is an evaluation (results.py on a loaded checkpoint) has only the defaults (2^4: clips at 28), so that
            # batch runs twice -- the first time to measure.  (Training needs no such pass: one clipped step among thousands.)
            _lib.check(self.lib.cp_encoder_forward(C.byref(cfg), C.byref(self._p), bn, x.data_ptr(), ws, nb, z.data_ptr(),
                                                   self._stream()), "cp_encoder_forward")
        self._fp8_seen_forward = True
        _lib.check(self.lib.cp_encoder_forward(C.byref(cfg), C.byref(self._p), bn, x.data_ptr(), ws, nb, z.data_ptr(),
                                               self._stream()), "cp_encoder_forward")
        if training and not self.adabn:
            self.num_batches_tracked += 1          # host counter; materialised by running_state()
        self._last = (n, training)
        self._last_x = x
        return z

    def running_state(self) -> "OrderedDict[str, torch.Tensor]":
        """BN buffers as the reference's state_dict holds them (running_mean/var/num_batches_tracked)."""
        for k, v in self.running.items():
            if k.endswith("num_batches_tracked"):
                v.fill_(self.num_batches_tracked)
        return self.running

    def head(self, z: torch.Tensor, labels: torch.Tensor, V: int, want_grad: bool, want_logits: bool = False,
             gneg: Optional[torch.Tensor] = None):
        """gneg: the {G, H} table of global_negatives() -> the column direction of the loss ranges over the global batch
        (cp_head_gneg); None = the reference's per-group loss (cp_head)."""
        n = z.shape[0]
        G = n // CP_TASKS
        assert labels.dtype == torch.int64 and labels.numel() * V == n
        cfg = self._cfg(n, self._last[1])
        out = torch.empty(2, dtype=torch.float32, device=self.device)
        pred = torch.empty(G, CP_TASKS, dtype=torch.int32, device=self.device)
        logits = torch.empty(G, CP_TASKS, CP_TASKS, dtype=torch.float32, device=self.device) if want_logits else None
        ws, nb = self._ws_args(n)
        if gneg is None:
            _lib.check(self.lib.cp_head(C.byref(cfg), C.byref(self._p), z.data_ptr(), labels.data_ptr(), G, V,
                                        1 if want_grad else 0, ws, nb, out.data_ptr(), pred.data_ptr(),
                                        logits.data_ptr() if want_logits else None, C.byref(self._g), self._stream()),
                       "cp_head")
        else:
            assert gneg.dtype == torch.float32 and gneg.numel() == 128 and gneg.is_cuda
            _lib.check(self.lib.cp_head_gneg(C.byref(cfg), C.byref(self._p), z.data_ptr(), labels.data_ptr(), G, V,
                                             1 if want_grad else 0, ws, nb, out.data_ptr(), pred.data_ptr(),
                                             logits.data_ptr() if want_logits else None, C.byref(self._g), gneg.data_ptr(),
                                             self._stream()), "cp_head_gneg")
        return out, pred, logits

    def global_negatives(self, z_all: torch.Tensor, labels: torch.Tensor, all_reduce=None) -> torch.Tensor:
        """The (2,64) table {G, H} that head(..., gneg=) takes.
        all_reduce=None (cp_global_negatives): z_all (n_all,16) f32 = the z rows of the GLOBAL batch (dist.all_gather_rows of every
        rank's encoder output; one rank: its own z).
        all_reduce=fn (cp_global_negatives_g / _h): z_all = THIS rank's rows only; fn(t) sums a 64-float device tensor over the ranks
        in place on the current stream (torch.distributed.all_reduce).  The class table is replicated, so G and H are sums of
        per-rank terms: two 256-byte collectives instead of the all-gather of z, same table."""
        assert z_all.dtype == torch.float32 and z_all.is_contiguous() and z_all.shape[1] == CP_D_E
        n_all = z_all.shape[0]
        need = self.lib.cp_global_negatives_scratch_floats(n_all)
        if getattr(self, "_gneg_scratch", None) is None or self._gneg_scratch.numel() < need:
            self._gneg_scratch = torch.empty(need, dtype=torch.float32, device=self.device)
        gh = torch.empty(2, 64, dtype=torch.float32, device=self.device)
        if all_reduce is None:
            _lib.check(self.lib.cp_global_negatives(C.byref(self._p), z_all.data_ptr(), n_all, labels.data_ptr(),
                                                    self._gneg_scratch.data_ptr(), gh.data_ptr(), self._stream()),
                       "cp_global_negatives")
            return gh
        _lib.check(self.lib.cp_global_negatives_g(C.byref(self._p), z_all.data_ptr(), n_all, labels.data_ptr(),
                                                  self._gneg_scratch.data_ptr(), gh.data_ptr(), self._stream()), "cp_global_negatives_g")
        all_reduce(gh[0])
        _lib.check(self.lib.cp_global_negatives_h(n_all, labels.data_ptr(), self._gneg_scratch.data_ptr(), gh.data_ptr(), self._stream()),
                   "cp_global_negatives_h")
        all_reduce(gh[1])
        return gh

    # ------------------------------------------------------------------ synchronised BatchNorm
    def set_sync_bn(self, allreduce=None, world: int = 1):
        """allreduce(tensor): sums a 1-D f32 device tensor over the ranks in place (e.g. torch.distributed.all_reduce),
        ordered on the current stream; None switches synchronised BatchNorm off.  While set, every BatchNorm of the sEMG
        encoder normalises with the statistics of the global batch (cp_config.stats_allreduce, 18 small collectives per
        training step).  Per engine: the hook travels in this engine's cp_config."""
        if allreduce is None:
            self._sync_cb = None
            self._sync_world = 1
            return
        if self.class_encoder == "glove":
            raise _lib.CpNativeError("synchronised BatchNorm covers the sEMG encoder; the glove-angle class encoder's BatchNorm stays local")
        if int(world) < 1:
            raise ValueError("world")
        eng = self

        def cb(user, row_ptr, count, stream):
            try:
                ws = eng._ws
                off = row_ptr - ws.data_ptr()
                if ws is None or off < 0 or off + 4 * count > ws.numel():
                    return 10003
                allreduce(ws[off:off + 4 * count].view(torch.float32))
                return 0
            except Exception as ex:                          # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return 10004

        self._sync_cb = _lib.ALLREDUCE_FN(cb)                # keep the trampoline alive
        self._sync_world = int(world)

    # ------------------------------------------------------------------ glove-angle class encoder (row f2)
    def _gws_args(self, rows: int):
        if self._gws is None or rows > self._gws_rows:
            self._gws = None
            self._gws = torch.empty(self.lib.cp_glove_workspace_bytes(rows, self.dtype), dtype=torch.uint8, device=self.device)
            self._gws_rows = rows
        # as with the main workspace the carve depends on `rows`: the buffer may be larger, never smaller
        return self._gws.data_ptr(), self._gws.numel()

    def glove_forward(self, glove: torch.Tensor, training: bool) -> torch.Tensor:
        """GLOVENet.forward, glove branch: glove (B,41,20) -> zg (B*41,16) f32."""
        if self._gp is None:
            raise _lib.CpNativeError("Engine was built with class_encoder='onehot'")
        x = glove.reshape(-1, GLOVE_DIM).to(torch.float32).contiguous()
        rows = x.shape[0]
        cfg = self._cfg(rows, training)
        zg = torch.empty(rows, CP_D_E, dtype=torch.float32, device=self.device)
        gws, nb = self._gws_args(rows)
        _lib.check(self.lib.cp_glove_forward(C.byref(cfg), C.byref(self._gp), x.data_ptr(), rows, gws, nb, zg.data_ptr(),
                                             self._stream()), "cp_glove_forward")
        self._last_glove_rows = rows
        return zg

    def head_glove(self, z: torch.Tensor, zg: torch.Tensor, labels: torch.Tensor, V: int, want_grad: bool,
                   want_logits: bool = False):
        n = z.shape[0]
        G = n // CP_TASKS
        assert labels.dtype == torch.int64 and labels.numel() * V == n and zg.shape[0] * V == n
        cfg = self._cfg(n, self._last[1])
        out = torch.empty(2, dtype=torch.float32, device=self.device)
        pred = torch.empty(G, CP_TASKS, dtype=torch.int32, device=self.device)
        logits = torch.empty(G, CP_TASKS, CP_TASKS, dtype=torch.float32, device=self.device) if want_logits else None
        ws, nb = self._ws_args(n)
        gws, gnb = self._gws_args(zg.shape[0])
        _lib.check(self.lib.cp_head_glove(C.byref(cfg), z.data_ptr(), zg.data_ptr(), labels.data_ptr(), G, V,
                                          1 if want_grad else 0, ws, nb, gws, gnb, out.data_ptr(), pred.data_ptr(),
                                          logits.data_ptr() if want_logits else None, self._stream()), "cp_head_glove")
        return out, pred, logits

    def glove_backward(self):
        rows = self._last_glove_rows
        cfg = self._cfg(rows, True)
        gws, nb = self._gws_args(rows)
        _lib.check(self.lib.cp_glove_backward(C.byref(cfg), C.byref(self._gp), rows, gws, nb, C.byref(self._gg),
                                              self._stream()), "cp_glove_backward")

    def encoder_backward(self, x: torch.Tensor):
        """self.fc_grads_ready (a recorded-once torch.cuda.Event, or None; dist.GradAllReduce sets it) is recorded on the
        stream when every gradient except the conv stack's is final (cp_encoder_backward_ev)."""
        x = x.reshape(-1, 12)
        n = x.shape[0]
        cfg = self._cfg(n, True)
        ws, nb = self._ws_args(n)
        ev = getattr(self, "fc_grads_ready", None)
        _lib.check(self.lib.cp_encoder_backward_ev(C.byref(cfg), C.byref(self._p), x.data_ptr(), ws, nb, C.byref(self._g),
                                                   self._stream(), ev.cuda_event if ev is not None else None),
                   "cp_encoder_backward")

    def vote(self, pred: torch.Tensor, labels: torch.Tensor, B: int, V: int):
        curve = torch.empty(B, V, dtype=torch.float32, device=self.device)
        y_pred = torch.empty(B, CP_TASKS, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.cp_vote(pred.data_ptr(), labels.data_ptr(), B, V, curve.data_ptr(), y_pred.data_ptr(),
                                    self._stream()), "cp_vote")
        return curve, y_pred

    def _hyper(self, params: dict, grad_scale: float = 1.0, lr_scale=(1.0, 1.0)) -> _lib.cp_adam_hyper:
        h = _lib.cp_adam_hyper()
        h.lr_emg = float(params.get("lr_emg", 0.0)) * lr_scale[0]
        h.lr_glove = float(params.get("lr_glove", 0.0)) * lr_scale[1]
        h.reg_emg = float(params["reg_emg"])
        h.reg_glove = float(params["reg_glove"])
        h.beta1, h.beta2, h.eps = 0.9, 0.999, 1e-8
        h.grad_scale = grad_scale
        return h

    def l2(self, params: dict) -> torch.Tensor:
        h = self._hyper(params)
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.cp_l2_norms(self.values.flat.data_ptr(), self._tab_off, self._tab_numel, self._tab_group,
                                        self._tab_l2, self._tab_n, C.byref(h), self._opt_scratch.data_ptr(),
                                        out.data_ptr(), self._stream()), "cp_l2_norms")
        return out

    def adam_step(self, params: dict, grad_scale: float = 1.0, lr_scale=(1.0, 1.0)) -> torch.Tensor:
        self.adam_steps += 1
        h = self._hyper(params, grad_scale, lr_scale)
        _lib.check(self.lib.cp_l2_adam_step(self.values.flat.data_ptr(), self.grads.flat.data_ptr(),
                                            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self._tab_off,
                                            self._tab_numel, self._tab_group, self._tab_l2, self._tab_n, C.byref(h),
                                            self.adam_steps, self._opt_scratch.data_ptr(), self._l2_out.data_ptr(),
                                            self._stream()), "cp_l2_adam_step")
        return self._l2_out

    def adam_step_graph(self, params: dict, grad_scale: float = 1.0) -> torch.Tensor:
        """cp_l2_adam_step with learning rates and bias corrections read from the device step state (graph capture)."""
        h = self._hyper(params, grad_scale)
        _lib.check(self.lib.cp_l2_adam_step_graph(self.values.flat.data_ptr(), self.grads.flat.data_ptr(),
                                                  self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self._tab_off,
                                                  self._tab_numel, self._tab_group, self._tab_l2, self._tab_n, C.byref(h),
                                                  self._graph_state.data_ptr(), self._opt_scratch.data_ptr(),
                                                  self._l2_out.data_ptr(), self._stream()), "cp_l2_adam_step_graph")
        return self._l2_out

    # ------------------------------------------------------------------ debug (tests)
    def debug_activation(self, layer: int) -> torch.Tensor:
        n, training = self._last
        C_ = 768 if layer < 2 else 512
        out = torch.empty(n, C_, dtype=torch.float32, device=self.device)
        cfg = self._cfg(n, training)
        ws, nb = self._ws_args(n)
        _lib.check(self.lib.cp_debug_activation(C.byref(cfg), C.byref(self._p), self._last_x.data_ptr(), ws, nb, layer,
                                                out.data_ptr(), self._stream()), "cp_debug_activation")
        return out

    def debug_bn_stats(self, layer: int) -> torch.Tensor:
        n, training = self._last
        C_ = 64 if layer < 2 else 512
        out = torch.empty(4, C_, dtype=torch.float32, device=self.device)
        cfg = self._cfg(n, training)
        ws, nb = self._ws_args(n)
        _lib.check(self.lib.cp_debug_bn_stats(C.byref(cfg), ws, nb, layer, out.data_ptr(), self._stream()),
                   "cp_debug_bn_stats")
        return out

    def fp8_scale_exponents(self) -> torch.Tensor:
        """CP_FP8: the scale table of the current workspace (csrc/fp8.cuh, Fp8State.e): stored = value * 2^e[t]; t = layer l for the
        saved activation of layer l (1..8), 9 + i for the dropout output feeding fc5 + i.  One host sync."""
        if self.dtype != CP_FP8 or self._ws is None:
            raise _lib.CpNativeError("no CP_FP8 workspace")
        return self._ws[:256].view(torch.int32).cpu()

    # ------------------------------------------------------------------ profiling (bench.py)
    def profile_enable(self, kinds=None, max_records: int = 4096):
        names = _lib.KERNEL_KINDS
        mask = 0
        for k in (kinds or names):
            mask |= 1 << names.index(k)
        _lib.check(self.lib.cp_profile_enable(mask, max_records), "cp_profile_enable")

    def profile_disable(self):
        self.lib.cp_profile_disable()

    def profile_resume(self):
        """Record again after profile_disable without dropping the records taken so far."""
        _lib.check(self.lib.cp_profile_resume(), "cp_profile_resume")

    def profile_summary(self) -> Dict[str, Tuple[float, int]]:
        """{kind: (total ms, launches)} of the records taken since profile_enable; sync first."""
        torch.cuda.synchronize(self.device)
        out = {}
        for i, name in enumerate(_lib.KERNEL_KINDS):
            ms, n = C.c_double(0), C.c_int64(0)
            _lib.check(self.lib.cp_profile_summary(i, C.byref(ms), C.byref(n)), "cp_profile_summary")
            if n.value:
                out[name] = (ms.value, n.value)
        return out

    # ------------------------------------------------------------------ state
    def init_parameters(self, seed: int):
        """PyTorch-default initialisation of code/models.py:67-85 (kaiming-uniform a=sqrt(5), i.e.
        U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases; BN gamma 1, beta 0)."""
        g = torch.Generator().manual_seed(seed)
        bns = set(bn_bases(self.adabn)) | {glove_bn_base(self.adabn)}
        for k, shp in self.specs.items():
            base = k.rsplit(".", 1)[0]
            if base in bns:
                self.values.views[k].fill_(1.0 if k.endswith("weight") else 0.0)
                continue
            wshape = shp if k.endswith("weight") else self.specs[base + ".weight"]
            bound = 1.0 / float(np.prod(wshape[1:])) ** 0.5
            self.values.views[k].copy_(((torch.rand(shp, generator=g) * 2 - 1) * bound).to(self.device))
        for k, v in self.running.items():
            if k.endswith("running_var"):
                v.fill_(1.0)
            else:
                v.zero_()

    def load_named(self, sd: Dict[str, torch.Tensor]):
        for k in self.specs:
            self.values.views[k].copy_(sd[k].to(self.device, torch.float32))
        for k in self.running:
            if k in sd:
                self.running[k].copy_(sd[k].to(self.device))
                if k.endswith("num_batches_tracked"):
                    self.num_batches_tracked = int(sd[k])


class GraphStep:
    """One whole training step -- gather, encoder forward, head, backward, L2 + Adam, ~110 kernel launches -- captured
    once in a HIP graph and replayed with a single launch.  At the reference's batch sizes (8..32 groups) a step is
    launch-latency-bound, so this is where its wall time goes.  What changes from step to step is not baked into
    the graph: the batch indices live in a fixed device buffer, and the dropout salt, Adam's bias corrections and
    the (scheduled) learning rates in a 32-byte device `cp_step_state` refreshed by one async copy per step.
    One instance per (engine, batch size); the class encoder is the one-hot table or, with `glove_table`, the glove one."""

    def __init__(self, engine: "Engine", table: torch.Tensor, emg_rand: torch.Tensor, batch: int, params: dict,
                 grad_scale: float = 1.0, glove=None):
        self.e = engine
        self.params = dict(params)
        self.B = int(batch)
        dev = engine.device
        self.perm = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.labels = torch.arange(CP_TASKS, device=dev).repeat(self.B)
        self.state = torch.zeros(8, dtype=torch.float32, device=dev)
        # per-step values travel through a RING of pinned slots: the host runs ahead of the GPU inside an epoch (no sync
        # per step), so one reused slot would be rewritten while earlier async copies from it are still queued and
        # those steps would read a later step's bias corrections and dropout salt.  A slot is reused only after the
        # event recorded behind its last copy has completed.
        self._host = torch.zeros(self.RING, 8, dtype=torch.float32).pin_memory()
        self._host_ev = [None] * self.RING
        self._slot = 0
        self.glove = glove                                  # callable perm -> (B,41,20) tensor, or None
        self.lr_scale = [1.0, 1.0]
        # the sampler table is re-drawn by TaskWrapper.reset() every epoch: the graph reads a copy at a fixed address
        self.table, self.emg_rand = table, emg_rand.clone()
        # the graph's own reference to the buffers whose addresses it bakes (see Engine.workspace)
        self._ws = engine.workspace(self.B * CP_TASKS)
        self._gws = None
        self._push()
        # the warm-up below is a real step on real state, and capturing runs the host side of every call once more:
        # snapshot what they touch and put it back
        keep = (engine.values.flat.clone(), engine.exp_avg.clone(), engine.exp_avg_sq.clone(),
                {k: v.clone() for k, v in engine.running.items()}, engine.step_count, engine.adam_steps,
                engine.num_batches_tracked)
        engine._graph_state = self.state
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                   # warm-up outside the capture (lazy module loads, allocator)
                self._body(grad_scale)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            # the warm-up was a real step on real state: roll the optimiser state back
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._body(grad_scale)
            self._gws = engine._gws                         # glove workspace captured with the graph, if any
        finally:
            engine._graph_state = None
            engine.values.flat.copy_(keep[0]); engine.exp_avg.copy_(keep[1]); engine.exp_avg_sq.copy_(keep[2])
            for k, v in keep[3].items():
                engine.running[k].copy_(v)
            engine.step_count, engine.adam_steps, engine.num_batches_tracked = keep[4], keep[5], keep[6]

    def _body(self, grad_scale):
        e = self.e
        x = e.gather(self.table, self.emg_rand, self.perm, 1)
        z = e.encoder_forward(x, training=True)
        if self.glove is not None:
            zg = e.glove_forward(self.glove(self.perm), training=True)
            out, pred, _ = e.head_glove(z, zg, self.labels, 1, want_grad=True)
            e.glove_backward()
            e.encoder_backward(x)
        else:
            out, pred, _ = e.head(z, self.labels, 1, want_grad=True)
            e.encoder_backward(x)
        e.adam_step_graph(self.params, grad_scale)
        return out

    RING = 64

    def _push(self):
        e = self.e
        t = e.adam_steps + 1
        i = self._slot
        self._slot = (i + 1) % self.RING
        if self._host_ev[i] is not None:
            self._host_ev[i].synchronize()                  # normally long done: 64 steps ago
        h = self._host[i]
        salt = (e.step_count + 1) * 0x9E3779B1 & 0xFFFFFFFF
        h.view(torch.int32)[0] = salt - (1 << 32) if salt >= (1 << 31) else salt
        b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))   # the betas as the C side holds them (float)
        h[1] = 1.0 - b1 ** t
        h[2] = 1.0 - b2 ** t
        h[3] = float(self.params.get("lr_emg", 0.0)) * self.lr_scale[0]
        h[4] = float(self.params.get("lr_glove", 0.0)) * self.lr_scale[1]
        self.state.copy_(h, non_blocking=True)
        if self._host_ev[i] is None:
            self._host_ev[i] = torch.cuda.Event()
        self._host_ev[i].record()

    def set_sampler(self, emg_rand: torch.Tensor):
        """TaskWrapper.reset() drew a new (41, D) table: refresh the graph's copy (same shape)."""
        self.emg_rand.copy_(emg_rand)

    def step(self, perm: torch.Tensor) -> torch.Tensor:
        """perm: the B item indices of this batch (DEVICE int64: a device-to-device copy is stream-ordered, a host
        tensor here would race with earlier replays).  Returns the device tensor (loss, #correct)."""
        assert perm.is_cuda, "GraphStep.step takes the batch indices as a device tensor"
        self.perm.copy_(perm, non_blocking=True)
        self._push()
        self.graph.replay()
        e = self.e
        e.step_count += 1
        e.adam_steps += 1
        if not e.adabn:
            e.num_batches_tracked += 1
        return self.out
