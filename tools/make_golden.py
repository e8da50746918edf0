#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the *reference itself* on CPU.

Runs only in the build container (needs /root/reference).  The reference is
imported read-only (no bytecode written) with three inert stand-in modules for
debug/profiling imports that are not installed (``ipdb``, ``line_profiler``,
``pyxis`` -- none takes part in any computation), and its hard-coded
``torch.device("cuda")`` is redirected to CPU by rebinding the module-global
``torch`` of ``models``/``utils``/``load`` to a thin proxy.

Weights never travel: every fixture is generated from
``oracle.ref_cpu.init_state_dict(seed)`` loaded into the reference through
``load_state_dict`` (strict), so tests regenerate identical weights from the
seed.  Inputs are regenerated from seeds as well; fixtures hold the seeds and
the reference's OUTPUTS.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

REF = "/root/reference/code"
OUT = os.path.join(ROOT, "tests", "golden")

BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0,
            lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)   # data/cross_val_keys.npy[54], dp=0


def import_reference():
    for name in ("ipdb", "pyxis"):
        sys.modules.setdefault(name, types.ModuleType(name))
    lp = types.ModuleType("line_profiler")

    class LineProfiler:                      # inert: reference never decorates anything
        def print_stats(self, *a, **k):
            pass
    lp.LineProfiler = LineProfiler
    sys.modules.setdefault("line_profiler", lp)
    sys.path.insert(0, REF)

    class TorchProxy:
        def __init__(self, t):
            self._t = t

        def __getattr__(self, k):
            return getattr(self._t, k)

        def device(self, *a, **k):
            return self._t.device("cpu")

    import utils as rutils
    import models as rmodels
    import load as rload
    proxy = TorchProxy(torch)
    rutils.torch = proxy
    rmodels.torch = proxy
    rload.torch = proxy
    return rutils, rmodels, rload


def summarize(name, t, out, full_bytes=65536):
    a = t.detach().numpy().astype(np.float32)
    if a.nbytes <= full_bytes:
        out[name + "/full"] = a
    else:
        out[name + "/head"] = a.reshape(-1)[:256].copy()
    out[name + "/norm"] = np.float64(np.linalg.norm(a.astype(np.float64)))
    out[name + "/sum"] = np.float64(a.astype(np.float64).sum())


def seeded_emg(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def main():
    from oracle import ref_cpu as oc
    rutils, rmodels, rload = import_reference()
    os.makedirs(OUT, exist_ok=True)
    T = oc.N_TASKS

    def ref_model(seed, adabn, params=BEST):
        m = rmodels.Model(dict(params), adabn=adabn)
        sd = oc.init_state_dict(seed, params["d_e"], adabn)
        m.load_state_dict(sd, strict=True)
        return m, sd

    # ---- fwd + bwd, train mode, B=8 (BASELINE config 1 shape) -----------------
    for adabn in (False, True):
        tag = "adabn" if adabn else "stockbn"
        m, sd = ref_model(11, adabn)
        m.set_train()
        B = 8
        EMG = seeded_emg(101, (B, T, 1, 1, 12))
        GLOVE = seeded_emg(102, (B, T, 20))
        label = torch.arange(T).repeat(B)
        logits = m.forward(EMG, GLOVE, label)
        loss = m.loss(logits, label)
        l2 = m.l2()
        (loss + l2).backward()
        out = dict(weight_seed=11, emg_seed=101, glove_seed=102, B=B,
                   logits=logits.detach().numpy(), loss=loss.detach().numpy(),
                   l2=l2.detach().numpy(), acc=np.float64(m.corrects[0]),
                   argmax=logits.detach().argmax(-1).numpy().astype(np.int32))
        srt = torch.sort(logits.detach(), dim=-1, descending=True)[0]
        out["min_top2_margin"] = np.float64((srt[..., 0] - srt[..., 1]).min())
        for k, p in m.named_parameters():
            if p.grad is not None:
                summarize("grad/" + k, p.grad, out)
        if not adabn:
            for k, v in m.state_dict().items():
                if k.endswith(("running_mean", "running_var")):
                    out["buf/" + k] = v.numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"train_B8_{tag}.npz"), **out)
        print("train_B8", tag, "loss", float(loss), "l2", float(l2), "acc", m.corrects[0],
              "margin", out["min_top2_margin"])

    # ---- BN modes: 3 train steps worth of running stats, then eval ------------
    m, sd = ref_model(12, False)
    m.set_train()
    for s in range(3):
        EMG = seeded_emg(200 + s, (4, T, 1, 1, 12))
        m.forward(EMG, torch.zeros(4, T, 20), torch.arange(T).repeat(4))
    out = dict(weight_seed=12)
    for k, v in m.state_dict().items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            out["buf/" + k] = v.numpy().copy()
    m.set_test()
    EMG = seeded_emg(210, (2, T, 25, 1, 12))
    label = torch.arange(T).repeat(2)
    with torch.no_grad():
        logits = m.forward(EMG, torch.zeros(2, T, 20), label)
        loss = m.loss(logits, label)
    out.update(eval_logits=logits.numpy(), eval_loss=loss.numpy(),
               vote=np.array(m.voting)[:, :24], vote_cols=np.int64(np.array(m.voting).shape[1]),
               y_pred=np.array(m.y_pred).astype(np.int32), y_true=np.array(m.y_true).astype(np.int32),
               acc=np.float64(m.corrects[0]))
    np.savez_compressed(os.path.join(OUT, "bn_stock_3steps_eval_B2.npz"), **out)
    print("bn_stock eval loss", float(loss), "acc", m.corrects[0])

    # ---- eval + vote with AdaBN (batch statistics in eval) --------------------
    m, sd = ref_model(13, True)
    m.set_test()
    EMG = seeded_emg(220, (2, T, 25, 1, 12))
    with torch.no_grad():
        logits = m.forward(EMG, torch.zeros(2, T, 20), label)
        loss = m.loss(logits, label)
    np.savez_compressed(os.path.join(OUT, "eval_vote_B2_adabn.npz"), weight_seed=13, emg_seed=220,
                        eval_logits=logits.numpy(), eval_loss=loss.numpy(),
                        vote=np.array(m.voting)[:, :24], y_pred=np.array(m.y_pred).astype(np.int32),
                        acc=np.float64(m.corrects[0]))
    print("adabn eval loss", float(loss), "acc", m.corrects[0])

    # ---- three full optimisation steps (train.py:95-108), dp=0 ----------------
    for adabn in (False, True):
        tag = "adabn" if adabn else "stockbn"
        m, sd = ref_model(14, adabn)
        m = m.to(torch.float32)
        m.set_train()
        oe = torch.optim.Adam(m.emg_net.parameters(), lr=BEST["lr_emg"], weight_decay=0)
        og = torch.optim.Adam(m.glove_net.parameters(), lr=BEST["lr_glove"], weight_decay=0)
        losses = []
        for s in range(3):
            EMG = seeded_emg(300 + s, (8, T, 1, 1, 12))
            label = torch.arange(T).repeat(8)
            logits = m.forward(EMG, torch.zeros(8, T, 20), label)
            loss = m.loss(logits, label)
            losses.append(loss.item())
            loss = loss + m.l2()
            oe.zero_grad(set_to_none=True)
            og.zero_grad(set_to_none=True)
            loss.backward()
            oe.step()
            og.step()
        out = dict(weight_seed=14, losses=np.array(losses))
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                summarize("w/" + k, v, out, full_bytes=4096)
        np.savez_compressed(os.path.join(OUT, f"adam_3steps_{tag}.npz"), **out)
        print("adam_3steps", tag, losses)

    # ---- dataset API: masks, re-slice, sampler, item, collate -----------------
    EMGr, GLOVEr = oc.synthetic_resident(1234, glove_d=64)
    db = rload.DB23(db2=False)
    db.EMG = EMGr
    db.glover.GLOVE = GLOVEr
    tw = rutils.TaskWrapper(db)
    out = dict(resident_seed=1234, glove_d=64,
               tasks_mask=db.tasks_mask.numpy(), people_mask=db.people_mask.numpy(),
               rep_train=db.rep_train.numpy(), rep_val=db.rep_val.numpy(), rep_test=db.rep_test.numpy())
    probe = np.array([0, 1, 2, 99, 100, 1799, 1800, 3601, 40000, 73799])
    for mode in ("train", "val", "test"):
        torch.manual_seed(700 + len(mode))
        getattr(tw, "set_" + mode)()
        out[f"{mode}/D"] = np.int64(db.D)
        out[f"{mode}/len"] = np.int64(len(db))
        out[f"{mode}/EMG_use_shape"] = np.array(db.EMG_use.shape)
        out[f"{mode}/tensor_shape"] = np.array(db.tensor.shape)
        pr = probe[probe < db.EMG_use.shape[0]]
        out[f"{mode}/EMG_use_probe_idx"] = pr
        out[f"{mode}/EMG_use_probe"] = db.EMG_use[pr].numpy()
        out[f"{mode}/EMG_use_sum"] = np.float64(db.EMG_use.double().sum())
        pt = pr[pr < db.tensor.shape[0]]
        out[f"{mode}/tensor_probe_idx"] = pt
        out[f"{mode}/tensor_probe"] = db.tensor[pt].numpy()
        out[f"{mode}/rand_seed"] = np.int64(700 + len(mode))
        out[f"{mode}/emg_rand_head"] = tw.emg_rand[:, :32].numpy()
        out[f"{mode}/emg_rand_wsum"] = np.int64(
            (tw.emg_rand * (torch.arange(tw.emg_rand.shape[1]) + 1)).sum())
        out[f"{mode}/glove_rand_head"] = tw.glove_rand[:, :32].numpy()
        e, g, lab = tw[5]
        out[f"{mode}/item5_emg"] = e.numpy()
        out[f"{mode}/item5_glove"] = g.numpy()
        out[f"{mode}/item5_label"] = lab.numpy()
    np.savez_compressed(os.path.join(OUT, "db23_sampler.npz"), **out)
    print("db23: D train/val/test", out["train/D"], out["val/D"], out["test/D"])


if __name__ == "__main__":
    main()
