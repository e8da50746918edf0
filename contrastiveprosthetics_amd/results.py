"""Test-set report with the surface of /root/reference/code/results.py (same flags, same output files), plus
the class-subset evaluation the reference's README describes (README.md:11-19) run on the device.

    python -m contrastiveprosthetics_amd.results --batch_size=8 --no_adabn --synthetic --save ../data/

`test(model, dataset, save)` follows code/results.py:24-64: one pass over the test split, raw logits to
`logs.npy`, the 250 ms majority-vote predictions to `y_pred.npy` / `y_true.npy`, the vote-length curve to
`voting.npy`, and the confusion matrix -- which the reference computes and prints but, by a slip, never
saves (it writes `voting.npy` twice, results.py:54,59); here it goes to `confusion_matrix.npy` in the
row-normalised form of the reference's published `data/confusion_matrix.npy`.

`subset_sweep` is the experiment behind the README figure ("average accuracy per prediction set size on 144
trials"): for every subset size k, `trials` random subsets of the 41 classes, rows and columns of every logits
tile restricted to the subset, argmax, 25-sample majority vote.  All subsets are scored by ONE kernel launch
(cp_subset_vote) over logits that never leave the GPU; the reference did this offline from logs.npy.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import engine as _engine
from .constants import MAX_TASKS
from .load import DB23
from .models import Model
from .utils import GroupLoader, TaskWrapper

N_TASKS = MAX_TASKS              # 41: 40 grasps + rest
args = None
shuff = True


def random_subsets(sizes, trials: int, seed: int = 0) -> np.ndarray:
    """(len(sizes)*trials, 41) uint8 membership masks, `trials` uniformly random subsets per size."""
    rng = np.random.default_rng(seed)
    out = np.zeros((len(sizes) * trials, N_TASKS), dtype=np.uint8)
    i = 0
    for k in sizes:
        for _ in range(trials):
            out[i, rng.choice(N_TASKS, size=int(k), replace=False)] = 1
            i += 1
    return out


def subset_accuracy(logits: torch.Tensor, labels: torch.Tensor, V: int, masks) -> np.ndarray:
    """Accuracy of every class subset for every vote length: (n_masks, V) float64.
    logits (B*V,41,41) on the device, group g = b*V + v; labels = labels[:41]; masks (n_masks,41)."""
    B = logits.shape[0] // V
    masks_t = torch.as_tensor(np.asarray(masks, dtype=np.uint8))
    correct = _engine.subset_vote(logits, labels[:N_TASKS].contiguous(), B, V, masks_t)
    k = masks_t.to(torch.float64).sum(1).clamp_min(1).numpy()
    return correct.cpu().numpy().astype(np.float64) / (B * k[:, None])


def subset_sweep(logits: torch.Tensor, labels: torch.Tensor, V: int, sizes=range(2, N_TASKS + 1), trials: int = 144,
                 seed: int = 0):
    """README figure: (sizes, mean, std, min, max) of the final-window accuracy over `trials` random subsets per
    size -- the columns of the reference's data/{mean,std,min,max}_grasp.xlsx."""
    sizes = list(sizes)
    acc = subset_accuracy(logits, labels, V, random_subsets(sizes, trials, seed))[:, -1].reshape(len(sizes), trials)
    return np.asarray(sizes), acc.mean(1), acc.std(1), acc.min(1), acc.max(1)


def test(model, dataset, save="../data/", sweep_trials: int = 0):
    """code/results.py:24-64"""
    dataset.set_test()
    model.set_test()
    total_loss, logs = [], []
    loader = GroupLoader(dataset, args.batch_size if args is not None else 8, shuffle=shuff)
    labels = None
    V = 1
    for (EMG, GLOVE, label) in loader:
        label = label.reshape(-1)
        V = EMG.shape[2]
        with torch.no_grad():
            logits = model.forward(EMG, GLOVE, label)          # (B*V, 41, 41)
            loss = model.loss(logits, label)
        total_loss.append(loss.detach().reshape(1))
        logs.append(logits)
        labels = label[:N_TASKS].to(torch.long).contiguous()
    logs_dev = torch.cat(logs).contiguous()
    os.makedirs(save, exist_ok=True)
    np.save(os.path.join(save, "logs.npy"), logs_dev.cpu().numpy())
    acc = model.correct()
    mean_loss = float(torch.cat(total_loss).mean().item())

    y_pred = model.y_pred_raw().flatten()
    y_true = model.y_true_raw().flatten()
    np.save(os.path.join(save, "y_pred.npy"), y_pred)
    np.save(os.path.join(save, "y_true.npy"), y_true)
    voting = model.voting_raw()
    np.save(os.path.join(save, "voting.npy"), voting)

    # confusion matrix of the 250 ms predictions, accumulated on the device
    yp = torch.cat(model.y_pred, 0).to(torch.int32).contiguous()
    counts = _engine.confusion(yp, labels).cpu().numpy()
    confusion_matrix = counts / np.maximum(counts.sum(1, keepdims=True), 1)
    np.save(os.path.join(save, "confusion_matrix.npy"), confusion_matrix)
    print(confusion_matrix, voting)

    if sweep_trials > 0:
        sizes, mean, std, lo, hi = subset_sweep(logs_dev, labels, V, trials=sweep_trials)
        np.save(os.path.join(save, "grasp_subsets.npy"), np.stack([sizes, mean, std, lo, hi], 1))
        for k, m, s in zip(sizes, mean, std):
            print(f"subset size {k:2d}: accuracy {m:.4f} +- {s:.4f}")
    return mean_loss, acc


def main(a):
    global args
    args = a
    dataset23 = DB23(db2=args.db2)
    print("Loading dataset")
    if args.synthetic:
        dataset23.load_synthetic()
    else:
        dataset23.load_stored()
    print("Dataset loaded")
    dataset23 = TaskWrapper(dataset23)

    # code/results.py:74-89: best model of the stored hyper-parameter search
    vpath, kpath = os.path.join(args.data_dir, "cross_val_values.npy"), os.path.join(args.data_dir, "cross_val_keys.npy")
    if os.path.exists(vpath) and os.path.exists(kpath):
        values, keys = np.load(vpath), np.load(kpath)
        best_key = keys[np.nanargmax(values[:, 1])]
    else:
        from .train import BEST_KEY
        best_key = BEST_KEY
    d_e, lr_e, reg_e, dp_e, lr_g, reg_g, dp_g = best_key
    k = 1 / 10 if args.load_model else 1
    params = {"d_e": int(d_e), "epochs": args.final_epochs, "lr_emg": lr_e * k, "dp_emg": dp_e, "reg_emg": reg_e,
              "lr_glove": lr_g * k, "dp_glove": dp_g, "reg_glove": reg_g}
    model = Model(params=params, train_model=True, adabn=args.no_adabn, prediction=args.prediction, glove=args.glove,
                  device="cuda", dtype=args.dtype).to(torch.float32)
    ckpt = os.path.join(args.checkpoint_dir, "contrastive.pt")
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, weights_only=True))
    else:
        print(f"no checkpoint at {ckpt}: reporting on the freshly initialised model")
    final_stats = test(model, dataset23, save=args.save, sweep_trials=args.subset_trials)
    print("loss,\t\t\tcorrect")
    print(final_stats)


def build_parser():
    parser = argparse.ArgumentParser(description="Test-set report on ninapro dataset")
    parser.add_argument("--crossval_size", type=int, default=10)
    parser.add_argument("--crossval_epochs", type=int, default=1)
    parser.add_argument("--batch_size", type=int, default=32)
    parser.add_argument("--final_epochs", type=int, default=10)
    parser.add_argument("--glove", action="store_true")
    parser.add_argument("--db2", action="store_true")
    parser.add_argument("--load_model", action="store_true")
    parser.add_argument("--crossval_load", action="store_true")
    parser.add_argument("--prediction", action="store_true")
    parser.add_argument("--no_adabn", action="store_false")
    parser.add_argument("--no_checkpoint", action="store_false")
    parser.add_argument("--no_verbose", action="store_false")
    parser.add_argument("--test", action="store_true")
    # additive
    parser.add_argument("--synthetic", action="store_true", help="seeded Ninapro-shaped tensors instead of emg.pt/glove.pt")
    parser.add_argument("--dtype", default="f32", choices=["f32", "bf16", "fp8"])
    parser.add_argument("--data_dir", default="../data")
    parser.add_argument("--checkpoint_dir", default="../checkpoints")
    parser.add_argument("--save", default="../data/")
    parser.add_argument("--subset_trials", type=int, default=0, help="random subsets per size for the README curve (144)")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
