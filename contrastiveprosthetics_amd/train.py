"""Training driver with the CLI surface of /root/reference/code/train.py:251-266 (same 13 flags, same
defaults and polarity -- note `--no_adabn`, `--no_checkpoint`, `--no_verbose` are store_false as in the
reference), plus additive flags for synthetic data, compute dtype and multi-GPU.

    python -m contrastiveprosthetics_amd.train --final_epochs=8 --crossval_size=150 --batch_size=8 \
        --crossval_load --test --no_adabn --synthetic            # the reference's go.sh recipe

Multi-GPU (one process per GPU, RCCL): launch with torch.distributed.run; groups are sharded over
ranks, gradients are all-reduced (sum) into the flat buffer and averaged inside the fused Adam step.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import dist as cpdist
from .load import DB23
from .engine import GraphStep
from .models import Model
from .utils import GroupLoader, TaskWrapper

# best row of the reference's stored hyper-parameter search (data/cross_val_keys.npy[54], val acc 0.2510)
BEST_KEY = [16, 9.761e-4, 7.103e-5, 0.0635, 2.653e-3, 2.840e-6, 0.3817]

args = None
shuff = True


def _evaluate(model, dataset, batch_size):
    total_loss = []
    loader = GroupLoader(dataset, batch_size, shuffle=shuff)
    for (EMG, GLOVE, label) in loader:
        label = label.reshape(-1)
        with torch.no_grad():
            logits = model.forward(EMG, GLOVE, label)
            loss = model.loss(logits, label)
            total_loss.append(loss.detach())
    acc = model.correct()
    mean_loss = float(torch.cat([l.reshape(1) for l in total_loss]).mean().item())
    return mean_loss, acc


def test(model, dataset):
    """code/train.py:27-44 (batch = 8 x batch_size)"""
    dataset.set_test()
    model.set_test()
    return _evaluate(model, dataset, args.batch_size * 8)


def validate(model, dataset):
    """code/train.py:46-63"""
    dataset.set_val()
    model.set_val()
    return _evaluate(model, dataset, args.batch_size)


def lr_scales(epoch: int, annealing: bool, final_epochs: int):
    """The reference's schedulers (code/train.py:75-80, stepped once per epoch at :112-113) as closed forms, applied as
    multipliers (emg, glove) to the fused optimiser's learning rates.  annealing: CosineAnnealingLR(T_max=final_epochs,
    eta_min=0) on both optimisers.  Otherwise the reference builds StepLR(step_size=5, gamma=.2) twice and BOTH wrap
    optimizer_glove (a quirk kept here): the glove rate falls by .2**2 every 5 epochs, the emg rate never moves.
    tests/test_cabi_and_host.py holds these to torch.optim.lr_scheduler."""
    if annealing:
        s = 0.5 * (1 + np.cos(np.pi * epoch / final_epochs))
        return [s, s]
    return [1.0, 0.2 ** (2 * (epoch // 5))]


def _sync_bn_buffers(model, world):
    """Before an evaluation under data parallelism: every rank takes rank 0's running statistics (training leaves them
    rank-local, so without this each rank would evaluate with different buffers)."""
    if world > 1 and model.engine.running:
        cpdist.broadcast_buffers_([v for k, v in model.engine.running.items() if v.dtype.is_floating_point])


def train_loop(dataset, params, checkpoint=False, checkpoint_dir="../checkpoints/model", annealing=False, load=None,
               verbose=False, solo=False):
    """code/train.py:65-138.  solo: this process trains the model alone even inside a multi-process job (packed sweep)."""
    world, rank = (1, 0) if solo else (cpdist.world_size(), cpdist.rank())
    # parameters start identical on every rank (seed 42, and rank 0's are broadcast below); the dropout stream is keyed
    # per rank so that the shards of one global batch do not share a mask
    model = Model(params=params, train_model=True, adabn=args.no_adabn, prediction=args.prediction, glove=args.glove,
                  device="cuda", dtype=args.dtype, class_encoder=getattr(args, "class_encoder", "onehot"),
                  dropout_seed=42 + rank,
                  global_negatives=(getattr(args, "global_negatives_mode", "gather") if getattr(args, "global_negatives", False) and not solo else False),
                  sync_bn=bool(getattr(args, "sync_bn", False)) and not solo).to(torch.float32)
    if load is not None:
        print("Loading model")
        model.load_state_dict(torch.load(load + ".pt", weights_only=True))
    if world > 1:
        cpdist.broadcast_(model.engine.values.flat)
        reduce_grads = cpdist.GradAllReduce(model.engine)     # two buckets, the large one beside the conv backward
    epochs = params["epochs"]
    dataset.set_train()
    model.set_train()
    gen = torch.Generator().manual_seed(42)
    loader = GroupLoader(dataset, args.batch_size, shuffle=shuff, rank=rank, world=world, generator=gen)
    val_losses = {}
    final_val_acc = None
    use_graph = (bool(getattr(args, "graph", False)) and world == 1 and model.class_encoder == "onehot"
                 and not model.global_negatives)
    graph_step = None
    print("Training...")
    for e in range(epochs):
        model.lr_scale = lr_scales(e, annealing, args.final_epochs)      # schedulers of code/train.py:75-80,112-113
        loss_train = []
        t0 = time.time()
        nwin = 0
        if use_graph:
            # the whole step as ONE graph launch (engine.GraphStep): at these batch sizes a step is ~110 launches of a few
            # microseconds each.  The loader's order is kept; a short last batch goes through the call-by-call path.
            if graph_step is None:
                graph_step = GraphStep(model.engine, dataset.EMG_use, dataset.emg_rand, args.batch_size, model.params)
            graph_step.set_sampler(dataset.emg_rand)
            graph_step.lr_scale = list(model.lr_scale)
            order = torch.randperm(len(dataset), generator=gen).to(model.device)
            full = (len(dataset) // args.batch_size) * args.batch_size
            for i in range(0, full, args.batch_size):
                out = graph_step.step(order[i:i + args.batch_size]).clone()
                loss_train.append(out[0:1])
                model.corrects.append(out[1] / float(args.batch_size * 41))
                nwin += args.batch_size * 41
            tail = [dataset.batch(order[full:])] if full < len(dataset) else []
        else:
            tail = loader
        for (EMG, GLOVE, label) in tail:
            label = label.reshape(-1)
            logits = model.forward(EMG, GLOVE, label)
            loss = model.loss(logits, label)
            loss_train.append(loss.detach())          # no host sync inside the step (reference: loss.item())
            model.backward()                          # == (loss + model.l2()).backward()
            if world > 1:
                reduce_grads()                        # conv-stack bucket now, the rest already under way
            model.optimizer_step(grad_scale=1.0 / world)
            nwin += label.numel()
        acc_train = model.correct()
        loss_train = float(torch.cat([l.reshape(1) for l in loss_train]).mean().item())
        dt = time.time() - t0
        if verbose:
            _sync_bn_buffers(model, world)
            loss_val, acc_val = validate(model, dataset)
            final_val_acc = (loss_val, acc_val)
            val_losses[e] = loss_val
            print("Epoch %d. Train loss: %.4f\tVal loss: %.4f\tVal acc: %.6f\tTrain acc: %.4f\t(%.0f windows/s)" %
                  (e, loss_train, loss_val, acc_val, acc_train, world * nwin / max(dt, 1e-9)))
        if checkpoint and verbose and rank == 0 and loss_val <= max(list(val_losses.values())):
            print("Checkpointing model...")
            os.makedirs(os.path.dirname(checkpoint_dir) or ".", exist_ok=True)
            torch.save(model.state_dict(), checkpoint_dir + ".pt")
        if checkpoint and verbose and world > 1:
            cpdist.barrier()              # rank 0's file is complete before any rank goes on (and, in the end, loads it)
        model.set_train()
        dataset.set_train()
    if not verbose:
        _sync_bn_buffers(model, world)
        loss_val, acc_val = validate(model, dataset)
        print("Epoch %d. Train loss: %.4f\tVal loss: %.4f\tVal acc: %.6f\tTrain acc: %.4f" %
              (epochs - 1, loss_train, loss_val, acc_val, acc_train))
        final_val_acc = (loss_val, acc_val)
    return final_val_acc, model


def cross_validate(des, hyperparams, dataset, id_, epochs=6, save=True, load=False, load_dir=None):
    """code/train.py:140-166.  The random search itself is host orchestration."""
    data_dir = args.data_dir
    vpath = os.path.join(data_dir, "cross_val_values%s.npy" % id_)
    kpath = os.path.join(data_dir, "cross_val_keys%s.npy" % id_)
    if load:
        if os.path.exists(vpath) and os.path.exists(kpath):
            return np.load(vpath), np.load(kpath)
        print("no stored search under %s: using the reference's published best row" % data_dir)
        return np.array([[3.2197, 0.2510]]), np.array([BEST_KEY])
    configs = [(d_e,) + tuple(hypervals) for d_e in des for hypervals in zip(*list(hyperparams.values()))]
    packed = bool(getattr(args, "hpo_pack", False))
    mine = cpdist.packed_indices(len(configs), cpdist.rank(), cpdist.world_size()) if packed else range(len(configs))
    done = {}
    for i in mine:
        if packed:
            torch.manual_seed(42 + i)      # a configuration's result must not depend on which rank ran it, or after what
        current = {k: v for k, v in zip(list(hyperparams.keys()), configs[i][1:])}
        print(current)
        params = {"d_e": configs[i][0], "epochs": epochs}
        params.update(current)
        (loss_t, acc_t), _ = train_loop(dataset, params, checkpoint=False, verbose=False, load=load_dir, solo=packed)
        done[i] = (loss_t, acc_t)
    results = cpdist.gather_packed(done, len(configs)) if packed else [done[i] for i in range(len(configs))]
    values, keys = np.array(results), np.array(configs)
    if save and cpdist.rank() == 0:
        os.makedirs(data_dir, exist_ok=True)
        np.save(vpath, values)
        np.save(kpath, keys)
    return values, keys


def main(a):
    global args
    args = a
    if args.hpo_pack:
        _, _, dev = cpdist.init_packed_from_env()
        torch.cuda.set_device(dev)
    else:
        cpdist.init_from_env()
    torch.manual_seed(42)                                # code/train.py:21 (torch.cuda.manual_seed(42); seeds the
    np.random.seed(42)                                   # sampler tables TaskWrapper.reset draws on the GPU), :22
    dataset23 = DB23(db2=args.db2)
    print("Loading dataset")
    if args.synthetic:
        dataset23.load_synthetic()
    else:
        dataset23.load_stored()
    print("Dataset loaded")
    dataset23 = TaskWrapper(dataset23)
    n = args.crossval_size
    hyperparams = {                                      # code/train.py:175-192
        "lr_emg": 10 ** np.random.uniform(low=-6, high=-1, size=(n,)),
        "reg_emg": 10 ** np.random.uniform(low=-9, high=-1, size=(n,)),
        "dp_emg": np.random.uniform(low=.4, high=.6, size=(n,)),
        "lr_glove": 10 ** np.random.uniform(low=-6, high=-1, size=(n,)),
        "reg_glove": 10 ** np.random.uniform(low=-9, high=-1, size=(n,)),
        "dp_glove": np.random.uniform(low=0, high=.9, size=(n,)),
    }
    values, keys = cross_validate([16], hyperparams, dataset23, id_="", epochs=args.crossval_epochs, save=True,
                                  load=args.crossval_load)
    best_key = keys[np.nanargmax(values[:, 1])]
    print("Best combination: %s" % str(best_key))
    if args.hpo_pack and cpdist.world_size() > 1:
        # the packed job exists for the search; the final model is trained by rank 0 alone
        if cpdist.rank() != 0:
            cpdist.shutdown()
            return
    solo_final = bool(args.hpo_pack)
    d_e, lr_e, reg_e, dp_e, lr_g, reg_g, dp_g = best_key
    k = 1 / 10 if args.load_model else 1
    params = {"d_e": int(d_e), "epochs": args.final_epochs, "lr_emg": lr_e * k, "dp_emg": dp_e, "reg_emg": reg_e,
              "lr_glove": lr_g * k, "dp_glove": dp_g, "reg_glove": reg_g}
    checkpoint_dir = os.path.join(args.checkpoint_dir, "contrastive")
    final_vals, model = train_loop(dataset23, params, checkpoint=args.no_checkpoint, annealing=True,
                                   checkpoint_dir=checkpoint_dir, verbose=args.no_verbose,
                                   load=checkpoint_dir if args.load_model else None, solo=solo_final)
    print("Final validation model statistics")
    print(final_vals)
    if not solo_final:
        cpdist.barrier()                  # nobody reads the checkpoint while rank 0 may still be writing it
    if os.path.exists(checkpoint_dir + ".pt"):
        model.load_state_dict(torch.load(checkpoint_dir + ".pt", weights_only=True))
    elif not solo_final:
        _sync_bn_buffers(model, cpdist.world_size())
    if args.test:
        final_stats = test(model, dataset23)
        print("loss,\t\t\tcorrect")
        print(final_stats)
    cpdist.shutdown()


def build_parser():
    parser = argparse.ArgumentParser(description="Training on ninapro dataset")
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
    parser.add_argument("--dtype", default="f32", choices=["f32", "bf16", "fp8"],
                        help="activation storage / matrix-instruction input type: f32 = the reference's precision (parity path), bf16, "
                             "fp8 = e4m3 activations + fc weights and e5m2 gradients on the block-scaled MFMA (BASELINE config 4; parity "
                             "unpinned by construction, tests/test_gpu_fp8_product.py holds its accuracy to the f32 path's)")
    parser.add_argument("--data_dir", default="../data")
    parser.add_argument("--checkpoint_dir", default="../checkpoints")
    parser.add_argument("--class_encoder", default="onehot", choices=["onehot", "glove"],
                        help="glove: class embeddings from the glove-angle rows (zero-shot path, BASELINE config 3)")
    parser.add_argument("--global_negatives", action="store_true",
                        help="extension: the class->EMG direction of the training loss ranges over the z embeddings of the GLOBAL "
                             "batch (one all-gather per step under data parallelism); off = the reference's per-group loss")
    parser.add_argument("--global_negatives_mode", default="gather", choices=["gather", "reduce"],
                        help="gather: the all-gather of z (BASELINE config 2); reduce: the same {G, H} table from per-rank partial sums "
                             "and two 64-float all-reduces (the class table is replicated: no z has to move)")
    parser.add_argument("--sync_bn", action="store_true",
                        help="extension: BatchNorm statistics over the global batch under data parallelism (18 small all-reduces "
                             "per step); off = every rank uses its shard's statistics, the reference at B_local")
    parser.add_argument("--graph", action="store_true",
                        help="replay each training step as one captured HIP graph (single process, one-hot class encoder)")
    parser.add_argument("--hpo_pack", action="store_true",
                        help="packed random search: every rank trains its share of the --crossval_size configurations alone "
                             "(any number of ranks per GPU, results gathered over gloo); rank 0 then trains the final model")
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
