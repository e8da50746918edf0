"""A whole training step captured in a HIP graph (engine.GraphStep) against the same steps issued call by call:
with dropout off the two are the same kernels on the same numbers, so parameters, Adam moments and BN running
statistics must agree bit for bit after several steps; with dropout on the replayed step must draw a fresh mask
every step (the per-step salt lives in device memory, not in the graph)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = 41
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)


def data(D=500, seed=5):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(T, 1, 12, generator=g)
    table = (mu + torch.randn(T, D, 12, generator=g)).reshape(T * D, 12).cuda()
    emg_rand = (torch.rand(T, D, generator=g).argsort(-1) + torch.arange(T).reshape(T, 1) * D).cuda()
    perms = [torch.randperm(D, generator=g)[:8].cuda() for _ in range(6)]
    return table, emg_rand, perms


@pytest.mark.parametrize("dtype,adabn", [("f32", False), ("bf16", True)])
def test_graph_step_equals_eager_steps(dtype, adabn):
    from contrastiveprosthetics_amd.engine import Engine, GraphStep
    table, emg_rand, perms = data()
    labels = torch.arange(T).repeat(8).cuda()
    engines = []
    for mode in ("eager", "graph"):
        e = Engine(adabn=adabn, dtype=dtype, dp_emg=0.0, device="cuda", seed=3)
        e.init_parameters(11)
        losses = []
        if mode == "graph":
            gs = GraphStep(e, table, emg_rand, 8, BEST)
            for s, perm in enumerate(perms):
                if s == 3:
                    gs.lr_scale = [0.5, 0.25]                      # a scheduler step between replays
                losses.append(gs.step(perm)[0].item())
        else:
            for s, perm in enumerate(perms):
                x = e.gather(table, emg_rand, perm, 1)
                z = e.encoder_forward(x, training=True)
                out, _, _ = e.head(z, labels, 1, want_grad=True)
                e.encoder_backward(x)
                e.adam_step(BEST, lr_scale=(0.5, 0.25) if s >= 3 else (1.0, 1.0))
                losses.append(out[0].item())
        torch.cuda.synchronize()
        engines.append((e, losses))
    (a, la), (b, lb) = engines
    assert la == lb
    assert torch.equal(a.values.flat, b.values.flat)
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    assert (a.step_count, a.adam_steps, a.num_batches_tracked) == (b.step_count, b.adam_steps, b.num_batches_tracked)
    for k in a.running:
        assert torch.equal(a.running_state()[k], b.running_state()[k]), k
    assert la[-1] < la[0]


def test_graph_step_draws_a_new_dropout_mask_each_replay():
    from contrastiveprosthetics_amd.engine import Engine, GraphStep
    table, emg_rand, perms = data()
    e = Engine(adabn=True, dtype="f32", dp_emg=0.3, device="cuda", seed=3)
    e.init_parameters(11)
    gs = GraphStep(e, table, emg_rand, 8, dict(BEST, dp_emg=0.3, lr_emg=0.0, lr_glove=0.0, reg_emg=0.0, reg_glove=0.0))
    masks = []
    for _ in range(3):
        gs.step(perms[0])                                          # same batch, frozen weights: only the mask can change
        torch.cuda.synchronize()
        e._last = (8 * T, True)
        e._last_x = e.gather(table, emg_rand, perms[0], 1).reshape(-1, 12)
        masks.append(e.debug_activation(9 + 3) != 0)               # u of fc7's output: zero where dropped
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    keep = float(masks[0].float().mean())
    assert 0.2 < keep < 0.75                                       # (ReLU zeros and dropped entries both read as zero)


def test_graph_replays_without_host_syncs_match_eager():
    """The training loop does not synchronise inside an epoch, so the host queues many replays ahead of the GPU.  Each
    replay's bias corrections, learning rates and dropout salt travel through pinned memory: a single reused slot would
    be overwritten before the earlier copies ran and the first Adam steps would use bias corrections of
    later steps.  40 back-to-back replays, no host read in between, against the call-by-call steps: bit for bit."""
    from contrastiveprosthetics_amd.engine import Engine, GraphStep
    table, emg_rand, _ = data()
    g = torch.Generator().manual_seed(77)
    perms = [torch.randperm(500, generator=g)[:8].cuda() for _ in range(40)]
    labels = torch.arange(T).repeat(8).cuda()
    out = []
    for mode in ("eager", "graph"):
        e = Engine(adabn=True, dtype="bf16", dp_emg=0.0, device="cuda", seed=3)
        e.init_parameters(11)
        if mode == "graph":
            gs = GraphStep(e, table, emg_rand, 8, BEST)
            torch.cuda.synchronize()
            for perm in perms:
                gs.step(perm)
        else:
            for perm in perms:
                x = e.gather(table, emg_rand, perm, 1)
                z = e.encoder_forward(x, training=True)
                e.head(z, labels, 1, want_grad=True)
                e.encoder_backward(x)
                e.adam_step(BEST)
        torch.cuda.synchronize()
        out.append(e)
    a, b = out
    assert torch.equal(a.values.flat, b.values.flat)
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)


def test_graph_survives_a_workspace_that_grows_after_capture():
    """An evaluation batch (25x the rows) makes Engine.workspace allocate a larger buffer after the graph was captured
    with the old one's address in every kernel node: the graph must keep replaying on ITS buffer,
    which must stay allocated, while call-by-call work moves to the new one."""
    from contrastiveprosthetics_amd.engine import Engine, GraphStep
    table, emg_rand, perms = data()
    labels = torch.arange(T).repeat(8).cuda()
    res = []
    for grow in (False, True):
        e = Engine(adabn=False, dtype="f32", dp_emg=0.0, device="cuda", seed=3)
        e.init_parameters(11)
        gs = GraphStep(e, table, emg_rand, 8, BEST)
        ws0 = e._ws.data_ptr()
        for s, perm in enumerate(perms):
            gs.step(perm)
            if grow and s == 1:
                # an eval forward of 8*41*25 windows: grows the workspace, then scribble over fresh allocations so that a
                # replay through a freed block would read garbage
                xe = torch.randn(8 * T * 25, 12, device="cuda")
                e.encoder_forward(xe, training=False)
                assert e._ws.data_ptr() != ws0 and gs._ws.data_ptr() == ws0
                junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]
                del junk
        torch.cuda.synchronize()
        res.append(e.values.flat.clone())
    assert torch.isfinite(res[1]).all()
    assert torch.equal(res[0], res[1])


def test_train_cli_graph_two_epochs_equals_eager(tmp_path, capsys):
    """train.py --graph with the default verbose setting validates after every epoch (workspace growth between replays);
    two epochs must leave exactly the parameters of the call-by-call run (dropout off: same kernels on the same numbers)."""
    import numpy as np
    from contrastiveprosthetics_amd import train
    cks = []
    for mode in ("eager", "graph"):
        d = tmp_path / mode
        (d / "data").mkdir(parents=True)
        np.save(d / "data" / "cross_val_values.npy", np.array([[3.2, 0.25]]))
        np.save(d / "data" / "cross_val_keys.npy", np.array([[16, 9.761e-4, 7.103e-5, 0.0, 2.653e-3, 2.840e-6, 0.0]]))
        argv = ["--final_epochs=2", "--batch_size=16", "--crossval_load", "--synthetic", "--dtype", "bf16",
                "--data_dir", str(d / "data"), "--checkpoint_dir", str(d / "ckpt")]
        if mode == "graph":
            argv.append("--graph")
        train.main(train.build_parser().parse_args(argv))
        cks.append(torch.load(d / "ckpt" / "contrastive.pt", weights_only=True))
    out = capsys.readouterr().out
    assert out.count("Checkpointing model") >= 4
    a, b = cks
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
