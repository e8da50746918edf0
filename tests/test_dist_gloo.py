"""world_size-2 plumbing of contrastiveprosthetics_amd.dist on CPU with the gloo backend.

The HIP kernels cannot run here, so the per-rank compute is the CPU oracle; what is under test is
the sharding, the flat-gradient all-reduce + 1/world averaging, the parameter broadcast and the
z all-gather with "every rank scores its own slice" (SURVEY.md 8e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

T = 41
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(sd_like, keys):
    return torch.cat([sd_like[k].reshape(-1) for k in keys])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from contrastiveprosthetics_amd import dist as cpdist
    from oracle import ref_cpu as oc
    cpdist.init_from_env("gloo")
    assert (cpdist.rank(), cpdist.world_size()) == (rank, world)

    B = 6                                                    # global groups
    g = torch.Generator().manual_seed(5)
    EMG = torch.randn(B, T, 1, 1, 12, generator=g)
    s, e = cpdist.shard_range(B, rank, world)
    label = torch.arange(T).repeat(e - s)

    # broadcast: rank 1 starts from different weights and must end up with rank 0's
    sd = oc.init_state_dict(100 + rank, 16, adabn=True)
    keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    flat = _flat(sd, keys)
    cpdist.broadcast_(flat)
    ref0 = _flat(oc.init_state_dict(100, 16, adabn=True), keys)
    assert torch.equal(flat, ref0)
    sd = oc.init_state_dict(100, 16, adabn=True)

    m = oc.OracleModel(sd, BEST, adabn=True, requires_grad=True)
    taps = {}
    logits = m.forward(EMG[s:e], torch.zeros(e - s, T, 20), label, taps)
    loss = m.loss_vectorized(logits, label)
    loss.backward()
    gkeys = [k for k in keys if m.sd[k].grad is not None]
    gflat = _flat({k: m.sd[k].grad for k in gkeys}, gkeys).clone()
    local = gflat.clone()
    cpdist.all_reduce_sum_(gflat)
    # the two-bucket reducer (conv stack | everything from fc1's weight on) gives the same sums
    from collections import OrderedDict
    from types import SimpleNamespace
    offs, o = OrderedDict(), 0
    for k in gkeys:
        offs[k] = (o, m.sd[k].numel())
        o += m.sd[k].numel()
    eng = SimpleNamespace(grads=SimpleNamespace(flat=local.clone(), offsets=offs))
    red = cpdist.GradAllReduce(eng)
    assert 0 < red.split < local.numel() and not red.active
    assert torch.equal(red(), gflat)
    gflat /= world
    torch.save(dict(local=local, mean=gflat, loss=loss.detach()), os.path.join(out_dir, f"r{rank}.pt"))

    # z all-gather: the global matrix holds every rank's rows, and the own slice is bit-identical
    z = taps["z"].detach()
    Z = cpdist.all_gather_rows(z)
    assert Z.shape == (world * z.shape[0], 16)
    assert torch.equal(cpdist.local_rows(Z, z.shape[0]), z)
    zn = cpdist.local_rows(Z, z.shape[0]).reshape(e - s, T, 16)
    zn = zn / zn.norm(dim=-1, keepdim=True)
    zc = m.encode_class(torch.zeros(e - s, T, 20), label)
    zc = zc / zc.norm(dim=-1, keepdim=True)
    again = m.loss_vectorized(torch.bmm(zn, zc.transpose(1, 2)), label)
    assert torch.allclose(again, loss.detach(), rtol=0, atol=0)
    cpdist.shutdown()


@pytest.mark.timeout(300)
def test_two_rank_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(world)]
    assert torch.equal(r[0]["mean"], r[1]["mean"])                       # every rank holds the same averaged gradient
    np.testing.assert_allclose(r[0]["mean"].numpy(), ((r[0]["local"] + r[1]["local"]) / 2).numpy(), rtol=1e-6, atol=1e-9)
    assert not torch.equal(r[0]["local"], r[1]["local"])                 # the shards really differed
