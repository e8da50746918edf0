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


# ---- global negatives (SURVEY 8e extension): rank-sharded == single process on the concatenated batch ----------------
def _gneg_local_terms(z_local, z_all, E, labels, world):
    """What one rank computes, written the way the HIP kernels compute it (csrc/head.cuh: gneg_g_kernel, gneg_h_kernel,
    head_kernel with a.gneg): G and H from the GATHERED z, then loss and gradients of the rank's own windows only."""
    T = 41
    lab = labels[:T]
    En = E / E.norm(dim=-1, keepdim=True)
    zn_all = z_all / z_all.norm(dim=-1, keepdim=True)
    S_all = (zn_all @ En.t()).reshape(-1, T, T)                         # [group][position][class]
    neg = (lab.reshape(T, 1) != torch.arange(T).reshape(1, T)).double()
    pos_of_class = torch.empty(T, dtype=torch.long)
    pos_of_class[lab] = torch.arange(T)
    G = (S_all.exp() * neg).sum((0, 1))
    pos_all = S_all[:, pos_of_class, torch.arange(T)]
    H = (1.0 / (pos_all.exp() + G)).sum(0)
    zn = z_local / z_local.norm(dim=-1, keepdim=True)
    S = (zn @ En.t()).reshape(-1, T, T)
    B = S.shape[0]
    c = 1.0 / (2.0 * B * T)
    P_row = torch.softmax(S, -1)
    onehot = torch.nn.functional.one_hot(lab, T).double()               # [position][class]
    pos = S[:, pos_of_class, torch.arange(T)]
    den = pos.exp() + G
    loss = (-(torch.log_softmax(S, -1) * onehot).sum() + (torch.log(den) - pos).sum()) * c
    dl = P_row - onehot                                                 # row direction
    dl = dl + onehot * (S.exp() / den.reshape(B, 1, T) - 1.0) + (1 - onehot) * S.exp() * H
    dl = dl * c
    dzn = dl @ En                                                       # (B,T,16)
    dEn = torch.einsum("bik,bid->kd", dl, zn.reshape(B, T, 16))
    zl = z_local.reshape(B, T, 16)
    nz = zl.norm(dim=-1, keepdim=True)
    dz = (dzn - zn.reshape(B, T, 16) * (zn.reshape(B, T, 16) * dzn).sum(-1, keepdim=True)) / nz
    nE = E.norm(dim=-1, keepdim=True)
    dE = (dEn - En * (En * dEn).sum(-1, keepdim=True)) / nE
    return loss, dz.reshape(-1, 16), dE


def _gneg_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from contrastiveprosthetics_amd import dist as cpdist
    from oracle import ref_cpu as oc
    cpdist.init_from_env("gloo")
    Bg = 6
    g = torch.Generator().manual_seed(11)
    z_all_true = torch.randn(Bg * T, 16, generator=g, dtype=torch.float64)
    E = torch.randn(T, 16, generator=g, dtype=torch.float64)
    labels = torch.arange(T).repeat(Bg)
    s, e = cpdist.shard_range(Bg, rank, world)
    z_local = z_all_true[s * T:e * T].clone()
    gathered = cpdist.all_gather_rows(z_local)                          # the collective under test
    assert torch.equal(gathered, z_all_true)
    loss, dz, dE = _gneg_local_terms(z_local, gathered, E, labels, world)
    # the same table without moving z (cp_global_negatives_g / _h, engine.global_negatives(all_reduce=...)): the class table is
    # replicated, so G and H are sums of per-rank terms -- two all-reduces of 41 numbers
    lab = labels[:T]
    En = E / E.norm(dim=-1, keepdim=True)
    neg = (lab.reshape(T, 1) != torch.arange(T).reshape(1, T)).double()
    pos_of_class = torch.empty(T, dtype=torch.long)
    pos_of_class[lab] = torch.arange(T)

    def table(rows, G=None):
        S = ((rows / rows.norm(dim=-1, keepdim=True)) @ En.t()).reshape(-1, T, T)
        if G is None:
            return (S.exp() * neg).sum((0, 1))
        return (1.0 / (S[:, pos_of_class, torch.arange(T)].exp() + G)).sum(0)
    G_red = table(z_local)
    cpdist.all_reduce_sum_(G_red)
    H_red = table(z_local, G_red)
    cpdist.all_reduce_sum_(H_red)
    np.testing.assert_allclose(G_red.numpy(), table(gathered).numpy(), rtol=1e-12)
    np.testing.assert_allclose(H_red.numpy(), table(gathered, table(gathered)).numpy(), rtol=1e-12)
    # single-process definition on the concatenated batch (oracle, autograd)
    za = z_all_true.clone().requires_grad_(True)
    Ea = E.clone().requires_grad_(True)
    logits = ((za / za.norm(dim=-1, keepdim=True)) @ (Ea / Ea.norm(dim=-1, keepdim=True)).t()).reshape(Bg, T, T)
    m = oc.OracleModel(oc.init_state_dict(0, 16, True), BEST, adabn=True)
    ref = m.loss_global_negatives(logits, labels)
    ref.backward()
    # this rank's windows: the kernel gradient, averaged over ranks by the optimiser (grad_scale = 1/world), is the global one
    np.testing.assert_allclose((dz / world).numpy(), za.grad[s * T:e * T].numpy(), rtol=1e-9, atol=1e-12)
    tot = torch.stack([loss.reshape(()), torch.zeros((), dtype=torch.float64)])
    cpdist.all_reduce_sum_(tot)
    assert tot[0].item() / world == pytest.approx(ref.item(), rel=1e-12)
    dEs = dE.clone()
    cpdist.all_reduce_sum_(dEs)
    np.testing.assert_allclose((dEs / world).numpy(), Ea.grad.numpy(), rtol=1e-9, atol=1e-12)
    open(os.path.join(out_dir, f"gneg_ok{rank}"), "w").write("ok")
    cpdist.shutdown()


def test_global_negatives_sharded_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_gneg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"gneg_ok{r}") for r in range(2))
