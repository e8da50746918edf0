"""SURVEY.md 8f row f2 / BASELINE config 3: the glove-angle class encoder (Linear(20->256, no bias) -> BN -> ReLU ->
Linear(256->16, no bias), per (group, class) rows) and the contrastive head with per-group class embeddings, through the
C ABI, against oracle/ref_cpu.py's un-commented restatement of code/models.py:386-391,461 ("parity unpinned": the
reference cannot run this branch, so the oracle is the definition; see its header)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as oc
from test_gpu_parity import BEST, T, device_relu_masks, randn

pytestmark = pytest.mark.gpu


def glove_state(seed, adabn):
    sd = oc.add_glove_encoder(oc.init_state_dict(seed, 16, adabn), seed + 1, adabn)
    g = torch.Generator().manual_seed(seed + 2)
    for b in oc.bn_bases(adabn) + [oc.glove_bn_base(adabn)]:                 # non-trivial BN affine
        sd[b + ".weight"] = 1.0 + 0.2 * torch.randn(sd[b + ".weight"].shape, generator=g)
        sd[b + ".bias"] = 0.1 * torch.randn(sd[b + ".bias"].shape, generator=g)
    return sd


def make_engine(sd, adabn, dtype):
    from contrastiveprosthetics_amd.engine import Engine
    e = Engine(adabn=adabn, dtype=dtype, dp_emg=0.0, device="cuda", class_encoder="glove")
    e.load_named(sd)
    return e


def step(e, EMG, GLOVE, label, train=True):
    x = EMG.reshape(-1, 12).cuda()
    e.grads.flat.zero_()
    z = e.encoder_forward(x, training=train)
    zg = e.glove_forward(GLOVE.cuda(), training=train)
    out, pred, logits = e.head_glove(z, zg, label.cuda(), EMG.shape[2], want_grad=train, want_logits=True)
    if train:
        e.encoder_backward(x)
        e.glove_backward()
    torch.cuda.synchronize()
    return zg, out, pred, logits


@pytest.mark.parametrize("adabn", [False, True])
@pytest.mark.parametrize("B", [7, 2])
def test_glove_encoder_f32_forward_backward(adabn, B):
    sd = glove_state(31, adabn)
    EMG, GLOVE = randn(501, (B, T, 1, 1, 12)), randn(502, (B, T, 20))
    label = torch.arange(T).repeat(B)
    e = make_engine(sd, adabn, "f32")
    assert [k for k in e.specs if k.startswith("glove_net")] == [k for k in sd if k.startswith("glove_net") and "running" not in k
                                                                 and "num_batches" not in k]
    zg, out, pred, logits = step(e, EMG, GLOVE, label)
    m = oc.OracleModel(sd, BEST, adabn=adabn, requires_grad=True, class_encoder="glove")
    taps = {}
    logits_ref = m.forward(EMG, GLOVE, label, taps, relu_masks=device_relu_masks(e))
    loss_ref = m.loss_vectorized(logits_ref, label)
    loss_ref.backward()
    np.testing.assert_allclose(zg.cpu().numpy(), taps["zg"].detach().numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(logits.cpu().numpy(), logits_ref.detach().numpy(), atol=2e-5, rtol=0)
    assert out[0].item() == pytest.approx(loss_ref.item(), rel=2e-6)
    assert np.array_equal(pred.cpu().numpy(), logits_ref.argmax(-1).numpy())                     # argmax bit-exact
    for k in e.specs:
        ref = m.sd[k].grad
        got = e.grads.views[k].cpu()
        if ref is None:                                                  # easy.* is unused by this class encoder
            assert float(got.abs().max()) == 0.0, k
            continue
        scale = float(ref.abs().max()) + 1e-12
        assert float((got - ref).abs().max()) / scale < 2e-4, (k, float((got - ref).abs().max()) / scale)
    if not adabn:
        gb = oc.glove_bn_base(False)
        rs = e.running_state()
        np.testing.assert_allclose(rs[gb + ".running_mean"].cpu().numpy(), m.sd[gb + ".running_mean"].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(rs[gb + ".running_var"].cpu().numpy(), m.sd[gb + ".running_var"].numpy(), rtol=1e-4, atol=1e-6)


def test_glove_eval_vote_expands_class_rows():
    """eval: (B,41,25,1,12) windows against (B,41,20) glove rows -- every one of the 25 samples of a group is scored
    against the same 41 class embeddings (code/models.py:463-464)."""
    adabn, B, V = True, 2, 25
    sd = glove_state(41, adabn)
    EMG, GLOVE = randn(601, (B, T, V, 1, 12)), randn(602, (B, T, 20))
    label = torch.arange(T).repeat(B)
    e = make_engine(sd, adabn, "f32")
    zg, out, pred, logits = step(e, EMG, GLOVE, label, train=False)
    m = oc.OracleModel(sd, BEST, adabn=adabn, class_encoder="glove")
    m.set_test()
    logits_ref = m.forward(EMG, GLOVE, label)
    assert tuple(logits.shape) == (B * V, T, T)
    np.testing.assert_allclose(logits.cpu().numpy(), logits_ref.numpy(), atol=3e-5, rtol=0)
    assert np.array_equal(pred.cpu().numpy(), logits_ref.argmax(-1).numpy())
    m.shape = (B, T, V)
    assert out[0].item() == pytest.approx(m.loss(logits_ref, label).item(), rel=3e-6)


def test_glove_encoder_bf16_and_full_batch():
    """bf16 storage at BASELINE config 3's batch (4096 groups = 167,936 windows and glove rows): finite, close to the
    f32 path in loss, gradients aligned."""
    from contrastiveprosthetics_amd.engine import Engine
    B = 4096
    g = torch.Generator().manual_seed(7)
    mu_e, mu_g = torch.randn(T, 12, generator=g), torch.randn(T, 20, generator=g)
    EMG = (mu_e[None] + torch.randn(B, T, 12, generator=g)).reshape(B, T, 1, 1, 12)
    GLOVE = mu_g[None] + 0.3 * torch.randn(B, T, 20, generator=g)
    label = torch.arange(T).repeat(B)
    res = {}
    for dt in ("f32", "bf16"):
        e = Engine(adabn=False, dtype=dt, dp_emg=0.0, device="cuda", class_encoder="glove")
        e.init_parameters(3)
        zg, out, pred, logits = step(e, EMG, GLOVE, label)
        assert torch.isfinite(e.grads.flat).all() and torch.isfinite(out).all()
        res[dt] = (float(out[0]), {k: e.grads.views[k].clone() for k in e.specs if k.startswith("glove_net.l")})
        del e
    assert res["bf16"][0] == pytest.approx(res["f32"][0], rel=5e-3)
    for k in res["f32"][1]:
        a, b = res["f32"][1][k].double().flatten(), res["bf16"][1][k].double().flatten()
        assert float(a @ b / (a.norm() * b.norm())) > 0.98, k


def test_glove_bench_config_vs_fp32_recompute():
    """BASELINE config 3 (glove-angle class encoder, 4096 groups, bf16) against a plain torch fp32 recomputation on the GPU, in the
    style of tests/test_gpu_fullsize.py: the class encoder's output from its own inputs and f32 weights, the head (per-group
    class rows) by autograd on the device's own z and zg, and the class encoder's three weight gradients by autograd through the
    recomputed encoder from the head's dL/dzg.  Bars: bf16 storage of the two intermediates (one rounding = 2^-9 relative; the
    BatchNorm in between re-amplifies it) -- 1.2e-2 of the tensor's max / 8e-3 of its rms, as for the sEMG encoder's layers;
    f32 arithmetic (loss) 2e-6."""
    from contrastiveprosthetics_amd.engine import Engine, GLOVE_LINEAR_KEY, glove_bn_base
    B = 4096
    N = B * T
    g = torch.Generator().manual_seed(7)
    mu_e, mu_g = torch.randn(T, 12, generator=g), torch.randn(T, 20, generator=g)
    EMG = (mu_e[None] + torch.randn(B, T, 12, generator=g)).reshape(B, T, 1, 1, 12)
    GLOVE = (mu_g[None] + 0.3 * torch.randn(B, T, 20, generator=g)).cuda()
    label = torch.arange(T).repeat(B)
    e = Engine(adabn=False, dtype="bf16", dp_emg=0.0, device="cuda", class_encoder="glove")
    e.init_parameters(3)
    gb = glove_bn_base(False)
    gg = torch.Generator().manual_seed(9)
    e.values.views[gb + ".weight"].copy_((1.0 + 0.2 * torch.randn(256, generator=gg)).cuda())      # non-trivial BN affine
    e.values.views[gb + ".bias"].copy_((0.1 * torch.randn(256, generator=gg)).cuda())
    x = EMG.reshape(-1, 12).cuda()
    e.grads.flat.zero_()
    z = e.encoder_forward(x, training=True)
    zg = e.glove_forward(GLOVE, training=True)
    out, pred, _ = e.head_glove(z, zg, label.cuda(), 1, want_grad=True)
    e.encoder_backward(x)
    e.glove_backward()
    torch.cuda.synchronize()

    def rel(got, ref):
        err = got.float() - ref.float()
        return (float(err.abs().max()) / (float(ref.abs().max()) + 1e-30), float(err.pow(2).mean().sqrt()) / (float(ref.pow(2).mean().sqrt()) + 1e-30))

    W1 = e.values.views[GLOVE_LINEAR_KEY].detach().clone().requires_grad_(True)                 # (256, 20)
    gam = e.values.views[gb + ".weight"].detach().clone().requires_grad_(True)
    bet = e.values.views[gb + ".bias"].detach().clone().requires_grad_(True)
    W2 = e.values.views["glove_net.last.0.weight"].detach().clone().requires_grad_(True)       # (16, 256)
    h = GLOVE.reshape(N, 20) @ W1.t()
    hn = (h - h.mean(0)) / torch.sqrt(h.var(0, unbiased=False) + 1e-5)
    zg_ref = torch.relu(hn * gam + bet) @ W2.t()
    a, b = rel(zg, zg_ref.detach())
    print(f"\nglove class encoder at {B} groups (bf16): zg max-err/max-ref {a:.2e}, rms-err/rms-ref {b:.2e}")
    assert a < 1.2e-2 and b < 8e-3, (a, b)
    # head by autograd on the device's own z and zg
    zt = z.detach().clone().requires_grad_(True)
    zgt = zg.detach().clone().requires_grad_(True)
    zn = (zt / zt.norm(dim=-1, keepdim=True)).reshape(B, T, 16)
    cn = (zgt / zgt.norm(dim=-1, keepdim=True)).reshape(B, T, 16)
    logits = zn @ cn.transpose(1, 2)
    tgt = torch.arange(T, device="cuda").repeat(B)
    loss = (torch.nn.functional.cross_entropy(logits.reshape(-1, T), tgt)
            + torch.nn.functional.cross_entropy(logits.transpose(1, 2).reshape(-1, T), tgt)) / 2
    loss.backward()
    assert out[0].item() == pytest.approx(loss.item(), rel=2e-6)
    agree = float((pred.reshape(-1).long() == logits.detach().argmax(-1).reshape(-1)).float().mean())
    assert agree > 0.9999, agree
    # the class encoder's weight gradients: autograd through the recomputed encoder from the head's dL/dzg
    zg_ref.backward(zgt.grad)
    G = e.grads.views
    worst = {}
    for name, ref in ((GLOVE_LINEAR_KEY, W1.grad), (gb + ".weight", gam.grad), (gb + ".bias", bet.grad), ("glove_net.last.0.weight", W2.grad)):
        err = float((G[name] - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        worst[name] = err
        # sums over 167,936 rows of products of bf16-rounded factors (dzg is stored in bf16, so are h and relu(BN(h))): the roundings
        # average out; 1e-2 of the tensor's max leaves room for the gradient's dependence on the bf16 forward values
        assert err < 1e-2, (name, err)
    print("  class-encoder weight gradients, max-err/max-ref:", {k: f"{v:.2e}" for k, v in worst.items()})
    # and the sEMG side of the head: the projection's weight gradient from the same dL/dz (stored in bf16 by the head kernel)
    assert torch.isfinite(e.grads.flat).all()


def test_model_api_glove_class_encoder(tmp_path):
    """The reference-shaped surface with class_encoder='glove': state_dict layout, a reference-style step (two torch
    Adams, loss + l2, autograd) against the oracle, checkpoint round trip, and the train CLI end to end."""
    from contrastiveprosthetics_amd.models import Model
    adabn = False
    sd = glove_state(51, adabn)
    model = Model(dict(BEST), adabn=adabn, device="cuda", dtype="f32", class_encoder="glove").to(torch.float32)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd, strict=True)
    model.set_train()
    opt_e = torch.optim.Adam(model.emg_net.parameters(), lr=BEST["lr_emg"], weight_decay=0)
    opt_g = torch.optim.Adam(model.glove_net.parameters(), lr=BEST["lr_glove"], weight_decay=0)
    o = oc.OracleModel(sd, BEST, adabn=adabn, requires_grad=True, class_encoder="glove")
    o.set_train()
    B = 6
    EMG, GLOVE = randn(701, (B, T, 1, 1, 12)), randn(702, (B, T, 20))
    label = torch.arange(T).repeat(B)
    ref_logits = o.forward(EMG, GLOVE, label)
    ref_loss, ref_l2 = o.loss(ref_logits, label), o.l2()
    (ref_loss + ref_l2).backward()
    logits = model.forward(EMG.cuda(), GLOVE.cuda(), label.cuda())
    loss = model.loss(logits, label.cuda())
    l2 = model.l2()
    assert loss.item() == pytest.approx(ref_loss.item(), rel=2e-6) and l2.item() == pytest.approx(ref_l2.item(), rel=2e-6)
    (loss + l2).backward()
    named = dict(model.named_parameters())
    for k in ("glove_net.linear.1.weight", "glove_net.linear.2.weight", "glove_net.linear.2.bias", "glove_net.last.0.weight",
              "emg_net.last.0.weight", "glove_net.easy.0.weight"):
        ref = o.sd[k].grad
        scale = float(ref.abs().max()) + 1e-12
        assert float((named[k].grad.cpu() - ref).abs().max()) / scale < 5e-3, k
    opt_e.step(); opt_g.step()
    assert model.correct() == pytest.approx(o.corrects[0], abs=1e-6)
    torch.save(model.state_dict(), tmp_path / "m.pt")
    again = Model(dict(BEST), adabn=adabn, device="cuda", dtype="f32", class_encoder="glove").to(torch.float32)
    again.load_state_dict(torch.load(tmp_path / "m.pt", weights_only=True))
    for k, v in model.state_dict().items():
        assert torch.equal(v, again.state_dict()[k]), k
    with pytest.raises(ValueError):
        model.forward(EMG.cuda(), None, label.cuda())

    from contrastiveprosthetics_amd import train
    a = train.build_parser().parse_args(["--crossval_load", "--final_epochs", "1", "--batch_size", "64", "--synthetic", "--test",
                                         "--class_encoder", "glove", "--dtype", "bf16", "--data_dir", str(tmp_path),
                                         "--checkpoint_dir", str(tmp_path)])
    train.main(a)
    assert (tmp_path / "contrastive.pt").exists()
    ck = torch.load(tmp_path / "contrastive.pt", weights_only=True)
    assert "glove_net.linear.1.weight" in ck and tuple(ck["glove_net.linear.1.weight"].shape) == (256, 20)
