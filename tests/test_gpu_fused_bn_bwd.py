"""bf16 backward with BatchNorm + ReLU backward applied in the data-gradient epilogue (the default where no dropout sits
between two layers) against the same step with the separate in-place pass (cp_config.options, CP_OPT_UNFUSED_BN_BWD).  The only arithmetic difference: the separate pass sees the incoming gradient rounded to bf16 first.
40,000 rows = 157 sample tiles (ragged last tile); dp = 0 fuses all seven fc layers and conv2, dp > 0 only fc1..fc3 and
conv2 (dropout follows fc4..fc7)."""
import pytest
import torch

from contrastiveprosthetics_amd import _lib

pytestmark = pytest.mark.gpu
T = 41


@pytest.mark.parametrize("dp", [0.0, 0.0635])
def test_bf16_fused_bn_backward_equals_separate_pass(dp):
    from contrastiveprosthetics_amd.engine import Engine
    n = 40000 - 40000 % T
    g = torch.Generator().manual_seed(3)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(n // T, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(n // T).cuda()
    grads = []
    for unfused in (False, True):
        e = Engine(adabn=False, dtype="bf16", dp_emg=dp, device="cuda", seed=123)
        e.init_parameters(5)
        gg = torch.Generator().manual_seed(9)
        for k in e.specs:                                    # non-trivial BN affine
            if ".bn" in k or k.split(".")[-2] in ("2", "5", "8", "11", "15", "19", "23"):
                v = e.values.views[k]
                if v.dim() == 1 and "linear" in k or "conv_emg.2" in k or "conv_emg.5" in k:
                    v.copy_((1.0 + 0.2 * torch.randn(v.shape, generator=gg) if k.endswith("weight") else 0.1 * torch.randn(v.shape, generator=gg)).cuda())
        e.grads.flat.zero_()
        e.options["unfused_bn_bwd"] = 1 if unfused else 0
        z = e.encoder_forward(x, training=True)
        e.head(z, labels, 1, want_grad=True)
        e.encoder_backward(x)
        torch.cuda.synchronize()
        assert torch.isfinite(e.grads.flat).all()
        grads.append({k: e.grads.views[k].clone() for k in e.specs})
    for k in grads[0]:
        a, b = grads[0][k].double().flatten(), grads[1][k].double().flatten()
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0, k
            continue
        cos = float(a @ b / (a.norm() * b.norm()))
        rel = float((a - b).norm() / b.norm())
        assert cos > 0.9995 and rel < 3e-2, (k, cos, rel)


def test_paired_weight_gradients_equal_unpaired():
    """bf16 with dropout: the weight gradients of fc7/fc6 and fc5/fc4 run as two problems of one launch with 32 splits
    each (api.hip, defer_wgrad) -- against one launch per layer with 64 splits (cp_config.options, CP_OPT_UNPAIRED_WGRAD).  Same products,
    f32 partial sums grouped differently."""
    from contrastiveprosthetics_amd.engine import Engine
    n = 40000 - 40000 % T
    g = torch.Generator().manual_seed(6)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(n // T, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(n // T).cuda()
    grads = []
    for unpaired in (False, True):
        e = Engine(adabn=False, dtype="bf16", dp_emg=0.0635, device="cuda", seed=123)
        e.init_parameters(8)
        e.grads.flat.zero_()
        e.options["unpaired_wgrad"] = 1 if unpaired else 0
        z = e.encoder_forward(x, training=True)
        e.head(z, labels, 1, want_grad=True)
        e.encoder_backward(x)
        torch.cuda.synchronize()
        grads.append({k: e.grads.views[k].clone() for k in e.specs})
    from contrastiveprosthetics_amd.engine import LINEAR_IDX
    behind_dropout = {f"emg_net.linear.{LINEAR_IDX[i]}.weight" for i in (4, 5, 6)} | {"emg_net.last.0.weight"}
    for k in grads[0]:
        a, b = grads[0][k].double().flatten(), grads[1][k].double().flatten()
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0, k
            continue
        rel = float((a - b).norm() / b.norm())
        # fc5..fc7: f32 sums of the same bf16 products, regrouped.  Below them the regrouped product of fc4 feeds fc3's
        # BatchNorm-backward sums (bn_bwd_sums_from_wgrad), so the gradient that flows on differs in its last bits.
        assert rel < (1e-5 if k in behind_dropout else 2e-3), (k, rel)
