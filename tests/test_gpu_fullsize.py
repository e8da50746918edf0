"""BASELINE config[1] size (4096 groups = 167,936 windows) through size-independent properties.

The oracle cannot run at this size in seconds, so the checks are:
  * bf16: every fc layer's stored output equals relu(BN(prev) W^T + b) recomputed independently with
    plain torch fp32 matmuls on the GPU from the PREVIOUS stored activation and its BN statistics
    (exercises all 656 row tiles x 2 feature tiles of the 256x256 LDS-DMA GEMM, the fold and the fc1
    column permutation); the stored BN statistics equal torch's mean/var of the stored activation;
  * f32: the directional derivative <dL/dtheta, v> predicted by the HIP backward pass equals the
    central difference (L(theta + eps v) - L(theta - eps v)) / 2 eps of the HIP forward pass (train-mode
    BatchNorm included), for a random direction over ALL parameters;
  * the loss of a freshly initialised model sits at the reference's structural floor/ceiling
    (cosine logits in [-1,1] => CE between log(1+40/e^2) and log(1+40 e^2)), and the mean over two
    half batches run separately brackets the full-batch loss.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T, B = 41, 4096
N = T * B
BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)


def synthetic(seed=3):
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None, :, :] + torch.randn(B, T, 12, generator=g)).reshape(N, 12).cuda()
    labels = torch.arange(T).repeat(B).cuda()
    return x, labels


def engine(dtype, seed=5):
    from contrastiveprosthetics_amd.engine import Engine
    e = Engine(adabn=True, dtype=dtype, dp_emg=0.0, device="cuda")
    e.init_parameters(seed)
    # non-trivial BN affine so the fold matters
    g = torch.Generator().manual_seed(9)
    for k in e.specs:
        if ".bn." in k:
            v = e.values.views[k]
            v.copy_((1.0 + 0.2 * torch.randn(v.shape, generator=g) if k.endswith("weight") else 0.1 * torch.randn(v.shape, generator=g)).cuda())
    return e


def test_bf16_fc_layers_recomputed_by_torch():
    e = engine("bf16")
    x, labels = synthetic()
    z = e.encoder_forward(x, training=True)
    torch.cuda.synchronize()
    assert torch.isfinite(z).all()
    lin = (0, 3, 6, 9, 13, 17, 21)
    prev = e.debug_activation(1)                                    # conv2 output, internal [w][c] order
    for i, li in enumerate(lin):
        st = e.debug_bn_stats(1 + i)                                # BN of the layer that feeds fc_i
        C = st.shape[1]
        # stored statistics == statistics of the stored activation
        pv = prev.reshape(-1, C)
        np.testing.assert_allclose(st[0].cpu().numpy(), pv.mean(0).cpu().numpy(), rtol=2e-4, atol=2e-5)
        var = pv.double().var(0, unbiased=False).float()
        np.testing.assert_allclose(st[1].cpu().numpy(), (1.0 / torch.sqrt(var + 1e-5)).cpu().numpy(), rtol=2e-3)
        W = e.values.views[f"emg_net.linear.{li}.weight"]
        b = e.values.views[f"emg_net.linear.{li}.bias"]
        if i == 0:      # internal feature order k' = w*64 + c  <->  reference k = c*12 + w
            u = (prev.reshape(N, 12, 64) * st[2] + st[3]).permute(0, 2, 1).reshape(N, 768)
        else:
            u = prev * st[2] + st[3]
        ref = torch.relu(u @ W.t() + b)
        got = e.debug_activation(2 + i)
        err = (got - ref).abs()
        scale = float(ref.abs().mean())
        assert float(err.mean()) < 6e-3 * scale + 1e-4, (i, float(err.mean()), scale)
        assert float(err.max()) < 0.05 * float(ref.abs().max()) + 1e-3, (i, float(err.max()))
        prev = got
        del u, ref, err


def test_f32_directional_derivative_full_size():
    e = engine("f32")
    x, labels = synthetic()
    theta0 = e.values.flat.clone()

    def loss_at(theta):
        e.values.flat.copy_(theta)
        z = e.encoder_forward(x, training=True)
        out, _, _ = e.head(z, labels, 1, want_grad=False)
        return float(out[0].item())

    e.grads.flat.zero_()
    z = e.encoder_forward(x, training=True)
    out, pred, _ = e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    torch.cuda.synchronize()
    L0 = float(out[0].item())
    lo, hi = math.log(1 + 40 * math.exp(-2)), math.log(1 + 40 * math.exp(2))
    assert lo < L0 < hi
    grad = e.grads.flat.clone()
    assert torch.isfinite(grad).all()
    # Direction = the normalised gradient itself: every component pushes the loss the same way, so
    # L(+)-L(-) = 2*eps*|g| stands far above the f32 rounding noise of a ~3.7 loss (a random direction over
    # 2M parameters gives a difference of ~1e-5, inside that noise).
    gn = float(grad.double().norm())
    v = (grad.double() / gn).float()
    predicted = gn                                   # <g, g/|g|>
    fds = []
    for eps in (0.02, 0.01):
        fds.append((loss_at(theta0 + eps * v) - loss_at(theta0 - eps * v)) / (2 * eps))
    e.values.flat.copy_(theta0)
    assert predicted > 1e-3
    # central differences converge to the predicted slope as eps shrinks
    assert abs(fds[1] - predicted) <= abs(fds[0] - predicted) + 0.01 * predicted
    assert fds[1] == pytest.approx(predicted, rel=3e-2), (fds, predicted)


def test_half_batches_bracket_full_batch_loss():
    e = engine("bf16")
    x, labels = synthetic(seed=4)
    z = e.encoder_forward(x, training=True)
    full = float(e.head(z, labels, 1, want_grad=False)[0][0].item())
    h = N // 2
    halves = []
    for part in (x[:h], x[h:]):
        zz = e.encoder_forward(part.contiguous(), training=True)
        halves.append(float(e.head(zz, labels[:h], 1, want_grad=False)[0][0].item()))
    # BatchNorm statistics of 84k-window halves differ from the full batch by O(1/sqrt(N)): the losses agree closely
    assert full == pytest.approx(sum(halves) / 2, rel=2e-3)
    assert math.isfinite(full)
