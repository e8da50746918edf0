"""The algebra behind round 4's backward kernels, checked in float64 numpy on small random cases (no GPU, no oracle import: the
identities are about the reference's layer definitions, code/models.py:248-264 and :296-315).

1. conv stack (csrc/kernels_misc.cuh, conv2_wgrad_finish_kernel; csrc/conv_kernels.cuh, conv2_dgrad_conv1_kernel): with
   u1 = s r1 + t zero-padded AFTER the affine map, P[o,tap,c] = sum g[n,q,o] r1[n,q+tap-1,c] and G[o,tap] = the column sums of g over
   the positions whose tap lands inside the window,
       dW2 = s P + t G,   sum g_v1 = sum_{o,tap} W2 G,   sum g_v1 r1 = sum_{o,tap} W2 P.
2. projection behind fc7's dropout (csrc/gemm_tn.cuh, proj_wgrad_sums_kernel / proj_wgrad_finish_kernel): with keep the dropout
   decision, A = dz^T (keep . r8), B = dz^T keep,
       dWp = (s A + t B) / (1 - p),   sum g = sum_j Wp B / (1 - p),   sum g r8 = sum_j Wp A / (1 - p)."""
import numpy as np


def _shift(t, d):
    """t: (N, 12, C); s[:, w] = t[:, w + d], zero outside 0..11"""
    out = np.zeros_like(t)
    if d == 0:
        out[:] = t
    elif d > 0:
        out[:, :12 - d] = t[:, d:]
    else:
        out[:, -d:] = t[:, :12 + d]
    return out


def test_conv2_weight_gradient_carries_batchnorm1_backward_sums():
    rng = np.random.default_rng(0)
    N, C = 5, 64
    r1 = np.maximum(rng.standard_normal((N, 12, C)), 0.0)
    s, t = rng.standard_normal(C) * 0.5 + 1.0, rng.standard_normal(C) * 0.3
    W2 = rng.standard_normal((C, C, 3)) * 0.1                      # [o][c][tap]
    g = rng.standard_normal((N, 12, C)) * 0.01                     # dL/d(conv2 pre-activation) [n][q][o]
    u1 = r1 * s + t                                                # zero padding happens in _shift: after the affine map
    # direct: conv2 forward is y[n,q,o] = sum_{tap,c} W2[o,c,tap] u1[n,q+tap-1,c]
    dW2 = np.stack([np.einsum("nqo,nqc->oc", g, _shift(u1, tap - 1)) for tap in range(3)], -1)
    g_v1 = sum(np.einsum("nqo,oc->nqc", _shift(g, 1 - tap), W2[:, :, tap]) for tap in range(3))
    S1, S2 = g_v1.sum((0, 1)), (g_v1 * r1).sum((0, 1))
    # the kernels' route
    P = np.stack([np.einsum("nqo,nqc->oc", g, _shift(r1, tap - 1)) for tap in range(3)], -1)      # raw product, [o][c][tap]
    col = g.sum(0)                                                  # [q][o]: the bias-gradient rows of fc1's data-gradient launch
    allq = col.sum(0)
    G = np.stack([allq - col[0], allq, allq - col[11]], -1)         # [o][tap]
    np.testing.assert_allclose(s[None, :, None] * P + t[None, :, None] * G[:, None, :], dW2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.einsum("oct,ot->c", W2, G), S1, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.einsum("oct,oct->c", W2, P), S2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(allq, g.sum((0, 1)), rtol=1e-12)     # conv2's bias gradient


def test_projection_weight_gradient_carries_fc7_batchnorm_backward_sums_behind_a_dropout():
    rng = np.random.default_rng(1)
    N, F, J, p = 37, 512, 16, 0.0635
    r8 = np.maximum(rng.standard_normal((N, F)), 0.0)
    keep = (rng.random((N, F)) >= p).astype(np.float64)
    s, t = rng.standard_normal(F) * 0.5 + 1.0, rng.standard_normal(F) * 0.3
    Wp = rng.standard_normal((J, F)) * 0.05
    dz = rng.standard_normal((N, J)) * 0.01
    inv_keep = 1.0 / (1.0 - p)
    u8 = keep * (r8 * s + t) * inv_keep
    dWp = dz.T @ u8
    gmask = keep * (dz @ Wp) * inv_keep                              # dL/d(BatchNorm8 output)
    A, B = dz.T @ (keep * r8), dz.T @ keep
    np.testing.assert_allclose((s * A + t * B) * inv_keep, dWp, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose((Wp * B).sum(0) * inv_keep, gmask.sum(0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose((Wp * A).sum(0) * inv_keep, (gmask * r8).sum(0), rtol=1e-10, atol=1e-12)
