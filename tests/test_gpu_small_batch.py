"""The small-batch path (csrc/small.cuh: up to 64 groups = 2624 windows, the reference's own batch sizes, code/train.py:185) against
the large-batch kernels on the same inputs (cp_config.options, CP_OPT_NO_SMALL), one whole training step: z, loss, BatchNorm running
statistics and every gradient.  The two paths compute the same sums in different groupings (the small path's BatchNorm totals are
fixed-point integers, its weight gradients whole-batch or 2..8 row splits; its data gradients carry BatchNorm + ReLU backward in the
staging of the NEXT launch instead of an epilogue).

What bounds the agreement of the GRADIENTS is the ReLU kink, not the arithmetic: a pre-activation within rounding distance of zero
is positive on one path and zero on the other, and the whole gradient of that element is then on or off.  Measured (tools/
diag_small_vs_large.py, round 3): f32 activations agree to 3e-6 of their scale, so about one element in a million flips -- a step
of 328 x 512 x 7 elements has ~1 -- and everything BELOW the first flip in the backward pass moves by ~3e-3 (one row of 328
perturbed by ~6 %), everything above agrees to 4e-6; bf16 activations agree to 2^-9 per layer (2 % of max at z), 0.2..0.6 % of the elements
of a layer flip (0.63 % at fc7, 8 groups), ~1 per row of 256 live features, ~4 % per layer in quadrature: 2.5 % at the projection (no ReLU above it: bf16 rounding of dz and
of the operands alone) to 20 % at conv1, cosine 0.96..0.98.  Any two bf16 pipelines differ like this and both differ from f32 by
1/sqrt(2) of it; it is zero-mean (a model trained 60 steps in f32 and in bf16 reaches the same loss, tests/test_gpu_parity.py).  So the bars
are two-tier: tight on what no ReLU precedes in the backward pass (the projection, the last BatchNorm's affine, the class table),
loose below, and the flip rate itself is measured and bounded.  8 groups = 328 rows (the reference's smallest), 33 groups = 1353 rows (ragged 32-row tiles, 6 splits),
64 groups = 2624 rows (the dispatch limit).  The parity tests against the oracle at B <= 64 (tests/test_gpu_parity.py,
test_gpu_api.py) run through the small path too; this file pins the two device paths to each other and the small path's
run-to-run bit-exactness (its reductions are order-independent by construction)."""
import pytest
import torch

from contrastiveprosthetics_amd import _lib

pytestmark = pytest.mark.gpu
T = 41


def _step(dtype, groups, no_small, dp=0.0635):
    from contrastiveprosthetics_amd.engine import Engine
    n = groups * T
    g = torch.Generator().manual_seed(17)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(groups, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(groups).cuda()
    e = Engine(adabn=False, dtype=dtype, dp_emg=dp, device="cuda", seed=321)
    e.init_parameters(7)
    gg = torch.Generator().manual_seed(5)
    for k in e.specs:                                        # non-trivial BatchNorm affine
        v = e.values.views[k]
        if k.startswith("emg_net.") and v.dim() == 1 and v.numel() in (64, 512) and (".bn" in k or "conv_emg.2" in k or "conv_emg.5" in k or
                                                                                       k.split(".")[-2] in ("2", "5", "8", "11", "15", "19", "23")):
            v.copy_((1.0 + 0.2 * torch.randn(v.shape, generator=gg) if k.endswith("weight") else 0.1 * torch.randn(v.shape, generator=gg)).cuda())
    e.grads.flat.fill_(float("nan"))
    e.options["no_small"] = 1 if no_small else 0
    z = e.encoder_forward(x, training=True).clone()
    out, _, _ = e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    torch.cuda.synchronize()
    act8 = e.debug_activation(8)                         # fc7's stored output relu(.)
    masks = [e.debug_activation(l) > 0 for l in range(1, 9)]          # the ReLU masks of conv2, fc1..fc7
    grads = {k: e.grads.views[k].clone() for k in e.specs if k.startswith("emg_net.") or k.startswith("glove_net.easy.")}
    running = {k: v.clone() for k, v in e.running.items() if torch.is_tensor(v) and v.is_floating_point()}
    return z, out.clone(), grads, running, act8, masks


@pytest.mark.parametrize("groups", [5, 8, 32, 33, 64])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_small_batch_path_equals_large_batch_kernels(dtype, groups):
    zs, outs, gs, rs, a8s, ms = _step(dtype, groups, no_small=False)
    zl, outl, gl, rl, a8l, ml = _step(dtype, groups, no_small=True)
    f32 = dtype == "f32"
    # f32: the two paths sum in different orders, so a pre-activation within a rounding step of zero can land on either side of it; the
    # forward output moves by 1e-9, that element's ReLU mask flips and its whole gradient term (1 / rows of a column sum) is there or
    # not (tools/relu_mask_flips.py, profiles/r04_relu_mask_flips.txt).  Where NO mask differs (5 and 32 groups with these seeds; the
    # kernels are deterministic) every gradient must agree to f32 rounding; elsewhere only the tensors above every ReLU can be held to that.
    n_flips = sum(int((u != v).sum()) for u, v in zip(ms, ml))
    if f32 and groups in (5, 32):
        assert n_flips == 0, n_flips
    zerr = float((zs - zl).abs().max() / zl.abs().max())
    assert zerr < (2e-5 if f32 else 3e-2), ("z", zerr)
    assert float(outs[0]) == pytest.approx(float(outl[0]), rel=2e-6 if f32 else 2e-4)
    for k in rs:
        assert torch.allclose(rs[k].float(), rl[k].float(), rtol=1e-5 if f32 else 5e-3, atol=1e-6 if f32 else 1e-3), k
    flips = float(((a8s > 0) != (a8l > 0)).float().mean())
    assert flips < (2e-5 if f32 else 1.5e-2), ("ReLU flips at fc7", flips)
    above_every_relu = ("emg_net.last.0.weight", "emg_net.linear.23.weight", "emg_net.linear.23.bias", "glove_net.easy.0.weight", "glove_net.easy.0.bias")
    for k in gl:
        a, b = gs[k].double().flatten(), gl[k].double().flatten()
        assert torch.isfinite(a).all(), k
        if float(b.norm()) == 0.0:
            assert float(a.norm()) < 1e-12, k
            continue
        cos = float(a @ b / (a.norm() * b.norm()))
        rel = float((a - b).norm() / b.norm())
        if k in above_every_relu:
            # bf16: fc7's bias gradient at 8 groups is a sum of 328 values that two bf16 pipelines round differently (measured 2.5-4.8 %
            # across dropout masks; the round-4 hash draws other masks than round 3's, which read 3.x %)
            assert rel < (2e-5 if f32 else 7e-2), (k, rel)
        elif f32 and n_flips == 0:
            dev = float((a - b).abs().max() / b.abs().max())
            assert dev < 2e-4 and rel < 2e-3, (k, dev, rel)          # (rel: bias / BatchNorm gradients are sums with heavy cancellation)
        elif f32:
            assert rel < 3e-2 * max(1, n_flips) and cos > 0.999, (k, rel, cos, n_flips)       # (n_flips flipped elements in 205..2624 rows)
        else:
            assert rel < 0.35 and cos > 0.94, (k, rel, cos)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_small_batch_step_is_bit_exact_run_to_run(dtype):
    """BatchNorm totals and BatchNorm-backward totals are 64-bit fixed-point integer atomics, weight-gradient splits are summed in slab
    order: no result depends on the order in which workgroups finish."""
    a = _step(dtype, 33, no_small=False)
    b = _step(dtype, 33, no_small=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.parametrize("no_small", [False, True])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_backward_twice_over_one_forward(dtype, no_small):
    """the small-batch form's BatchNorm-backward totals are fixed-point atomics zeroed by the FORWARD's preparation launch, so a
    second cp_encoder_backward over the same forward (gradient checks, timing the backward alone, C-ABI callers) used to add onto the
    first pass's sums.  The library now knows (per workspace) that it is a repeat and zeroes them first: both passes give the same
    gradients, bit for bit, on either kernel path; and a backward whose configuration differs from its forward's is refused."""
    from contrastiveprosthetics_amd.engine import Engine
    groups = 16
    n = groups * T
    g = torch.Generator().manual_seed(23)
    x = (torch.randn(T, 12, generator=g)[None] + torch.randn(groups, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(groups).cuda()
    e = Engine(adabn=False, dtype=dtype, dp_emg=0.0635, device="cuda", seed=3)
    e.options["no_small"] = 1 if no_small else 0
    e.init_parameters(7)
    z = e.encoder_forward(x, training=True)
    e.head(z, labels, 1, want_grad=True)
    e.grads.flat.fill_(float("nan"))
    e.encoder_backward(x)
    first = e.grads.flat.clone()
    e.grads.flat.fill_(float("nan"))
    e.encoder_backward(x)
    torch.cuda.synchronize()
    emg = torch.cat([e.grads.views[k].reshape(-1) for k in e.specs if k.startswith("emg_net.")])
    emg1 = torch.cat([first[o:o + m] for k, (o, m) in e.grads.offsets.items() if k.startswith("emg_net.")])
    assert torch.isfinite(emg).all()
    assert torch.equal(emg, emg1)
    # the other path's backward over this forward is an error, not a silent mix of two workspace layouts
    e.options["no_small"] = 0 if no_small else 1
    with pytest.raises(_lib.CpNativeError):
        e.encoder_backward(x)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_large_inputs_at_small_batches(dtype):
    """the small-batch form's fixed-point totals must not wrap on large values.  (1) inputs x 300 (sums of squares ~1e5 x
    larger than normalised data): the two kernel paths' BatchNorm statistics and embeddings agree as they do on ordinary data.
    (2) inputs so large that a total leaves the fixed-point range: NaN statistics (the sentinel of csrc/common.cuh, sm_acc_add), not
    silently wrapped ones.  (A NaN in the INPUT is a different matter and is not claimed: the kernels' ReLU is v_max_f32 /
    v_pk_max_i16, which returns 0 for a NaN operand where torch.relu returns NaN -- on either kernel path a NaN window is treated as
    an inactive unit instead of poisoning the batch)"""
    from contrastiveprosthetics_amd.engine import Engine
    groups = 16
    g = torch.Generator().manual_seed(2)
    x0 = (torch.randn(T, 12, generator=g)[None] + torch.randn(groups, T, 12, generator=g)).reshape(groups * T, 12).cuda()
    res = {}
    for no_small in (False, True):
        e = Engine(adabn=False, dtype=dtype, dp_emg=0.0, device="cuda", seed=3)
        e.options["no_small"] = 1 if no_small else 0
        e.init_parameters(7)
        z = e.encoder_forward(300.0 * x0, training=True)
        assert torch.isfinite(z).all(), no_small
        res[no_small] = (z.clone(), e.debug_bn_stats(0).clone(), e.debug_bn_stats(1).clone())
    for l in (1, 2):
        assert torch.allclose(res[False][l][0], res[True][l][0], rtol=1e-4 if dtype == "f32" else 2e-2, atol=1e-3), l
    assert float((res[False][0] - res[True][0]).abs().max()) <= (1e-3 if dtype == "f32" else 8e-2) * float(res[True][0].abs().max())
    e = Engine(adabn=False, dtype=dtype, dp_emg=0.0, device="cuda", seed=3)
    e.init_parameters(7)
    e.encoder_forward(3.0e8 * x0, training=True)          # conv1's sums of squares ~1e19 per workgroup: far beyond 2^30 in 2^-24 steps
    assert torch.isnan(e.debug_bn_stats(0)[1]).any()      # invstd of conv1's BatchNorm: poisoned, not wrapped
