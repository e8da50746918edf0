"""The two tile schedules of the persistent fc GEMM kernels (cp_set_tile_schedule, include/cpnative.h): the outputs of
the GEMMs are identical (same tiles, same arithmetic, whichever workgroup computes them); the BatchNorm partial sums
are grouped per sample tile (dynamic) or per workgroup (static) and then added in f64, so the statistics and everything
downstream agree to fp32 rounding of the partial rows; each schedule is reproducible run to run -- the dynamic one
although which workgroup runs which tile changes from launch to launch.
40,000 rows = 157 sample tiles (ragged last tile), 167,936 rows = the bench size (656 tiles, more than one round)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = 41
STATIC, DYNAMIC = 0, 1


def _backward(schedule, n, dp, seed=5):
    from contrastiveprosthetics_amd import _lib
    from contrastiveprosthetics_amd.engine import Engine
    lib = _lib.load()
    prev = lib.cp_get_tile_schedule()
    _lib.check(lib.cp_set_tile_schedule(schedule), "cp_set_tile_schedule")
    try:
        assert lib.cp_get_tile_schedule() == schedule
        g = torch.Generator().manual_seed(3)
        mu = torch.randn(T, 12, generator=g)
        x = (mu[None] + torch.randn(n // T, T, 12, generator=g)).reshape(n, 12).cuda()
        labels = torch.arange(T).repeat(n // T).cuda()
        e = Engine(adabn=False, dtype="bf16", dp_emg=dp, device="cuda", seed=123)
        e.init_parameters(seed)
        e.grads.flat.zero_()
        z = e.encoder_forward(x, training=True)
        e.head(z, labels, 1, want_grad=True)
        e.encoder_backward(x)
        torch.cuda.synchronize()
        assert torch.isfinite(e.grads.flat).all()
        return z.float().clone(), e.grads.flat.clone()
    finally:
        lib.cp_set_tile_schedule(prev)


@pytest.mark.parametrize("n", [40000 - 40000 % T, 167936])
def test_each_schedule_is_reproducible(n):
    for schedule in (STATIC, DYNAMIC):
        z0, g0 = _backward(schedule, n, 0.0635)
        z1, g1 = _backward(schedule, n, 0.0635)
        assert torch.equal(z0, z1), schedule
        assert torch.equal(g0, g1), schedule


@pytest.mark.parametrize("dp", [0.0, 0.0635])
def test_schedules_agree(dp):
    n = 167936
    zs, gs = _backward(STATIC, n, dp)
    zd, gd = _backward(DYNAMIC, n, dp)
    # embeddings: seven layers of BatchNorm statistics whose fp32 partial rows were grouped differently
    assert float((zs - zd).abs().max()) <= 2e-2 * float(zs.abs().max())
    a, b = gs.double(), gd.double()
    cos = float(a @ b / (a.norm() * b.norm()))
    assert cos > 0.9995, cos


def test_setter_rejects_unknown_mode():
    from contrastiveprosthetics_amd import _lib
    lib = _lib.load()
    before = lib.cp_get_tile_schedule()
    assert lib.cp_set_tile_schedule(7) != 0
    assert lib.cp_get_tile_schedule() == before
