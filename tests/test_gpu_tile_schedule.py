"""The two tile schedules of the persistent fc GEMM kernels (cp_config.tile_schedule, include/cpnative.h): the outputs of
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
    from contrastiveprosthetics_amd.engine import Engine
    g = torch.Generator().manual_seed(3)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(n // T, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(n // T).cuda()
    e = Engine(adabn=False, dtype="bf16", dp_emg=dp, device="cuda", seed=123)
    e.tile_schedule = schedule                               # per engine: carried in its cp_config
    e.init_parameters(seed)
    e.grads.flat.zero_()
    z = e.encoder_forward(x, training=True)
    e.head(z, labels, 1, want_grad=True)
    e.encoder_backward(x)
    torch.cuda.synchronize()
    assert torch.isfinite(e.grads.flat).all()
    return z.float().clone(), e.grads.flat.clone()


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


def test_unknown_mode_is_rejected():
    from contrastiveprosthetics_amd import _lib
    from contrastiveprosthetics_amd.engine import Engine
    e = Engine(adabn=False, dtype="bf16", device="cuda")
    e.init_parameters(1)
    e.tile_schedule = 7
    with pytest.raises(_lib.CpNativeError):
        e.encoder_forward(torch.zeros(T * 100, 12, device="cuda"), training=True)


def test_two_engines_in_one_process_keep_their_own_settings():
    """options, tile schedule and hooks travel in each call's cp_config.  Two engines with different schedules
    and options, steps interleaved, give what each gives alone."""
    n = 40000 - 40000 % T
    alone = {s: _backward(s, n, 0.0635) for s in (STATIC, DYNAMIC)}
    from contrastiveprosthetics_amd.engine import Engine
    g = torch.Generator().manual_seed(3)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(n // T, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(n // T).cuda()
    engines = {}
    for s in (STATIC, DYNAMIC):
        e = Engine(adabn=False, dtype="bf16", dp_emg=0.0635, device="cuda", seed=123)
        e.tile_schedule = s
        e.init_parameters(5)
        e.grads.flat.zero_()
        engines[s] = e
    zs = {s: e.encoder_forward(x, training=True) for s, e in engines.items()}          # forwards interleaved, then heads, then backwards
    for s, e in engines.items():
        e.head(zs[s], labels, 1, want_grad=True)
    for s, e in engines.items():
        e.encoder_backward(x)
    torch.cuda.synchronize()
    for s, e in engines.items():
        assert torch.equal(zs[s].float(), alone[s][0]) and torch.equal(e.grads.flat, alone[s][1]), s
