"""The library's own launch profiler (cp_profile_*, include/cpnative.h) that bench.py's live roofline numbers come from:
one record per launch of an enabled kind, on the launch stream; cp_profile_disable / cp_profile_resume sample steps
without dropping earlier records; cp_profile_enable starts over."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = 41


def test_records_survive_pause_and_resume():
    from contrastiveprosthetics_amd.engine import Engine
    n = 8200 - 8200 % T
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, 12, generator=g).cuda()
    labels = torch.arange(T).repeat(n // T).cuda()
    e = Engine(adabn=False, dtype="bf16", dp_emg=0.0635, device="cuda", seed=1)
    e.init_parameters(3)

    def step():
        z = e.encoder_forward(x, training=True)
        e.head(z, labels, 1, want_grad=True)
        e.encoder_backward(x)

    kinds = ["fc_fwd", "fc_fwd_ws", "fc_wgrad"]
    e.profile_enable(kinds, max_records=256)
    step()
    e.profile_disable()
    step()                                   # not recorded
    e.profile_resume()
    step()
    e.profile_disable()
    torch.cuda.synchronize()
    prof = e.profile_summary()
    assert set(prof) == set(kinds)
    # seven fc layers, two recorded steps: fc1 (K = 768) on the tile-staged kernel, fc2..fc7 on the weight-stationary one
    assert prof["fc_fwd"][1] == 2 * 1 and prof["fc_fwd_ws"][1] == 2 * 6
    # bf16 with dropout: fc7 + fc6 share a launch; with the second stream (the default) fc5 goes alone beside the critical path,
    # without it fc5 + fc4 share one too
    assert prof["fc_wgrad"][1] == 2 * (6 if e._aux is not None else 5)
    assert all(ms > 0 for ms, _ in prof.values())
    e.profile_enable(kinds, max_records=256)  # starts over
    e.profile_disable()
    assert e.profile_summary() == {} or all(nrec == 0 for _, nrec in e.profile_summary().values())
