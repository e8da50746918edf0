"""cp_encoder_backward_ev (include/cpnative.h): the caller's event is recorded when every gradient except the conv
stack's is final.  A copy of that part of the flat gradient buffer taken on a side stream right behind the event must
equal the buffer after the whole backward -- nothing that runs after the event may touch it -- and the gradients are
those of the plain call."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = 41


@pytest.mark.parametrize("dtype,dp", [("bf16", 0.0635), ("bf16", 0.0), ("f32", 0.0635)])
def test_fc_gradients_are_final_at_the_event(dtype, dp):
    from contrastiveprosthetics_amd.engine import Engine
    n = 20000 - 20000 % T
    g = torch.Generator().manual_seed(11)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(n // T, T, 12, generator=g)).reshape(n, 12).cuda()
    labels = torch.arange(T).repeat(n // T).cuda()

    def run(with_event, second_stream=False):
        e = Engine(adabn=False, dtype=dtype, dp_emg=dp, device="cuda", seed=123)
        e.aux_stream_enabled = second_stream                # (an engine with an event registered runs on one stream anyway: engine._cfg)
        e.init_parameters(5)
        e.grads.flat.fill_(float("nan"))                    # whatever is not written shows
        split = e.grads.offsets["emg_net.linear.0.weight"][0]
        snap = None
        z = e.encoder_forward(x, training=True)
        e.head(z, labels, 1, want_grad=True)
        if with_event:
            ev = torch.cuda.Event()
            ev.record()
            e.fc_grads_ready = ev
            side = torch.cuda.Stream()
            e.encoder_backward(x)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                snap = e.grads.flat[split:].clone()
        else:
            e.encoder_backward(x)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in e.grads.views.items()}, e.grads.flat.clone(), split, snap

    plain, _, split, _ = run(False)
    views, flat, split2, snap = run(True)
    assert split == split2 and 0 < split < flat.numel()
    # final at the event (NaN padding between tensors included: nobody writes there either)
    assert torch.equal(torch.nan_to_num(snap, nan=-7.0), torch.nan_to_num(flat[split:], nan=-7.0))
    for k, v in views.items():
        if not (k.startswith("emg_net.") or k.startswith("glove_net.easy.")):
            continue                                        # glove-branch layers the one-hot path never uses: no gradient
        assert torch.isfinite(v).all(), k                   # every gradient was written
        assert torch.equal(v, plain[k]), k                  # and the call computes what the plain (one-stream) one does
    # the conv stack is what is still being computed behind the event: it is the (small) front of the buffer
    assert split * 4 < 0.2 * 2 ** 20
    # with the second stream (the default without an event) fc5's and fc4's weight gradients are summed over 64 row splits each
    # instead of 32: the same numbers to f32 summation order, here and in everything below fc4 that their sums feed
    two, _, _, _ = run(False, second_stream=True)
    for k, v in two.items():
        if k.startswith("emg_net."):
            ref = plain[k].double()
            assert float((v.double() - ref).norm()) <= 2e-3 * float(ref.norm()) + 1e-12, k


def test_glove_class_encoder_gradients_are_final_at_the_event():
    """With the glove-angle class encoder the class encoder's gradients live in the same bucket; its
    backward is enqueued first (models.py, bench.py), so they too are final when the event fires."""
    from contrastiveprosthetics_amd.engine import Engine
    groups = 300
    n = groups * T
    g = torch.Generator().manual_seed(21)
    mu = torch.randn(T, 12, generator=g)
    x = (mu[None] + torch.randn(groups, T, 12, generator=g)).reshape(n, 12).cuda()
    glove = (torch.randn(T, 20, generator=g)[None] + 0.3 * torch.randn(groups, T, 20, generator=g)).cuda()
    labels = torch.arange(T).repeat(groups).cuda()
    e = Engine(adabn=False, dtype="bf16", dp_emg=0.0635, device="cuda", seed=5, class_encoder="glove")
    e.init_parameters(7)
    e.grads.flat.zero_()
    split = e.grads.offsets["emg_net.linear.0.weight"][0]
    ev = torch.cuda.Event()
    ev.record()
    e.fc_grads_ready = ev
    side = torch.cuda.Stream()
    z = e.encoder_forward(x, training=True)
    zg = e.glove_forward(glove, training=True)
    e.head_glove(z, zg, labels, 1, want_grad=True)
    e.glove_backward()
    e.encoder_backward(x)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        snap = e.grads.flat[split:].clone()
    torch.cuda.synchronize()
    assert torch.equal(snap, e.grads.flat[split:])
    moved = [k for k, v in e.grads.views.items() if k.startswith("glove_net.") and float(v.abs().sum()) > 0]
    assert moved, "the class encoder's gradients were written"
