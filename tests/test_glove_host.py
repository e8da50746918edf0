"""Host-side checks of the glove-angle class encoder option: the parameter table follows the
order nn.Module.state_dict() would give the reference's un-commented GLOVENet, the L2 membership follows
GLOVENet.l2 (code/models.py:467-472), and the oracle's branch trains."""
import pytest
import torch

from oracle import ref_cpu as oc

BEST = dict(d_e=16, lr_emg=9.761e-4, reg_emg=7.103e-5, dp_emg=0.0, lr_glove=2.653e-3, reg_glove=2.840e-6, dp_glove=0.0)


@pytest.mark.parametrize("adabn", [False, True])
def test_param_table_order_and_l2_membership(adabn):
    from contrastiveprosthetics_amd.engine import l2_member, param_specs
    sd = oc.add_glove_encoder(oc.init_state_dict(1, 16, adabn), 2, adabn)
    specs = param_specs(adabn, 16, "glove")
    trainable = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k and k != "logit_scale"]
    assert list(specs) == trainable
    for k in specs:
        assert tuple(specs[k]) == tuple(sd[k].shape), k
    glove = [k for k in specs if k.startswith("glove_net.")]
    bn = "glove_net.linear.2.bn" if adabn else "glove_net.linear.2"
    assert glove == ["glove_net.linear.1.weight", bn + ".weight", bn + ".bias", "glove_net.easy.0.weight",
                     "glove_net.easy.0.bias", "glove_net.last.0.weight"]
    m = oc.OracleModel(sd, BEST, adabn=adabn, class_encoder="glove")
    assert [k for k in specs if l2_member(k)] == m.l2_keys()[0] + m.l2_keys()[1]
    # stock BN's gamma is L2-regularised (its name holds neither 'bn' nor 'bias'), AdaBN's is not -- as in the sEMG net
    assert l2_member(bn + ".weight") == (not adabn)
    # the one-hot table stays in the state_dict, unused, exactly as `last` does in the reference's one-hot mode
    assert param_specs(adabn, 16, "onehot").keys() < specs.keys()


def test_oracle_glove_branch_learns():
    torch.manual_seed(0)
    sd = oc.add_glove_encoder(oc.init_state_dict(3, 16, True), 4, True)
    m = oc.OracleModel(sd, dict(BEST, lr_emg=3e-3, lr_glove=3e-3), adabn=True, requires_grad=True, class_encoder="glove")
    m.set_train()
    opts = m.make_optimizers()
    g = torch.Generator().manual_seed(1)
    mu_e, mu_g = torch.randn(41, 12, generator=g), torch.randn(41, 20, generator=g)
    losses = []
    for _ in range(12):
        EMG = (mu_e[None] + 0.5 * torch.randn(8, 41, 12, generator=g)).reshape(8, 41, 1, 1, 12)
        GLOVE = mu_g[None] + 0.1 * torch.randn(8, 41, 20, generator=g)
        losses.append(m.train_step(EMG, GLOVE, torch.arange(41).repeat(8), opts))
    assert losses[-1] < losses[0] - 0.05
    assert m.sd["glove_net.easy.0.bias"].grad is None            # unused by this encoder, outside the L2 term
