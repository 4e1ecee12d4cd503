"""Groundwork for SURVEY 8(f) rank 1: the NoiseConditionedSFNO CPU oracle (oracle/csfno.py) against outputs of the real
reference module (tests/golden/make_golden_csfno.py -> gen_csfno.pt).  The forward draws its conditioning noise from
torch's global RNG; seeding it as the golden run did reproduces the reference's draws (isotropic: real parts then
imaginary parts of the spectral coefficients; gaussian: one randn)."""
import os

import pytest
import torch

from oracle.csfno import CSFNOConfig, CSFNOOracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_csfno.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


@pytest.mark.parametrize("name", ["isotropic_affine_bigskipnorm", "gaussian_groups2", "equiangular_nomlp"])
def test_csfno_oracle_matches_reference(gold, name):
    case = gold[name]
    cfg = CSFNOConfig(in_chans=5, out_chans=4, img_shape=(12, 24), **case["kwargs"])
    net = CSFNOOracle(cfg, case["state"], dtype=torch.float32)
    torch.manual_seed(case["forward_seed"])
    y = net.forward(case["x"])
    torch.testing.assert_close(y, case["y"])
    # the fp64 evaluation with the SAME noise agrees to fp32 rounding
    net64 = CSFNOOracle(cfg, case["state"], dtype=torch.float64)
    torch.manual_seed(case["forward_seed"])
    y64 = net64.forward(case["x"])
    assert float((y64.float() - case["y"]).abs().max() / case["y"].abs().max()) < 1e-5


def test_csfno_oracle_rejects_what_it_does_not_model(gold):
    case = gold["gaussian_groups2"]
    state = dict(case["state"])
    state["conditional_model.blocks.0.norm0.W_scale.weight"] = torch.zeros(16, 3)     # scalar context embedding: not in this family
    with pytest.raises(NotImplementedError):
        CSFNOOracle(CSFNOConfig(5, 4, (12, 24), **case["kwargs"]), state)


# ---- label / positional context (stochastic_sfno.py:88-175, conditional_sfno/layers.py:160-318)
CTX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_csfno_context.pt")


@pytest.fixture(scope="module")
def ctx_gold():
    return torch.load(CTX, map_location="cpu", weights_only=False)


class _Info:
    def __init__(self, shape, labels):
        self.img_shape = shape
        self.all_labels = set(labels)


@pytest.mark.parametrize("name", ["labels3_pos4", "labels3_embed2_pos2_isotropic", "labels2_nopos"])
def test_csfno_oracle_with_labels_and_positional_context(ctx_gold, name):
    """one-hot and soft labels, optional label embedding, positional context with its label interaction: the oracle against the
    reference's own output (same RNG draws for the noise)"""
    case = ctx_gold[name]
    cfg = CSFNOConfig(in_chans=5, out_chans=4, img_shape=(12, 24), **case["kwargs"])
    torch.manual_seed(case["forward_seed"])
    y = CSFNOOracle(cfg, case["state"], dtype=torch.float32).forward(case["x"], labels=case["labels"])
    torch.testing.assert_close(y, case["y"])
    with pytest.raises(ValueError):
        CSFNOOracle(cfg, case["state"]).forward(case["x"])                       # labels must be provided


@pytest.mark.parametrize("name", ["labels3_pos4", "labels3_embed2_pos2_isotropic", "labels2_nopos"])
def test_native_module_mirrors_the_context_parameters_and_merges_them(ctx_gold, name):
    """ace_amd's NoiseConditionedSFNO: (1) the reference's state_dict names, order and shapes incl. the wrapper's label_embedding /
    pos_embed / label_pos_embed and every norm's W_*_labels / W_*_pos, strict load; (2) the host-side merge the native path relies
    on - ONE 1 x 1 convolution over cat(noise, positional context, label planes, ones) with [W_2d | W_pos | W_labels.weight |
    W_labels.bias] - equals the reference formula scale = 1 + W_2d(noise) + W_labels(labels) + W_pos(pos) (layers.py:262-318)."""
    import ace_amd
    from ace_amd.labels import BatchLabels
    case = ctx_gold[name]
    mod = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=dict(case["kwargs"]), conditional=True).build(
        5, 4, _Info((12, 24), case["all_labels"]))
    net = mod.torch_module
    assert list(net.state_dict()) == list(case["state"])
    assert all(tuple(v.shape) == tuple(case["state"][k].shape) for k, v in net.state_dict().items())
    mod.load_state({**case["state"], "label_encoding": {"labels": case["all_labels"]}})
    assert mod._label_encoding.names == case["all_labels"]
    assert mod.get_state()["label_encoding"] == {"labels": case["all_labels"]}
    B, H, W = 3, 12, 24
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(B, net.cfg.noise_embed_dim, H, W, generator=g)
    lab = case["labels"]
    if hasattr(net, "label_embedding"):
        lab = torch.nn.functional.linear(lab, net.label_embedding.weight, net.label_embedding.bias)
    fields = [noise]
    pos = None
    if net.pos_dim > 0:
        pos = net.pos_embed.detach().repeat(B, 1, 1, 1) + torch.einsum("bl,lpxy->bpxy", lab, net.label_pos_embed.detach())
        fields.append(pos)
    fields += [lab[:, :, None, None].expand(B, lab.shape[1], H, W), torch.ones(B, 1, H, W)]
    cond = torch.cat(fields, dim=1).detach()
    assert cond.shape[1] == net.cond_dim
    n0 = net.conditional_model.blocks[0].norm0
    for which in ("scale", "bias"):
        merged = torch.nn.functional.conv2d(cond.double(), n0.merged(which).double()[:, :, None, None])
        want = torch.nn.functional.conv2d(noise.double(), getattr(n0, f"W_{which}_2d").weight.detach().double())
        lin = getattr(n0, f"W_{which}_labels")
        want = want + torch.nn.functional.linear(lab.detach().double(), lin.weight.detach().double(), lin.bias.detach().double())[:, :, None, None]
        if pos is not None:
            want = want + torch.nn.functional.conv2d(pos.detach().double(), getattr(n0, f"W_{which}_pos").weight.detach().double())
        assert float((merged - want).abs().max()) <= 1e-12
    # the registry wrapper conforms BatchLabels to the module's encoding (module.py:74-84)
    bl = BatchLabels(torch.eye(len(case["all_labels"]))[:, list(reversed(range(len(case["all_labels"]))))], list(reversed(case["all_labels"])))
    assert torch.equal(bl.conform_to_encoding(mod._label_encoding).tensor, torch.eye(len(case["all_labels"])))
    with pytest.raises(TypeError):
        mod(torch.zeros(1, 5, 12, 24))                                           # labels are required for conditional models


# ---- the reference-HELD, RNG-free golden of this family (conditional_sfno/test_sfnonet.py:162-191)
def test_csfno_oracle_reproduces_the_reference_held_checkpoint_golden():
    """testdata/test_sfnonet_checkpoint_{input,output}.pt: the state dict loads through the legacy spectral-filter layout hooks
    (s2convolutions.py:279-365: (1, in, out, L, 2) -> (G, L, out, in, 2)), the scalar context embedding rides on the label vector
    (tests/_util.py: csfno_reference_checkpoint_case), and the oracle meets the reference's own assert_close bar - no RNG involved."""
    import ace_amd
    from _util import csfno_reference_checkpoint_case
    c = csfno_reference_checkpoint_case()
    mod = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=dict(c["kwargs"]), conditional=True).build(
        2, 3, _Info((9, 18), c["labels"]))
    net = mod.torch_module
    net.load_state_dict(c["state"], strict=True)               # old filter layout in, grouped layout held
    assert tuple(net.state_dict()["conditional_model.blocks.0.filter.filter.weight"].shape) == (1, 9, 16, 16, 2)
    kw = {k: v for k, v in c["kwargs"].items() if k != "filter_type"}
    oracle = CSFNOOracle(CSFNOConfig(in_chans=2, out_chans=3, img_shape=(9, 18), **kw), dict(net.state_dict()), dtype=torch.float32)
    y = oracle.forward(c["x"], noise=c["noise"], labels=c["label_vector"])
    torch.testing.assert_close(y, c["y"])                      # default tolerances: the bar of the reference's validate_tensor
    # the ungrouped 4-D layout (before the group dimension existed) loads to the same parameters
    old = dict(c["state"])
    for k in list(old):
        if k.endswith("filter.filter.weight"):
            old[k] = old[k][0]
    mod2 = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=dict(c["kwargs"]), conditional=True).build(
        2, 3, _Info((9, 18), c["labels"]))
    mod2.torch_module.load_state_dict(old, strict=True)
    for k, v in net.state_dict().items():
        assert torch.equal(v, mod2.torch_module.state_dict()[k]), k


@pytest.mark.parametrize("name", ["csfno_block", "csfno_block_8_groups"])
def test_csfno_oracle_reproduces_the_reference_held_block_goldens(name):
    """fme/core/benchmark/testdata/csfno_block{,_8_groups}-regression.pt (conditional_sfno/benchmark.py:100-119; one dense and one
    8-group spectral filter, lobatto 9 x 18, noise + label + positional context): the ONE-block network of tests/_util.py
    (identity encoder / decoder) minus norm0(x) is the stand-alone block, at the reference's own assert_close bar."""
    import ace_amd
    from _util import csfno_block_case, csfno_block_norm0
    c = csfno_block_case(name)
    net = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=dict(c["kwargs"]), conditional=True).build(
        16, 16, _Info((9, 18), c["labels"])).torch_module
    net.load_state_dict(c["state"], strict=True)
    oracle = CSFNOOracle(CSFNOConfig(in_chans=16, out_chans=16, img_shape=(9, 18), **c["kwargs"]), dict(net.state_dict()), dtype=torch.float32)
    y = oracle.forward(c["x"], noise=c["noise"], labels=c["label_vector"])
    torch.testing.assert_close((y.double() - csfno_block_norm0(c)).float(), c["y"])
