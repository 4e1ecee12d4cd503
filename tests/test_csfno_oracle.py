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
    torch.testing.assert_close(y, case["y"], rtol=2e-5, atol=2e-6)
    # the fp64 evaluation with the SAME noise agrees to fp32 rounding
    net64 = CSFNOOracle(cfg, case["state"], dtype=torch.float64)
    torch.manual_seed(case["forward_seed"])
    y64 = net64.forward(case["x"])
    assert float((y64.float() - case["y"]).abs().max() / case["y"].abs().max()) < 1e-5


def test_csfno_oracle_rejects_what_it_does_not_model(gold):
    case = gold["gaussian_groups2"]
    state = dict(case["state"])
    state["conditional_model.blocks.0.norm0.W_scale_labels.weight"] = torch.zeros(16, 3)
    with pytest.raises(NotImplementedError):
        CSFNOOracle(CSFNOConfig(5, 4, (12, 24), **case["kwargs"]), state)
