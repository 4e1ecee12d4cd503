"""Static spatial masking (ace_amd/masking.py) against the reference's own code (tests/golden/make_golden_masking.py: mask lookup,
name matcher, input masking with float / mean fills and exclusions, the provider's NaN output masker), and its place in the stepper:
inputs of every step masked, outputs masked, a checkpoint's ``input_masking`` / ``mask_provider`` carried in."""
import copy

import pytest
import torch

import ace_amd
from ace_amd.masking import NameMatcher, SpatialMaskProvider, StaticSpatialMaskingConfig
from _util import load_golden


@pytest.fixture(scope="module")
def gold():
    return load_golden("gen_masking.pt")


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-12345.0), torch.nan_to_num(b, nan=-12345.0)) and torch.equal(torch.isnan(a), torch.isnan(b))


def test_lookup_matcher_and_maskers_match_the_reference(gold):
    provider = SpatialMaskProvider(gold["masks"])
    for name, want in gold["lookup"].items():
        got = provider.get_mask_tensor_for(name)
        assert (got is None) == (want is None) and (want is None or torch.equal(got, want)), name
    for cfg, decisions in gold["matcher"]:
        m = NameMatcher(cfg)
        for name, want in decisions.items():
            assert m.match(name) == want, (cfg, name)
    for case in gold["cases"]:
        masker = StaticSpatialMaskingConfig.from_state(case["config"]).build(mask=provider, means=gold["means"])
        out = masker(gold["data"])
        assert list(out) == list(case["out"])
        for k, v in case["out"].items():
            assert torch.equal(out[k], v), (case["config"], k)
        assert all(torch.equal(gold["data"][k], v) for k, v in gold["data"].items())        # the input mapping is not mutated
    out = provider.build_output_spatial_masker()(gold["data"])
    for k, v in gold["output_masked"].items():
        assert _same(out[k], v), k
    with pytest.raises(ValueError, match="mask_"):
        SpatialMaskProvider({"sst": torch.ones(2, 2)})
    with pytest.raises(ValueError, match="0 or 1"):
        StaticSpatialMaskingConfig(mask_value=2)
    with pytest.raises(ValueError, match="fill_values mapping"):
        StaticSpatialMaskingConfig(mask_value=0, fill_value="mean").build(mask=provider)
    with pytest.raises(KeyError, match="missing key"):
        StaticSpatialMaskingConfig(mask_value=0, fill_value="mean").build(mask=provider, means={"sst": torch.tensor(1.0)})(gold["data"])
    assert SpatialMaskProvider.from_state(provider.get_state()).masks.keys() == provider.masks.keys()


def test_stepper_masks_inputs_and_outputs():
    """input_masking on args.input AND next_step_input_data before the step (single_module.py:1045-1075), the provider's output
    masker after it: with a stub network the effect is visible field by field."""
    from ace_amd.registry import Module
    from ace_amd.step import NormalizationConfig, SingleModuleStep

    class Net(torch.nn.Module):          # in: [f, p] -> out: [p, d]
        def forward(self, x):
            return torch.stack([x[:, 1] + x[:, 0], 2.0 * x[:, 0]], dim=1)

    names = ["f", "p", "d"]
    norm = NormalizationConfig(means={"f": 1.0, "p": 2.0, "d": 3.0}, stds={k: 1.0 for k in names})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 4, "num_layers": 1}),
        in_names=["f", "p"], out_names=["p", "d"], normalization=norm)
    mask = torch.ones(4, 8)
    mask[:2] = 0.0                                          # rows 0-1: no valid data
    provider = SpatialMaskProvider({"mask_2d": mask})
    info = ace_amd.DatasetInfo((4, 8), mask_provider=provider)
    step = SingleModuleStep(cfg, info, cfg.normalization.build(names), device="cpu")
    step.module = Module(Net(), None)
    stepper = ace_amd.Stepper(step, dataset_info=info, input_masking={"mask_value": 0, "fill_value": "mean"})
    ic = {"p": torch.full((1, 1, 4, 8), 5.0)}
    forcing = {"f": torch.full((1, 3, 4, 8), 4.0)}
    out, state = stepper.predict(ic, forcing)
    # masked rows: inputs replaced by the means -> normalised 0 -> network gives 0, 0 -> denormalised means -> then NaN by the output masker
    assert bool(torch.isnan(out["p"][..., :2, :]).all()) and bool(torch.isnan(out["d"][..., :2, :]).all())
    # valid rows: f_norm = 3, p_norm = 3 -> p = 3 + 3 + mean 2 = 8, d = 6 + 3 = 9
    assert torch.equal(out["p"][:, 0, 2:], torch.full((1, 2, 8), 8.0)) and torch.equal(out["d"][:, 0, 2:], torch.full((1, 2, 8), 9.0))
    # the fed-back state carries the NaNs; the next step's input masking fills them with the mean again (finite outputs where valid)
    assert bool(torch.isfinite(out["p"][:, 1, 2:]).all()) and bool(torch.isnan(state["p"][..., :2, :]).all())
    with pytest.raises(ValueError, match="mask_provider"):
        ace_amd.Stepper(step, dataset_info=ace_amd.DatasetInfo((4, 8)), input_masking={"mask_value": 0})
    from ace_amd.rollout import RolloutEngine
    with pytest.raises(NotImplementedError, match="masking"):
        RolloutEngine(stepper, batch=1, n_forward_steps=2)


def test_checkpoint_carries_input_masking_and_masks():
    from test_checkpoint_cpu import _reference_style_checkpoint
    from ace_amd.checkpoint import load_stepper
    ckpt, _ = _reference_style_checkpoint()
    plain = load_stepper(copy.deepcopy(ckpt), device="cpu").stepper
    assert not plain._masks
    ckpt["stepper"]["config"]["input_masking"] = {"mask_value": 0, "fill_value": 0.0, "exclude_names_and_prefixes": None}
    ckpt["stepper"]["dataset_info"]["mask_provider"] = {"masks": {"mask_2d": torch.ones(8, 16)}}
    loaded = load_stepper(ckpt, device="cpu")
    assert loaded.stepper._masks and loaded.dataset_info.mask_provider.get_mask_tensor_for("anything") is not None
    assert "dataset_info.mask_provider" not in loaded.ignored
    bad = copy.deepcopy(ckpt)
    bad["stepper"]["dataset_info"]["mask_provider"] = None
    with pytest.raises(ValueError, match="mask_provider"):
        load_stepper(bad, device="cpu")
