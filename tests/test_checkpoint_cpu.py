"""CPU tests of the 'next' rows: checkpoint ingestion (fme/ace/stepper/single_module.py:1358-1429, 1909-1927) and the
geometry-free post-step hooks (ForcePositive: fme/core/corrector/utils.py:26-80; prescribed SST: fme/core/ocean.py:167-215,
fme/core/prescriber.py:54-117), with the hook order of step_with_adjustments (fme/core/step/single_module.py:669-716)."""
import copy
import datetime

import pytest
import torch

import ace_amd
from ace_amd.checkpoint import load_stepper, stepper_config_from_state
from ace_amd.corrector import AtmosphereCorrectorConfig, force_positive
from ace_amd.ocean import OceanConfig, Prescriber, replace_on_mask
from ace_amd.registry import Module
from ace_amd.step import NormalizationConfig

IN, OUT = ["f", "sst", "q", "frac"], ["sst", "q", "d"]
NET = {"embed_dim": 8, "num_layers": 1, "encoder_layers": 1, "operator_type": "dhconv"}


def _reference_style_checkpoint(wrap_multi_call=True, corrector=None, ocean=None):
    """A stepper state laid out like Stepper.get_state (single_module.py:1337-1347) for a small SFNO; the module
    weights are a seeded ace_amd module's state_dict under the reference's names with the wrapper's 'module.' prefix."""
    torch.manual_seed(3)
    sel = ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config=dict(NET))
    mod = sel.build(len(IN), len(OUT), ace_amd.DatasetInfo((8, 16)))
    weights = {f"module.{k}": v.clone() for k, v in mod.torch_module.state_dict().items()}
    names = sorted(set(IN) | set(OUT))
    step_cfg = {
        "builder": {"type": "SphericalFourierNeuralOperatorNet", "config": dict(sel.config)},
        "in_names": list(IN), "out_names": list(OUT),
        "normalization": {"network": {"global_means_path": None, "global_stds_path": None,
                                      "means": {n: 0.25 * (i + 1) for i, n in enumerate(names)},
                                      "stds": {n: 1.0 + 0.5 * i for i, n in enumerate(names)}},
                          "loss": None, "residual": None},
        "secondary_decoder": None, "ocean": ocean,
        "corrector": corrector if corrector is not None else dataclasses_asdict(AtmosphereCorrectorConfig()),
        "next_step_forcing_names": [], "prescribed_prognostic_names": [], "residual_prediction": False,
        "include_channel_mask_inputs": False, "global_mean_removal": None, "input_dropout": None,
    }
    step_sel = {"type": "single_module", "config": step_cfg}
    step_state = {"module": {**weights, "label_encoding": None}, "secondary_decoder": None}
    if wrap_multi_call:
        step_sel = {"type": "multi_call", "config": {"wrapped_step": step_sel, "config": None,
                                                     "include_multi_call_in_loss": True}}
        step_state = {"wrapped_step": step_state}
    return {"stepper": {
        "config": {"step": step_sel, "input_masking": None, "derived_forcings": {"insolation": None}},
        "dataset_info": {"horizontal_coordinates": {"lat": torch.linspace(-80, 80, 8), "lon": torch.arange(16.0) * 22.5},
                         "vertical_coordinate": None, "mask_provider": None,
                         "timestep": datetime.timedelta(hours=6) // datetime.timedelta(microseconds=1),
                         "variable_metadata": None, "gridded_operations": None, "img_shape": None, "all_labels": []},
        "step": step_state, "training_history": [{"git_sha": "abc"}],
    }}, mod


def dataclasses_asdict(dc):
    import dataclasses
    return dataclasses.asdict(dc)


@pytest.mark.parametrize("wrap", [True, False])
def test_load_stepper_from_reference_style_checkpoint(tmp_path, wrap):
    ckpt, mod = _reference_style_checkpoint(wrap_multi_call=wrap)
    path = tmp_path / "ckpt.tar"
    torch.save(ckpt, path)
    loaded = load_stepper(path, device="cpu")
    st = loaded.stepper
    assert loaded.dataset_info.img_shape == (8, 16)
    assert loaded.dataset_info.timestep == datetime.timedelta(hours=6)
    assert st._step_obj.in_names == IN and st._step_obj.out_names == OUT
    assert st.prognostic_names == ["sst", "q"]
    assert "training_history" in loaded.ignored
    ref_sd = mod.torch_module.state_dict()
    got_sd = st.modules[0].state_dict()
    assert set(ref_sd) == set(got_sd)
    for k in ref_sd:
        assert torch.equal(ref_sd[k], got_sd[k]), k
    names = sorted(set(IN) | set(OUT))
    assert float(st.normalizer.means["q"]) == pytest.approx(0.25 * (names.index("q") + 1))
    assert float(st.normalizer.stds["sst"]) == pytest.approx(1.0 + 0.5 * names.index("sst"))
    # the same dict, already loaded, and the bare stepper state are accepted too
    assert load_stepper(ckpt, device="cpu").stepper.prognostic_names == ["sst", "q"]
    assert load_stepper(ckpt["stepper"], device="cpu").stepper.prognostic_names == ["sst", "q"]


def test_load_stepper_legacy_format():
    """pre-StepSelector checkpoints (single_module.py:1370-1413): flat config + 'module' + 'normalizer' + 'img_shape'."""
    ckpt, mod = _reference_style_checkpoint(wrap_multi_call=False)
    new = ckpt["stepper"]
    step_cfg = copy.deepcopy(new["config"]["step"]["config"])
    norm = step_cfg.pop("normalization")["network"]
    for k in ("secondary_decoder", "include_channel_mask_inputs", "global_mean_removal", "input_dropout"):
        step_cfg.pop(k)
    legacy = {"config": {**step_cfg, "normalization": {"global_means_path": "x.nc", "global_stds_path": "y.nc"},
                         "loss": {"type": "LpLoss"}, "parameter_init": {}},
              "module": new["step"]["module"],
              "normalizer": {"means": {k: torch.tensor(v) for k, v in norm["means"].items()},
                             "stds": {k: torch.tensor(v) for k, v in norm["stds"].items()}},
              "img_shape": (8, 16), "encoded_timestep": 6 * 3600 * 10**6, "sigma_coordinates": {"ak": [0.0], "bk": [1.0]}}
    loaded = load_stepper({"stepper": legacy}, device="cpu")
    assert loaded.dataset_info.img_shape == (8, 16)
    assert "loss" in loaded.ignored
    sd = loaded.stepper.modules[0].state_dict()
    for k, v in mod.torch_module.state_dict().items():
        assert torch.equal(sd[k], v)


def test_checkpoint_unsupported_options_are_loud():
    ckpt, _ = _reference_style_checkpoint(corrector={**dataclasses_asdict(AtmosphereCorrectorConfig()),
                                                     "conserve_dry_air": True, "force_positive_names": ["q"]})
    with pytest.raises(NotImplementedError, match="conserve_dry_air"):
        load_stepper(ckpt, device="cpu")
    loaded = load_stepper(ckpt, device="cpu", ignore_unsupported=True)
    assert "corrector.conserve_dry_air" in loaded.ignored
    assert loaded.stepper._step_obj._corrector.force_positive_names == ["q"]
    bad = copy.deepcopy(ckpt)
    bad["stepper"]["config"]["step"]["config"]["wrapped_step"]["type"] = "something_else"
    with pytest.raises(NotImplementedError):
        load_stepper(bad, device="cpu")
    bad = copy.deepcopy(ckpt)
    bad["stepper"]["step"]["wrapped_step"]["module"].pop("module.pos_embed")
    with pytest.raises(RuntimeError):                      # strict load_state_dict, as the reference
        load_stepper(bad, device="cpu", ignore_unsupported=True)
    bad = copy.deepcopy(ckpt)
    bad["stepper"]["config"]["step"]["config"]["wrapped_step"]["config"]["surprise"] = 1
    with pytest.raises(ValueError, match="surprise"):
        load_stepper(bad, device="cpu", ignore_unsupported=True)


def test_force_positive_and_prescriber_semantics():
    g = torch.Generator().manual_seed(0)
    data = {"q": torch.randn(2, 4, 8, generator=g), "t": torch.randn(2, 4, 8, generator=g)}
    out = force_positive(data, ["q"])
    assert set(out) == {"q"} and float(out["q"].min()) >= 0.0
    assert torch.equal(out["q"], torch.clamp(data["q"], min=0.0))
    mask = torch.tensor([[0.0, 0.49, 0.5, 0.51, 1.0, 1.49, 1.5, 2.0]]).expand(4, 8)
    orig, repl = torch.zeros(4, 8), torch.ones(4, 8)
    got = replace_on_mask(orig, repl, mask, 1)
    # torch.round is round-half-to-even: 0.5 -> 0, 1.5 -> 2 (spatial_masking.py:25)
    assert got[0].tolist() == [0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 0.0]
    p = Prescriber("sst", "frac", 1, interpolate=True)
    frac = torch.rand(2, 4, 8, generator=g)
    gen = {"sst": torch.randn(2, 4, 8, generator=g), "q": data["q"]}
    tgt = {"sst": torch.randn(2, 4, 8, generator=g)}
    res = p({"frac": frac}, gen, tgt)
    assert torch.equal(res["sst"], frac * tgt["sst"] + (1 - frac) * gen["sst"]) and res["q"] is gen["q"]
    with pytest.raises(ValueError):
        Prescriber("sst", "frac", 0, interpolate=True)
    with pytest.raises(ValueError):
        p({"frac": frac}, {"q": data["q"]}, tgt)


def test_step_hook_order_corrector_then_ocean_then_prescribed():
    """single_module.py:669-716: denormalise -> corrector -> ocean -> prescribed prognostics; the ocean reads its mask
    and target SST from next_step_input_data."""
    from ace_amd.step import SingleModuleStep

    class Net(torch.nn.Module):      # (B,4,H,W) -> (B,3,H,W): sst' = sst - 10, q' = -|q| - 1 (negative), d = f
        def forward(self, x):
            return torch.stack([x[:, 1] - 10.0, -x[:, 2].abs() - 1.0, x[:, 0]], dim=1)

    names = sorted(set(IN) | set(OUT))
    norm = NormalizationConfig(means={n: 0.0 for n in names}, stds={n: 1.0 for n in names})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config=dict(NET)),
        in_names=IN, out_names=OUT, normalization=norm,
        ocean={"surface_temperature_name": "sst", "ocean_fraction_name": "frac", "interpolate": False, "slab": None},
        corrector={"force_positive_names": ["q"]})
    assert set(cfg.next_step_input_names) == {"f", "frac", "sst"}          # ocean forcing names join the forcing set
    step = SingleModuleStep(cfg, ace_amd.DatasetInfo((4, 8)), cfg.normalization.build(names), device="cpu")
    step.module = Module(Net(), None)
    g = torch.Generator().manual_seed(1)
    inp = {n: torch.randn(2, 4, 8, generator=g) for n in IN}
    frac = (torch.rand(2, 4, 8, generator=g) > 0.5).float()
    nxt = {"f": torch.randn(2, 4, 8, generator=g), "frac": frac, "sst": torch.full((2, 4, 8), 300.0)}
    out = step.step(ace_amd.StepArgs(inp, nxt)).output
    assert float(out["q"].max()) == 0.0                                    # network output was negative everywhere
    assert torch.equal(out["sst"], torch.where(frac == 1, nxt["sst"], inp["sst"] - 10.0))
    assert torch.equal(out["d"], inp["f"])
    slab = ace_amd.SingleModuleStepConfig(builder=cfg.builder, in_names=IN, out_names=OUT, normalization=norm,
                                          ocean={"surface_temperature_name": "sst", "ocean_fraction_name": "frac",
                                                 "slab": {"q_flux_name": "qf", "mixed_layer_depth_name": "mld"}})
    assert slab.ocean.is_slab and {"qf", "mld", "frac"} == set(slab.ocean.forcing_names)   # the checkpoint form of the slab ocean
    with pytest.raises(NotImplementedError):
        ace_amd.SingleModuleStepConfig(builder=cfg.builder, in_names=IN, out_names=OUT, normalization=norm,
                                       corrector={"conserve_dry_air": True}).get_step(ace_amd.DatasetInfo((4, 8)))


# ---- format pinned on the reference itself: tests/golden/gen_checkpoint.pt holds Stepper.get_state() of REAL reference
# steppers (tests/golden/make_golden_checkpoint.py, oracle/ref_loader.load_stepper_ref)
def _golden_checkpoint():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_checkpoint.pt")
    return torch.load(path, map_location="cpu", weights_only=True)


def test_reference_checkpoint_state_is_ingested():
    g = _golden_checkpoint()["ace2_like"]
    state = g["state"]
    assert set(state) == {"config", "dataset_info", "step", "training_history"}
    loaded = load_stepper({"stepper": state}, device="cpu")
    assert loaded.ignored == []
    ref_cfg = state["config"]["step"]["config"]
    cfg = loaded.config
    assert cfg.in_names == ref_cfg["in_names"] and cfg.out_names == ref_cfg["out_names"]
    assert cfg.next_step_forcing_names == ["DSWRFtoa"]
    assert cfg.ocean.surface_temperature_name == "surface_temperature" and cfg.ocean.ocean_fraction_name == "ocean_fraction"
    c = cfg.corrector
    assert c.conserve_dry_air and c.moisture_budget_correction == "advection_and_precipitation"
    assert c.total_energy_budget_correction.method == "constant_temperature"
    assert c.total_energy_budget_correction.constant_unaccounted_heating == pytest.approx(0.1)
    assert c.force_positive_names == ref_cfg["corrector"]["config"]["force_positive_names"]
    # every default the reference serialises for the builder is understood by the native builder
    assert cfg.builder.type == "SphericalFourierNeuralOperatorNet"
    for k, v in ref_cfg["builder"]["config"].items():
        assert cfg.builder.config[k] == v, k
    # weights: the reference's names (with the wrapper's "module." prefix) map one-to-one onto the native module
    ref_w = {k[len("module."):]: v for k, v in state["step"]["module"].items() if k != "label_encoding"}
    got_w = loaded.stepper.modules[0].state_dict()
    got_w = {k[len("module."):] if k.startswith("module.") else k: v for k, v in got_w.items()}
    assert set(ref_w) == set(got_w)
    for k in ref_w:
        assert torch.equal(ref_w[k], got_w[k].cpu()), k
    # normaliser and geometry
    norm = loaded.stepper._step_obj.normalizer
    for n, m in ref_cfg["normalization"]["network"]["means"].items():
        assert float(norm.means[n]) == pytest.approx(m, rel=1e-7)
        assert float(norm.stds[n]) == pytest.approx(ref_cfg["normalization"]["network"]["stds"][n], rel=1e-7)
    info = loaded.dataset_info
    assert info.img_shape == (8, 16) and info.timestep == datetime.timedelta(hours=6)
    assert torch.equal(info.vertical_coordinate.ak, state["dataset_info"]["vertical_coordinate"]["ak"])
    assert torch.equal(info.vertical_coordinate.bk, state["dataset_info"]["vertical_coordinate"]["bk"])


def test_reference_multi_call_noise_conditioned_checkpoint_is_ingested():
    state = _golden_checkpoint()["multi_call_csfno"]["state"]
    assert state["config"]["step"]["type"] == "multi_call" and list(state["step"]) == ["wrapped_step"]
    loaded = load_stepper(state, device="cpu")
    assert loaded.config.builder.type == "NoiseConditionedSFNO"
    ref_w = {k[len("module."):]: v for k, v in state["step"]["wrapped_step"]["module"].items() if k != "label_encoding"}
    got_w = loaded.stepper.modules[0].state_dict()
    got_w = {k[len("module."):] if k.startswith("module.") else k: v for k, v in got_w.items()}
    assert set(ref_w) == set(got_w)
    for k in ref_w:
        assert torch.equal(ref_w[k], got_w[k].cpu()), k


def test_stepper_override_config():
    """load_stepper(path, StepperOverrideConfig(...)) / apply_stepper_override (single_module.py:1848-1960)."""
    from ace_amd.checkpoint import StepperOverrideConfig, apply_stepper_override
    state = _golden_checkpoint()["ace2_like"]["state"]
    kept = load_stepper(state, device="cpu")
    assert kept.config.ocean is not None and kept.stepper.get_prescribed_prognostic_names() == []
    over = load_stepper(state, StepperOverrideConfig(ocean=None, prescribed_prognostic_names=["surface_temperature"]),
                        device="cpu")
    assert over.config.ocean is None and over.stepper._step_obj._ocean is None
    assert over.stepper.get_prescribed_prognostic_names() == ["surface_temperature"]
    assert "surface_temperature" in over.stepper._step_obj.next_step_input_names
    assert "ocean_fraction" in over.config.in_names          # still a network input: only the SST prescription is gone
    for k, v in kept.stepper.modules[0].state_dict().items():     # the weights are untouched
        assert torch.equal(v, over.stepper.modules[0].state_dict()[k])
    apply_stepper_override(over.stepper, StepperOverrideConfig(
        ocean={"surface_temperature_name": "surface_temperature", "ocean_fraction_name": "ocean_fraction"}))
    assert over.stepper._step_obj._ocean is not None
    with pytest.raises(ValueError, match="must be in out_names"):
        over.stepper.replace_prescribed_prognostic_names(["HGTsfc"])
    apply_stepper_override(over.stepper, StepperOverrideConfig(derived_forcings={"insolation": None}))
    assert not over.stepper.forcing_deriver.needs_time
    with pytest.raises(ValueError, match="missing"):
        apply_stepper_override(over.stepper, StepperOverrideConfig(multi_call={"forcing_name": "co2"}))
    apply_stepper_override(over.stepper, StepperOverrideConfig(multi_call=None))


def test_checkpoint_with_derived_insolation():
    """StepperConfig.derived_forcings of a checkpoint (single_module.py:532-539): the insolation is computed from the window's
    time axis; the override may change its parameters but not its name (derived_forcings.py:44-62)."""
    from ace_amd.checkpoint import StepperOverrideConfig, apply_stepper_override
    from ace_amd.timeaxis import TimeAxis
    ckpt, _ = _reference_style_checkpoint()
    name = IN[0] if IN[0] not in OUT else [n for n in IN if n not in OUT][0]
    ckpt["stepper"]["config"]["derived_forcings"] = {"insolation": {
        "insolation_name": name, "solar_constant": {"value": 1360.0, "dtype": "float32"}, "obliquity": 23.439,
        "eccentricity": 0.0167, "longitude_of_perhelion": 102.932}}
    loaded = load_stepper(ckpt, device="cpu")
    st = loaded.stepper
    assert st.forcing_deriver.needs_time and st.derived_forcings.insolation.insolation_name == name
    assert name not in st.forcing_names_from_data()
    assert set(st.forcing_names_from_data()) == set(n for n in IN if n not in OUT) - {name}
    time = TimeAxis.regular((2020, 1, 1), datetime.timedelta(hours=6), 3, 2)
    win = st.forcing_deriver({}, time)
    assert win[name].shape == (2, 3, 8, 16) and win[name].max() > 500
    apply_stepper_override(st, StepperOverrideConfig(derived_forcings={"insolation": {
        "insolation_name": name, "solar_constant": {"value": 1370.0}}}))
    assert st.forcing_deriver({}, time)[name].max() > win[name].max()
    with pytest.raises(ValueError, match="insolation_name"):
        apply_stepper_override(st, StepperOverrideConfig(derived_forcings={"insolation": {
            "insolation_name": "something_else", "solar_constant": {"value": 1370.0}}}))
    # without the grid's coordinates the insolation cannot be computed: loud, or dropped on request
    bare = copy.deepcopy(ckpt)
    bare["stepper"]["dataset_info"]["horizontal_coordinates"] = None
    bare["stepper"]["dataset_info"]["img_shape"] = (8, 16)
    with pytest.raises(NotImplementedError, match="latitudes"):
        load_stepper(bare, device="cpu")
    assert "derived_forcings" in load_stepper(bare, device="cpu", ignore_unsupported=True).ignored


@pytest.mark.parametrize("case", ["ace2_like", "residual_prescribed", "ace2_like_override"])
def test_oracle_rollout_restates_the_reference_stepper(case):
    """oracle network + host step logic (normalise, residual, corrector, ocean, prescribed prognostics) in fp32 on CPU
    against the rollout the REAL reference stepper produced for the fixture: identical to the last bit in the build
    container (same torch CPU kernels, same operation order); held to 1e-6 of the field maximum here so that a
    different CPU / thread count cannot fail it - except the ill-conditioned advective tendency (see conditioning_floor)."""
    from _util import checkpoint_case, conditioning_floor, oracle_checkpoint_rollout
    g = checkpoint_case(_golden_checkpoint(), case)
    got = oracle_checkpoint_rollout(g, torch.float32)
    floor = conditioning_floor(g)
    nbit = 0
    for s, want_all in enumerate(g["steps"]):
        assert set(want_all) == set(got[s])
        for k, want in want_all.items():
            err = float((got[s][k] - want).abs().max() / want.abs().max())
            assert err <= max(1e-6, 3.0 * floor[s][k]), (k, s, err)
            nbit += int(torch.equal(got[s][k], want))
    print(f"{case}: {nbit} of {sum(len(x) for x in g['steps'])} (step, field) pairs bit-identical")
