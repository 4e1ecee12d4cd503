"""The post-step atmosphere corrector against golden vectors emitted by the reference itself
(tests/golden/make_golden_corrector.py -> gen_corrector.pt): every option of AtmosphereCorrectorConfig
(fme/core/corrector/atmosphere.py:223-398), two consecutive steps (the dry-air reference mass is seeded on the first and
carried in the corrector state), and the step-level plumbing (StepperState)."""
import datetime
import os

import pytest
import torch

import ace_amd
from ace_amd.corrector import AtmosphereCorrectorConfig, CorrectorState

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_corrector.pt")
CONFIGS = {
    "force_positive": dict(force_positive_names=["PRATEsfc", "specific_total_water_0"]),
    "dry_air": dict(conserve_dry_air=True),
    "zero_advection": dict(zero_global_mean_moisture_advection=True),
    "moisture_precipitation": dict(moisture_budget_correction="precipitation"),
    "moisture_evaporation": dict(moisture_budget_correction="evaporation"),
    "moisture_advection_and_precipitation": dict(moisture_budget_correction="advection_and_precipitation",
                                                 clip_frozen_precipitation=True),
    "moisture_advection_and_evaporation": dict(moisture_budget_correction="advection_and_evaporation"),
    "energy": dict(total_energy_budget_correction={"method": "constant_temperature", "constant_unaccounted_heating": 0.3}),
    "ace2_like": dict(conserve_dry_air=True, moisture_budget_correction="advection_and_precipitation",
                      force_positive_names=["PRATEsfc", "specific_total_water_0", "specific_total_water_1"],
                      total_energy_budget_correction={"method": "constant_temperature"}, clip_frozen_precipitation=True),
}


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


def _info(g):
    return ace_amd.DatasetInfo((8, 16), timestep=datetime.timedelta(seconds=g["timestep_seconds"]), lat=g["lat"],
                               lon=g["lon"], ak=g["ak"], bk=g["bk"])


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_corrector_matches_reference(gold, name):
    corrector = AtmosphereCorrectorConfig.from_state(CONFIGS[name]).get_corrector(_info(gold))
    exp = gold["expected"][name]
    out0, st0 = corrector(gold["input0"], gold["gen0"], gold["forcing"], None)
    assert set(out0) == set(exp["step0"])
    for k, v in exp["step0"].items():
        torch.testing.assert_close(out0[k], v, rtol=1e-6, atol=0.0, msg=lambda m: f"{name} step0 {k}: {m}")
    if exp["global_dry_air_mass"] is None:
        assert st0 is None or st0.global_dry_air_mass is None
    else:
        assert st0.global_dry_air_mass.dtype == torch.float64
        torch.testing.assert_close(st0.global_dry_air_mass, exp["global_dry_air_mass"], rtol=1e-12, atol=0.0)
    out1, st1 = corrector({**out0, **gold["forcing"]}, gold["gen1"], gold["forcing"], st0)
    for k, v in exp["step1"].items():
        torch.testing.assert_close(out1[k], v, rtol=1e-6, atol=0.0, msg=lambda m: f"{name} step1 {k}: {m}")
    if st0 is not None and st0.global_dry_air_mass is not None:
        assert st1.global_dry_air_mass is st0.global_dry_air_mass          # the reference mass is never re-seeded


def test_conservation_properties(gold):
    """what the corrections are for: the global dry-air mass stays at its initial value and the global moisture budget
    closes (size-independent properties, checked here on the golden inputs)."""
    from ace_amd.atmosphere import AreaWeightedMean, AtmosphereData, HybridSigmaPressureCoordinate
    info = _info(gold)
    mean = AreaWeightedMean(info.area_weights)
    vc = HybridSigmaPressureCoordinate(gold["ak"], gold["bk"])
    c = AtmosphereCorrectorConfig(conserve_dry_air=True, moisture_budget_correction="advection_and_precipitation"
                                  ).get_corrector(info)
    out, st = c(gold["input0"], gold["gen0"], gold["forcing"], None)
    dry_in = mean(AtmosphereData(gold["input0"], vc).surface_pressure_due_to_dry_air.double())
    dry_out = mean(AtmosphereData(out, vc).surface_pressure_due_to_dry_air.double())
    torch.testing.assert_close(dry_out, dry_in, rtol=1e-6, atol=0.0)
    a_in, a_out = AtmosphereData(gold["input0"], vc), AtmosphereData(out, vc)
    tend = mean((a_out.total_water_path - a_in.total_water_path) / gold["timestep_seconds"])
    budget = mean(a_out.evaporation_rate - a_out.precipitation_rate)
    torch.testing.assert_close(tend, budget, rtol=2e-3, atol=1e-9)
    torch.testing.assert_close(mean(a_out.tendency_of_total_water_path_due_to_advection),
                               torch.zeros(2), rtol=0.0, atol=2e-8)


def test_corrector_needs_geometry_and_is_loud(gold):
    cfg = AtmosphereCorrectorConfig(conserve_dry_air=True, zero_global_mean_moisture_advection=True)
    with pytest.raises(NotImplementedError, match="conserve_dry_air"):
        cfg.get_corrector(ace_amd.DatasetInfo((8, 16)))
    only_lat = ace_amd.DatasetInfo((8, 16), lat=gold["lat"], lon=gold["lon"])
    assert cfg.unsupported(only_lat) == ["conserve_dry_air"]            # the advection fix needs area weights only
    c = cfg.get_corrector(only_lat, ignore_unsupported=True)
    assert c.corrections == ["zero_global_mean_moisture_advection"]
    with pytest.raises(ValueError):
        AtmosphereCorrectorConfig(moisture_budget_correction="rain")
    with pytest.raises(ValueError):
        ace_amd.DatasetInfo((8, 16), lat=torch.zeros(5))
    assert AtmosphereCorrectorConfig().get_corrector(ace_amd.DatasetInfo((8, 16))) is None


def test_step_threads_corrector_state(gold):
    """SingleModuleStep applies the corrector after denormalisation and threads its state through StepperState
    (fme/core/step/single_module.py:669-693): two steps through the Stepper equal two direct corrector calls."""
    from ace_amd.registry import Module
    from ace_amd.step import NormalizationConfig, SingleModuleStep
    prog = sorted(gold["gen0"])
    names = prog + ["HGTsfc", "DSWRFtoa"]
    in_names, out_names = ["HGTsfc", "DSWRFtoa"] + prog, prog

    class Replay(torch.nn.Module):       # ignores its input: emits the golden 'generated' states in turn
        def __init__(self, gens):
            super().__init__()
            self.gens, self.i = gens, 0

        def forward(self, x):
            gen = self.gens[self.i]
            self.i += 1
            return torch.stack([gen[n] for n in out_names], dim=1)

    norm = NormalizationConfig(means={n: 0.0 for n in names}, stds={n: 1.0 for n in names})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 8, "num_layers": 1}),
        in_names=in_names, out_names=out_names, normalization=norm, corrector=CONFIGS["ace2_like"])
    step = SingleModuleStep(cfg, _info(gold), cfg.normalization.build(names), device="cpu")
    step.module = Module(Replay([gold["gen0"], gold["gen1"]]), None)
    stepper = ace_amd.Stepper(step)
    ic = {n: gold["input0"][n][:, None] for n in prog}
    forcing = {n: torch.stack([gold["forcing"][n]] * 3, dim=1) for n in ("HGTsfc", "DSWRFtoa")}
    outs = list(stepper.predict_generator(ic, forcing, 2))
    exp = gold["expected"]["ace2_like"]
    for k in out_names:
        torch.testing.assert_close(outs[0].output[k], exp["step0"][k], rtol=1e-6, atol=0.0)
        torch.testing.assert_close(outs[1].output[k], exp["step1"][k], rtol=1e-6, atol=0.0)
    assert isinstance(outs[1].stepper_state.corrector_state, CorrectorState)
    torch.testing.assert_close(outs[1].stepper_state.corrector_state.global_dry_air_mass, exp["global_dry_air_mass"])


@pytest.mark.parametrize("interpolate", [False, True])
def test_ocean_matches_reference(gold, interpolate):
    """prescribed SST (fme/core/ocean.py:167-215) incl. the half-to-even rounding of the mask at 0.5 / 1.5"""
    from ace_amd.ocean import OceanConfig
    o = gold["ocean"]
    cfg = OceanConfig(surface_temperature_name="sst", ocean_fraction_name="frac", interpolate=interpolate)
    assert sorted(cfg.forcing_names) == o["forcing_names"]
    out = cfg.build(["sst", "frac", "q"], ["sst", "q"])(o["input"], o["gen"], o["target"])
    assert set(out) == set(o["expected"][interpolate])
    for k, v in o["expected"][interpolate].items():
        assert torch.equal(out[k], v), k


@pytest.mark.parametrize("interpolate", [False, True])
def test_slab_ocean_matches_reference(gold, interpolate):
    """slab ocean (fme/core/ocean.py:14-29, 64-92, 233-254): SST_in + (F_net + Q) / (rho c_p depth) dt over ocean, F_net the
    generated net surface energy flux without the frozen-precipitation term - bitwise against the reference's own output;
    configuration from dataclasses and from the state-dict form a checkpoint holds."""
    import datetime
    from ace_amd.ocean import OceanConfig, SlabOceanConfig
    o = gold["slab_ocean"]
    dt = datetime.timedelta(seconds=o["timestep_seconds"])
    flux = ["DLWRFsfc", "ULWRFsfc", "DSWRFsfc", "USWRFsfc", "LHTFLsfc", "SHTFLsfc"]
    for cfg in (OceanConfig("sst", "frac", interpolate, SlabOceanConfig(mixed_layer_depth_name="mld", q_flux_name="qflux")),
                OceanConfig.from_state({"surface_temperature_name": "sst", "ocean_fraction_name": "frac", "interpolate": interpolate,
                                        "slab": {"mixed_layer_depth_name": "mld", "q_flux_name": "qflux"}})):
        assert cfg.is_slab and sorted(cfg.forcing_names) == o["forcing_names"]
        ocean = cfg.build(["sst", "frac", "q"], ["sst", "q"] + flux, dt)
        out = ocean(o["input"], o["gen"], o["target"])
        assert set(out) == set(o["expected"][interpolate])
        for k, v in o["expected"][interpolate].items():
            assert torch.equal(out[k], v), k
    with pytest.raises(ValueError):
        cfg.build(["sst", "frac"], ["sst"] + flux, None)               # no timestep
    with pytest.raises(ValueError):
        OceanConfig.from_state({"surface_temperature_name": "sst", "ocean_fraction_name": "frac", "slab": {"depth": "mld"}})


def test_slab_ocean_through_the_step_config():
    """SingleModuleStepConfig accepts the slab form (state-dict as in a checkpoint) and asks for its forcings."""
    import ace_amd
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 8, "num_layers": 1}),
        in_names=["sst", "frac", "a"], out_names=["sst", "a"],
        normalization=ace_amd.step.NormalizationConfig(means={k: 0.0 for k in ("sst", "frac", "a")}, stds={k: 1.0 for k in ("sst", "frac", "a")}),
        ocean={"surface_temperature_name": "sst", "ocean_fraction_name": "frac", "slab": {"mixed_layer_depth_name": "mld", "q_flux_name": "qflux"}})
    assert cfg.ocean.is_slab
    assert {"mld", "qflux", "frac"} <= set(cfg.get_next_step_forcing_names()) | set(cfg.ocean.forcing_names)
