"""Golden checkpoints emitted by the REAL reference stepper (fme/ace/stepper/single_module.py ``Stepper``), imported
under stubs (oracle/ref_loader.load_stepper_ref) - build container only.  Writes tests/golden/gen_checkpoint.pt:

  "ace2_like"   ``Stepper.get_state()`` of a small SphericalFourierNeuralOperatorNet stepper with the ACE2-style
                atmosphere corrector (dry air, moisture budget, energy budget, positivity), a prescribed-SST ocean and a
                next-step forcing, the (initial condition, forcing) it was run on and the reference's own
                ``predict_generator`` output of every step of a 3-step rollout on CPU
  "residual_prescribed"  the same for residual_prediction=True with a prescribed prognostic (equiangular data grid, no
                big skip, no position embedding, no hooks)
  "ace2_like_override"  the first stepper re-loaded with ``StepperOverrideConfig(ocean=None, prescribed_prognostic_names=
                ["surface_temperature"])`` and its 3-step rollout
  "multi_call_csfno"  the state of a multi_call-wrapped NoiseConditionedSFNO stepper (state ingestion only: its
                rollout draws noise from the global torch RNG)

Data only: plain dicts / lists / numbers / tensors, loadable with ``torch.load(weights_only=True)``."""
import datetime
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

NZ, B, H, W, T = 2, 2, 8, 16, 3
SCALES = {"PRESsfc": (98000.0, 1500.0), "HGTsfc": (200.0, 300.0), "DSWRFtoa": (340.0, 50.0), "PRATEsfc": (3e-5, 2e-5),
          "LHTFLsfc": (80.0, 40.0), "SHTFLsfc": (20.0, 15.0), "tendency_of_total_water_path_due_to_advection": (0.0, 2e-5),
          "DSWRFsfc": (180.0, 40.0), "USWRFsfc": (30.0, 10.0), "DLWRFsfc": (330.0, 30.0), "ULWRFsfc": (390.0, 30.0),
          "ULWRFtoa": (240.0, 20.0), "USWRFtoa": (100.0, 15.0), "surface_temperature": (288.0, 8.0),
          "ocean_fraction": (0.5, 0.3)}
for k in range(NZ):
    SCALES[f"specific_total_water_{k}"] = (2e-3 * (k + 1), 5e-4)
    SCALES[f"air_temperature_{k}"] = (220.0 + 30.0 * k, 5.0)

FORCING = ["DSWRFtoa", "HGTsfc", "ocean_fraction"]
PROGNOSTIC = ["PRESsfc", "surface_temperature"] + [f"{p}_{k}" for p in ("specific_total_water", "air_temperature") for k in range(NZ)]
DIAGNOSTIC = ["PRATEsfc", "LHTFLsfc", "SHTFLsfc", "tendency_of_total_water_path_due_to_advection", "DSWRFsfc", "USWRFsfc",
              "DLWRFsfc", "ULWRFsfc", "ULWRFtoa", "USWRFtoa"]


def field(g, name, *lead):
    mean, std = SCALES[name]
    x = torch.randn(*lead, H, W, generator=g) * std + mean
    return x.clamp(0.0, 1.0) if name == "ocean_fraction" else x


def step_config(builder, **extra):
    in_names, out_names = FORCING + PROGNOSTIC, PROGNOSTIC + DIAGNOSTIC
    names = sorted(set(in_names + out_names))
    return {"type": "single_module", "config": dict(
        builder=builder, in_names=in_names, out_names=out_names,
        # the network's (normalised) output is O(1): keep the de-normalised fields physical
        normalization={"network": {"means": {n: SCALES[n][0] for n in names},
                                   "stds": {n: 0.2 * SCALES[n][1] for n in names}}}, **extra)}


def plain(o):
    """dataclasses.asdict output is already plain; make sure nothing but data is pickled."""
    if isinstance(o, dict):
        return {k: plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    if isinstance(o, torch.Tensor):
        return o.detach().clone()
    assert o is None or isinstance(o, (bool, int, float, str)), type(o)
    return o


def main():
    ref = ref_loader.load_stepper_ref()
    lat = torch.linspace(-78.75, 78.75, H)
    lon = torch.arange(float(W)) * (360.0 / W)
    info = ref.DatasetInfo(
        horizontal_coordinates=ref.LatLonCoordinates(lat=lat, lon=lon),
        vertical_coordinate=ref.HybridSigmaPressureCoordinate(ak=torch.tensor([100.0, 8000.0, 0.0]),
                                                              bk=torch.tensor([0.0, 0.3, 1.0])),
        timestep=datetime.timedelta(hours=6))
    out = {}

    # ---- ACE2-like stepper with a rollout
    sfno = {"type": "SphericalFourierNeuralOperatorNet",
            "config": {"embed_dim": 16, "num_layers": 2, "operator_type": "dhconv", "scale_factor": 1,
                       "filter_type": "linear", "data_grid": "legendre-gauss"}}
    cfg = {"step": step_config(
        sfno, next_step_forcing_names=["DSWRFtoa"],
        ocean={"surface_temperature_name": "surface_temperature", "ocean_fraction_name": "ocean_fraction"},
        corrector={"type": "atmosphere_corrector", "config": {
            "conserve_dry_air": True, "moisture_budget_correction": "advection_and_precipitation",
            "force_positive_names": ["PRATEsfc"] + [f"specific_total_water_{k}" for k in range(NZ)],
            "total_energy_budget_correction": {"method": "constant_temperature", "constant_unaccounted_heating": 0.1}}})}
    torch.manual_seed(0)
    stepper = ref.StepperConfig.from_stepper_state({"config": cfg}).get_stepper(dataset_info=info)
    with torch.no_grad():       # the reference's zero-initialised biases / unit norm weights: make them non-trivial
        g = torch.Generator().manual_seed(1)
        for p in stepper.modules.parameters():
            if p.ndim <= 1 or p.abs().max() == 0:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    g = torch.Generator().manual_seed(2)
    ic = {n: field(g, n, B, 1) for n in PROGNOSTIC}
    forcing = {n: field(g, n, B, T + 1) for n in FORCING}
    forcing["HGTsfc"] = forcing["HGTsfc"][:, :1].expand(B, T + 1, H, W).clone()
    forcing["surface_temperature"] = field(g, "surface_temperature", B, T + 1)      # the ocean's prescribed SST target
    steps = []
    with torch.no_grad():
        for res in stepper.predict_generator(ic, forcing, T, ref.NullOptimization(), labels=None):
            steps.append({k: v.clone() for k, v in res.output.items()})
    for k in steps[0]:
        assert all(torch.isfinite(s[k]).all() for s in steps), k
    out["ace2_like"] = {"state": plain(stepper.get_state()), "ic": ic, "forcing": forcing, "steps": steps}
    print("ace2_like: outputs", sorted(steps[0]), "| PRESsfc step2 mean", float(steps[-1]["PRESsfc"].mean()))

    # ---- residual prediction + a prescribed prognostic, equiangular data grid, no hooks
    sfno_eq = {"type": "SphericalFourierNeuralOperatorNet",
               "config": {"embed_dim": 12, "num_layers": 2, "operator_type": "dhconv", "data_grid": "equiangular",
                          "big_skip": False, "pos_embed": False}}
    cfg3 = {"step": step_config(sfno_eq, residual_prediction=True, prescribed_prognostic_names=["surface_temperature"])}
    torch.manual_seed(4)
    stepper3 = ref.StepperConfig.from_stepper_state({"config": cfg3}).get_stepper(dataset_info=info)
    steps3 = []
    with torch.no_grad():
        for res in stepper3.predict_generator(ic, forcing, T, ref.NullOptimization(), labels=None):
            steps3.append({k: v.clone() for k, v in res.output.items()})
    out["residual_prescribed"] = {"state": plain(stepper3.get_state()), "ic": ic, "forcing": forcing, "steps": steps3}
    print("residual_prescribed: PRESsfc step2 mean", float(steps3[-1]["PRESsfc"].mean()))

    # ---- inference-time override of the first stepper (load_stepper(path, StepperOverrideConfig(...)), single_module.py:1909-1960):
    # no ocean, the surface temperature prescribed from the forcing record instead
    sm = ref.module
    over = {"ocean": None, "prescribed_prognostic_names": ["surface_temperature"]}
    stepper4 = sm.Stepper.from_state(out["ace2_like"]["state"])
    sm.apply_stepper_override(stepper4, sm.StepperOverrideConfig(**over))
    steps4 = []
    with torch.no_grad():
        for res in stepper4.predict_generator(ic, forcing, T, ref.NullOptimization(), labels=None):
            steps4.append({k: v.clone() for k, v in res.output.items()})
    out["ace2_like_override"] = {"override": over, "steps": steps4}
    print("ace2_like_override: max |dT_sfc| vs no override",
          float((steps4[-1]["air_temperature_1"] - steps[-1]["air_temperature_1"]).abs().max()))

    # ---- multi_call-wrapped noise-conditioned stepper (state only)
    csfno = {"type": "NoiseConditionedSFNO",
             "config": {"embed_dim": 8, "num_layers": 2, "noise_embed_dim": 4, "noise_type": "isotropic",
                        "filter_num_groups": 2}}
    cfg2 = {"step": {"type": "multi_call", "config": {"wrapped_step": step_config(csfno), "config": None,
                                                      "include_multi_call_in_loss": False}}}
    torch.manual_seed(3)
    stepper2 = ref.StepperConfig.from_stepper_state({"config": cfg2}).get_stepper(dataset_info=info)
    out["multi_call_csfno"] = {"state": plain(stepper2.get_state())}
    print("multi_call_csfno: step state keys", list(out["multi_call_csfno"]["state"]["step"].keys()))

    path = os.path.join(HERE, "gen_checkpoint.pt")
    torch.save(out, path)
    torch.load(path, weights_only=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
