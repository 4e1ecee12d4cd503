"""Golden vectors for the post-step corrector, emitted by the REAL reference (fme/core/corrector/atmosphere.py) imported
under stubs (oracle/ref_loader.load_corrector) - build container only.  Writes tests/golden/gen_corrector.pt:
inputs of a small synthetic atmosphere (2 samples, 8 x 16 grid, 4 layers) and, per corrector configuration, the
reference's corrected fields for two consecutive steps (the second step re-uses the dry-air state of the first)."""
import datetime
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

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


def synthetic(seed, B=2, H=8, W=16, NZ=4):
    g = torch.Generator().manual_seed(seed)
    r = lambda scale=1.0, shift=0.0: torch.randn(B, H, W, generator=g) * scale + shift
    d = {"PRESsfc": r(1500.0, 98000.0), "HGTsfc": r(300.0, 200.0), "DSWRFtoa": r(50.0, 340.0).abs(),
         "PRATEsfc": (r(2e-5, 3e-5)), "LHTFLsfc": r(40.0, 80.0), "SHTFLsfc": r(15.0, 20.0),
         "tendency_of_total_water_path_due_to_advection": r(2e-5),
         "total_frozen_precipitation_rate": r(2e-5, 1e-5).abs(),
         "DSWRFsfc": r(40.0, 180.0).abs(), "USWRFsfc": r(10.0, 30.0).abs(), "DLWRFsfc": r(30.0, 330.0),
         "ULWRFsfc": r(30.0, 390.0), "ULWRFtoa": r(20.0, 240.0), "USWRFtoa": r(15.0, 100.0).abs()}
    for k in range(NZ):
        d[f"specific_total_water_{k}"] = r(1e-3, 2e-3 * (k + 1))
        d[f"air_temperature_{k}"] = r(5.0, 220.0 + 20.0 * k)
    return d


def main():
    ref = ref_loader.load_corrector()
    lat = torch.linspace(-78.75, 78.75, 8)
    lon = torch.arange(16.0) * 22.5
    ak = torch.tensor([100.0, 5000.0, 12000.0, 6000.0, 0.0])
    bk = torch.tensor([0.0, 0.05, 0.35, 0.75, 1.0])
    vc = ref.HybridSigmaPressureCoordinate(ak=ak, bk=bk)
    ops = ref.LatLonCoordinates(lat=lat, lon=lon).get_gridded_operations()
    timestep = datetime.timedelta(hours=6)
    inp0, gen0, gen1 = synthetic(1), synthetic(2), synthetic(3)
    forcing = {"DSWRFtoa": synthetic(4)["DSWRFtoa"], "HGTsfc": inp0["HGTsfc"]}
    # forcing-only fields are not generated; the step input is (state, forcing) as in Stepper.predict_generator
    for d in (gen0, gen1):
        del d["DSWRFtoa"], d["HGTsfc"]
    inp0 = {**inp0, **forcing}
    out = {"lat": lat, "lon": lon, "ak": ak, "bk": bk, "timestep_seconds": timestep.total_seconds(),
           "input0": inp0, "gen0": gen0, "gen1": gen1, "forcing": forcing, "expected": {}}
    for name, kw in CONFIGS.items():
        kw = dict(kw)
        if "total_energy_budget_correction" in kw:
            kw["total_energy_budget_correction"] = ref.EnergyBudgetConfig(**kw["total_energy_budget_correction"])
        corrector = ref.AtmosphereCorrectorConfig(**kw)._build(ops, vc, timestep)
        r0 = corrector(inp0, gen0, forcing, None)
        # second step: the corrected output of step 0 is the input, the state carries the dry-air reference
        r1 = corrector({**r0.corrected, **forcing}, gen1, forcing, r0.corrector_state)
        mass = None if r0.corrector_state is None else r0.corrector_state.global_dry_air_mass
        out["expected"][name] = {"step0": {k: v.clone() for k, v in r0.corrected.items()},
                                 "step1": {k: v.clone() for k, v in r1.corrected.items()},
                                 "global_dry_air_mass": mass}
        changed = sorted(k for k in gen0 if not torch.equal(r0.corrected[k], gen0[k]))
        print(f"{name}: modified {changed}")
    # prescribed-SST ocean (fme/core/ocean.py:95-222) from the real reference
    import importlib
    ocean_mod = importlib.import_module("fme.core.ocean")
    g = torch.Generator().manual_seed(7)
    frac = torch.rand(2, 8, 16, generator=g)
    frac[0, :2] = 0.5      # ties: torch.round is half-to-even
    frac[1, :2] = 1.5
    oc_in = {"sst": torch.randn(2, 8, 16, generator=g) + 290.0}
    oc_gen = {"sst": torch.randn(2, 8, 16, generator=g) + 288.0, "q": torch.randn(2, 8, 16, generator=g)}
    oc_target = {"sst": torch.randn(2, 8, 16, generator=g) + 285.0, "frac": frac}
    out["ocean"] = {"input": oc_in, "gen": oc_gen, "target": oc_target, "expected": {}}
    for interp in (False, True):
        ocean = ocean_mod.OceanConfig(surface_temperature_name="sst", ocean_fraction_name="frac", interpolate=interp
                                      ).build(["sst", "frac", "q"], ["sst", "q"], timestep)
        out["ocean"]["expected"][interp] = {k: v.clone() for k, v in ocean(oc_in, oc_gen, oc_target).items()}
        out["ocean"].setdefault("forcing_names", sorted(ocean.forcing_names))
    # slab ocean (fme/core/ocean.py:14-29, 64-92, 233-254): SST_next = SST_in + (F_net + Q) / (rho c_p depth) * dt over ocean
    flux_names = ["DLWRFsfc", "ULWRFsfc", "DSWRFsfc", "USWRFsfc", "LHTFLsfc", "SHTFLsfc"]
    sl_gen = dict(oc_gen)
    for i, n in enumerate(flux_names):
        sl_gen[n] = 150.0 + 40.0 * torch.randn(2, 8, 16, generator=g) + 10.0 * i
    sl_target = {"frac": frac, "mld": 20.0 + 60.0 * torch.rand(2, 8, 16, generator=g), "qflux": 15.0 * torch.randn(2, 8, 16, generator=g)}
    out["slab_ocean"] = {"input": oc_in, "gen": sl_gen, "target": sl_target, "expected": {}, "timestep_seconds": timestep.total_seconds()}
    for interp in (False, True):
        ocean = ocean_mod.OceanConfig(surface_temperature_name="sst", ocean_fraction_name="frac", interpolate=interp,
                                      slab=ocean_mod.SlabOceanConfig(mixed_layer_depth_name="mld", q_flux_name="qflux")
                                      ).build(["sst", "frac", "q"], ["sst", "q"] + flux_names, timestep)
        out["slab_ocean"]["expected"][interp] = {k: v.clone() for k, v in ocean(oc_in, sl_gen, sl_target).items()}
        out["slab_ocean"].setdefault("forcing_names", sorted(ocean.forcing_names))
    path = os.path.join(HERE, "gen_corrector.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
