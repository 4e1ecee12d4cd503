"""Golden vectors for the derived output variables, emitted by the REAL reference (fme/core/derived_variables.py
``compute_derived_quantities``) imported under stubs (oracle/ref_loader.load_stepper_ref) - build container only.
Writes tests/golden/gen_derived.pt: a synthetic 5-level time series (2 samples, 4 time levels, 8 x 16 grid, 2 layers), the
forcing it is paired with, and the reference's outputs for (a) every input present, (b) winds and the advective tendency
missing (those variables are skipped), plus the registry order."""
import datetime
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402
from make_golden_checkpoint import SCALES, field  # noqa: E402


def main():
    ref = ref_loader.load_stepper_ref()
    dv = importlib.import_module("fme.core.derived_variables")
    vc = ref.HybridSigmaPressureCoordinate(ak=torch.tensor([100.0, 8000.0, 0.0]), bk=torch.tensor([0.0, 0.3, 1.0]))
    timestep = datetime.timedelta(hours=6)
    g = torch.Generator().manual_seed(11)
    SCALES.update({"UGRD10m": (2.0, 5.0), "VGRD10m": (-1.0, 4.0)})
    names = [n for n in SCALES if n not in ("DSWRFtoa", "HGTsfc", "ocean_fraction")]
    data = {n: field(g, n, 2, 4) for n in names}
    forcing = {n: field(g, n, 2, 4) for n in ("DSWRFtoa", "HGTsfc")}
    out = {"data": data, "forcing": forcing, "ak": vc.ak, "bk": vc.bk, "timestep_seconds": timestep.total_seconds(),
           "registry": list(dv._DERIVED_VARIABLE_REGISTRY)}
    full = dv.compute_derived_quantities(dict(data), vc, timestep, forcing_data=dict(forcing))
    out["full"] = {k: v.clone() for k, v in full.items() if k not in data and k not in forcing}
    part_in = {k: v for k, v in data.items() if k not in ("UGRD10m", "tendency_of_total_water_path_due_to_advection")}
    part = dv.compute_derived_quantities(dict(part_in), vc, timestep, forcing_data=dict(forcing))
    out["partial"] = {k: v.clone() for k, v in part.items() if k not in data and k not in forcing}
    print("full:", sorted(out["full"]))
    print("partial misses:", sorted(set(out["full"]) - set(out["partial"])))
    path = os.path.join(HERE, "gen_derived.pt")
    torch.save(out, path)
    torch.load(path, weights_only=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
