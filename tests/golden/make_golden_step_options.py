"""tests/golden/gen_step_options.pt from the REAL reference stepper (oracle/ref_loader.load_stepper_ref - build container only): a
small SphericalFourierNeuralOperatorNet stepper wrapped in ``multi_call`` WITH an active multi-call configuration (two CO2
multipliers, one 2-D and one level-suffixed output) and with a secondary decoder (registry "MLP", two diagnostics) - its
``get_state()``, the inputs, and the reference's own ``predict_generator`` output of every step of a 3-step rollout on CPU;
plus the "MLP" network on its own (seeded build, state dict, input, output).  Data only."""
import datetime
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

B, H, W, T = 2, 8, 16, 3
FORCING = ["co2", "f0"]
PROGNOSTIC = ["p0", "T_1"]
DIAGNOSTIC = ["ULWRFtoa", "h_3"]
SECONDARY = ["s0", "s1"]


def plain(o):
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
    for pkg in ["fme.core.models.mlp"]:
        if pkg not in sys.modules:
            ref_loader._ns(pkg, os.path.join(ref_loader.REF, *pkg.split(".")))
    mlp = importlib.import_module("fme.core.models.mlp.mlp")          # registers "MLP"
    info = ref.DatasetInfo(horizontal_coordinates=ref.LatLonCoordinates(lat=torch.linspace(-78.75, 78.75, H),
                                                                       lon=torch.arange(float(W)) * (360.0 / W)),
                           vertical_coordinate=ref.HybridSigmaPressureCoordinate(ak=torch.tensor([100.0, 8000.0, 0.0]),
                                                                                 bk=torch.tensor([0.0, 0.3, 1.0])),
                           timestep=datetime.timedelta(hours=6))
    names = FORCING + PROGNOSTIC + DIAGNOSTIC + SECONDARY
    inner = {"type": "single_module", "config": dict(
        builder={"type": "SphericalFourierNeuralOperatorNet",
                 "config": {"embed_dim": 12, "num_layers": 2, "operator_type": "dhconv", "data_grid": "legendre-gauss"}},
        in_names=FORCING + PROGNOSTIC, out_names=PROGNOSTIC + DIAGNOSTIC,
        normalization={"network": {"means": {n: 0.1 * (i + 1) for i, n in enumerate(names)},
                                   "stds": {n: 1.0 + 0.1 * i for i, n in enumerate(names)}}},
        secondary_decoder={"secondary_diagnostic_names": SECONDARY, "network": {"type": "MLP", "config": {"hidden_dim": 10, "depth": 3}}})}
    multi = {"forcing_name": "co2", "forcing_multipliers": {"_doubled_co2": 2.0, "_halved_co2": 0.5}, "output_names": ["ULWRFtoa", "h_3"]}
    cfg = {"step": {"type": "multi_call", "config": {"wrapped_step": inner, "config": multi, "include_multi_call_in_loss": False}}}
    torch.manual_seed(0)
    stepper = ref.StepperConfig.from_stepper_state({"config": cfg}).get_stepper(dataset_info=info)
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for p in stepper.modules.parameters():
            if p.ndim <= 1 or p.abs().max() == 0:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    g = torch.Generator().manual_seed(2)
    ic = {n: torch.randn(B, 1, H, W, generator=g) for n in PROGNOSTIC}
    forcing = {n: torch.randn(B, T + 1, H, W, generator=g) + (2.0 if n == "co2" else 0.0) for n in FORCING}
    steps = []
    with torch.no_grad():
        for res in stepper.predict_generator(ic, forcing, T, ref.NullOptimization(), labels=None):
            steps.append({k: v.clone() for k, v in res.output.items()})
    print("outputs", sorted(steps[0]))
    out = {"stepper": {"state": plain(stepper.get_state()), "ic": ic, "forcing": forcing, "steps": steps}}

    torch.manual_seed(7)
    net = mlp.MLPConfig(hidden_dim=24, depth=3).build(6, 5, info)
    x = torch.randn(3, 6, 9, 20, generator=g)
    with torch.no_grad():
        y = net(x)
    torch.manual_seed(8)
    one = mlp.MLPConfig(hidden_dim=4, depth=1).build(4, 2, info)
    x1 = torch.randn(1, 4, 5, 6, generator=g)                        # a row length that is not a multiple of 4
    with torch.no_grad():
        y1 = one(x1)
    out["mlp"] = {"deep": {"config": {"hidden_dim": 24, "depth": 3}, "seed": 7, "n_in": 6, "n_out": 5,
                           "state_dict": {k: v.clone() for k, v in net.state_dict().items()}, "x": x, "y": y},
                  "single": {"config": {"hidden_dim": 4, "depth": 1}, "seed": 8, "n_in": 4, "n_out": 2,
                             "state_dict": {k: v.clone() for k, v in one.state_dict().items()}, "x": x1, "y": y1}}
    path = os.path.join(HERE, "gen_step_options.pt")
    torch.save(out, path)
    torch.load(path, weights_only=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
