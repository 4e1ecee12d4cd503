"""Golden vectors for the reference's RNG contract of stochastic modules (fme/core/rand.py:39-104, fme/core/random_state.py,
fme/ace/stepper/single_module.py:1063-1068), emitted by the REAL reference stepper imported under stubs
(oracle/ref_loader.load_stepper_ref) - build container only.  Writes tests/golden/gen_rng.pt:

  per case ("isotropic", "gaussian_groups2"): the ``Stepper.get_state()`` of a small NoiseConditionedSFNO stepper (its
  conditioning weights randomised - the reference initialises them to zero, which would hide the noise), the (initial
  condition, forcing), the seed, and the reference's own ``predict_generator`` output of every step of a 3-step rollout
  with ``stepper_state=StepperState(random_state=RandomState.from_seed(seed))`` on CPU; plus
    "steps_unseeded_differ"   max |difference| of the same rollout under another seed (the noise matters)
    "generator_state_after"   the advanced generator state after the 3 steps (a restart continues the sequence)
    "next_draw"               torch.randn(4) from that state

Data only: loadable with ``torch.load(weights_only=True)``."""
import datetime
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

B, H, W, T = 2, 12, 24, 3
FORCING = ["f0", "f1"]
PROGNOSTIC = ["p0", "p1", "p2"]
DIAGNOSTIC = ["d0"]
CASES = {
    "isotropic": dict(embed_dim=16, noise_embed_dim=8, noise_type="isotropic", num_layers=2, affine_norms=True),
    "gaussian_groups2": dict(embed_dim=16, noise_embed_dim=4, noise_type="gaussian", num_layers=2, filter_num_groups=2),
}
SEED = 20260928


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
    StepperState = importlib.import_module("fme.core.stepper_state").StepperState
    RandomState = importlib.import_module("fme.core.random_state").RandomState
    lat = torch.linspace(-82.5, 82.5, H)
    lon = torch.arange(float(W)) * (360.0 / W)
    info = ref.DatasetInfo(horizontal_coordinates=ref.LatLonCoordinates(lat=lat, lon=lon),
                           vertical_coordinate=ref.HybridSigmaPressureCoordinate(ak=torch.tensor([100.0, 8000.0, 0.0]),
                                                                                 bk=torch.tensor([0.0, 0.3, 1.0])),
                           timestep=datetime.timedelta(hours=6))
    in_names, out_names = FORCING + PROGNOSTIC, PROGNOSTIC + DIAGNOSTIC
    names = sorted(set(in_names + out_names))
    out = {}
    for case, kw in CASES.items():
        cfg = {"step": {"type": "single_module", "config": dict(
            builder={"type": "NoiseConditionedSFNO", "config": kw}, in_names=in_names, out_names=out_names,
            normalization={"network": {"means": {n: 0.1 for n in names}, "stds": {n: 1.3 for n in names}}})}}
        torch.manual_seed(0)
        stepper = ref.StepperConfig.from_stepper_state({"config": cfg}).get_stepper(dataset_info=info)
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for k, p in stepper.modules.named_parameters():
                if "W_scale_2d" in k or "W_bias_2d" in k:
                    p.copy_(0.3 * torch.randn(p.shape, generator=g))
                if ".norm.weight" in k:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                if ".norm.bias" in k or k.endswith("filter.filter.bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        ic = {n: torch.randn(B, 1, H, W, generator=g) for n in PROGNOSTIC}
        forcing = {n: torch.randn(B, T + 1, H, W, generator=g) for n in FORCING}

        def rollout(seed):
            torch.manual_seed(999)     # the global RNG must not matter under a seeded random state
            state = StepperState(random_state=RandomState.from_seed(seed))
            steps = []
            with torch.no_grad():
                for res in stepper.predict_generator(ic, forcing, T, ref.NullOptimization(), labels=None, stepper_state=state):
                    steps.append({k: v.clone() for k, v in res.output.items()})
                    assert res.stepper_state is not None and res.stepper_state.random_state is state.random_state
            return steps, state

        steps, state = rollout(SEED)
        again, _ = rollout(SEED)
        for a, b in zip(steps, again):
            for k in a:
                assert torch.equal(a[k], b[k]), (case, k)         # reproducible
        other, _ = rollout(SEED + 1)
        differ = max(float((a[k] - b[k]).abs().max()) for a, b in zip(steps, other) for k in a)
        assert differ > 1e-3, differ
        gstate = state.random_state.generator.get_state().clone()
        g2 = torch.Generator()
        g2.set_state(gstate)
        out[case] = {"kwargs": kw, "state": plain(stepper.get_state()), "ic": ic, "forcing": forcing, "seed": SEED,
                     "in_names": in_names, "out_names": out_names, "steps": steps, "steps_unseeded_differ": differ,
                     "generator_state_after": gstate, "next_draw": torch.randn(4, generator=g2)}
        print(case, "outputs", sorted(steps[0]), "| max |y|", max(float(v.abs().max()) for v in steps[-1].values()),
              "| other seed differs by", differ)
    path = os.path.join(HERE, "gen_rng.pt")
    torch.save(out, path)
    torch.load(path, weights_only=True)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
