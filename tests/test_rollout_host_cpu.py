"""Host logic of the rollout engine on an emulation of the C ABI (tests/_fake_sfno.py: the forward is the CPU oracle network built
from the weights the engine uploads; pack / unpack walk the engine's pointer tables) - static buffers, per-step pointer tables and
strides, forcing indices (next-step forcing), state feedback, residual prediction, prescribed prognostics, post-step hooks as torch
ops, derived forcings, the window feeder - against the oracle's stepper loop (fme/ace/stepper/single_module.py:1124-1167 restated in
oracle/stepper.py) and against ``Stepper.predict`` with the same oracle network as its module."""
import datetime
import warnings

import pytest
import torch

import ace_amd
from ace_amd.registry import Module
from ace_amd.rollout import RolloutEngine
from ace_amd.step import NormalizationConfig
from _fake_sfno import fake_sfno

H, W = 8, 16


def _config(in_names, out_names, **kw):
    names = sorted(set(in_names) | set(out_names))
    norm = NormalizationConfig(means={k: 0.1 * (i + 1) for i, k in enumerate(names)}, stds={k: 1.0 + 0.1 * i for i, k in enumerate(names)})
    return ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet",
                                       config={"embed_dim": 8, "num_layers": 2, "operator_type": "dhconv"}),
        in_names=in_names, out_names=out_names, normalization=norm, **kw), names, norm


def _oracle_net(stepper, n_in, n_out):
    from oracle.sfno import SFNOConfig, SFNOOracle
    cfg = SFNOConfig(in_chans=n_in, out_chans=n_out, img_shape=(H, W), embed_dim=8, num_layers=2, operator_type="dhconv")
    return SFNOOracle(cfg, stepper.modules[0].state_dict(), dtype=torch.float32)


class _OracleModule(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self._net = net

    def forward(self, x):
        return self._net.forward(x)


def _reference_stepper(config, info, stepper, **kw):
    """a Stepper with the same configuration whose module is the oracle network (CPU), weights of `stepper`"""
    ref = ace_amd.Stepper.from_config(config, info, device="cpu", **kw)
    ref._step_obj.module = Module(_OracleModule(_oracle_net(stepper, len(config.in_names), len(config.out_names))), None)
    return ref


@pytest.mark.parametrize("graph", [None, "step"])
def test_engine_pointer_tables_and_forcing_indices_vs_the_oracle_loop(graph):
    from oracle import stepper as ostep
    in_names, out_names = ["f0", "p0", "p1", "f1"], ["p1", "d0", "p0"]
    config, names, norm = _config(in_names, out_names, next_step_forcing_names=["f1"])
    torch.manual_seed(0)
    stepper = ace_amd.Stepper.from_config(config, ace_amd.DatasetInfo((H, W)), device="cpu")
    B, T = 2, 4
    g = torch.Generator().manual_seed(1)
    ic = {k: torch.randn(B, 1, H, W, generator=g) for k in ["p0", "p1"]}
    forcing = {k: torch.randn(B, T + 1, H, W, generator=g) for k in ["f0", "f1"]}
    with fake_sfno() as fake:
        eng = RolloutEngine(stepper, batch=B, n_forward_steps=T, graph=graph)
        out, state = eng.predict(ic, forcing)
        out = {k: v.clone() for k, v in out.items()}
        again, _ = eng.predict(ic, forcing)
        assert sum(n.forwards for n in fake._nets.values()) == 2 * T
    means = {k: torch.tensor(norm.means[k]) for k in names}
    stds = {k: torch.tensor(norm.stds[k]) for k in names}
    ref = ostep.predict(_oracle_net(stepper, 4, 3), ic, forcing, T, in_names, out_names, means, stds, next_step_forcing_names=["f1"])
    for k in out_names:
        want = torch.stack([o[k] for o in ref], 1)
        assert float((out[k] - want).abs().max()) <= 2e-6 * float(want.abs().max()), k
        assert torch.equal(again[k], out[k])
    for k in ["p0", "p1"]:
        assert torch.equal(state[k], out[k][:, -1:])


def test_configs0_s80_plumbing_at_180x360_four_steps():
    """BASELINE.json configs[0] / SURVEY 8(d) "S80" on the CPU: 80 distinct variables (8 forcing-only + 36 prognostic in = 44; 36
    prognostic + 36 diagnostic out = 72) on the full 1-degree grid, mu = 0.1 / sigma = 1.1 for every name (test_single_module.py:
    2331-2333), 4 six-hourly steps - names -> channels, normalise, network, denormalise, state feedback, forcing index - through the
    engine (emulated C ABI) and through Stepper.predict, against the oracle's stepper loop.  The network is a small SFNO (embed 16 x
    2 layers) so that the 180 x 360 plumbing runs in seconds; the full-width arithmetic is the GPU suite's business."""
    from oracle import stepper as ostep
    from oracle.sfno import SFNOConfig, SFNOOracle
    HH, WW, T, B = 180, 360, 4, 1
    forcing_names = [f"forcing_{i}" for i in range(8)]
    prog = [f"prog_{i}" for i in range(36)]
    diag = [f"diag_{i}" for i in range(36)]
    in_names, out_names = forcing_names + prog, prog + diag
    names = forcing_names + prog + diag
    assert len(set(names)) == 80 and len(in_names) == 44 and len(out_names) == 72
    norm = NormalizationConfig(means={k: 0.1 for k in names}, stds={k: 1.1 for k in names})
    config = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 16, "num_layers": 2, "operator_type": "dhconv"}),
        in_names=in_names, out_names=out_names, normalization=norm)
    torch.manual_seed(0)
    stepper = ace_amd.Stepper.from_config(config, ace_amd.DatasetInfo((HH, WW)), device="cpu")
    g = torch.Generator().manual_seed(1)
    ic = {k: torch.randn(B, 1, HH, WW, generator=g) for k in prog}
    forcing = {k: torch.randn(B, T + 1, HH, WW, generator=g) for k in forcing_names}
    net = SFNOOracle(SFNOConfig(in_chans=44, out_chans=72, img_shape=(HH, WW), embed_dim=16, num_layers=2, operator_type="dhconv"),
                     stepper.modules[0].state_dict(), dtype=torch.float32)
    means = {k: torch.tensor(0.1) for k in names}
    stds = {k: torch.tensor(1.1) for k in names}
    ref = ostep.predict(net, ic, forcing, T, in_names, out_names, means, stds)
    with fake_sfno():
        out, state = RolloutEngine(stepper, batch=B, n_forward_steps=T, graph="step").predict(ic, forcing)
    for k in out_names:
        want = torch.stack([o[k] for o in ref], 1)
        assert out[k].shape == (B, T, HH, WW)
        assert float((out[k] - want).abs().max()) <= 2e-6 * float(want.abs().max()), k
    for k in prog:
        assert torch.equal(state[k], out[k][:, -1:])
    # the same through Stepper.predict with the oracle network as its module
    cpu = _reference_stepper(config, ace_amd.DatasetInfo((HH, WW)), stepper)
    cpu._step_obj.module = Module(_OracleModule(net), None)
    got, _ = cpu.predict(ic, forcing)
    for k in out_names:
        assert float((got[k] - out[k]).abs().max()) <= 2e-6 * float(out[k].abs().max()), k


def test_engine_residual_prescribed_and_continue_from_last():
    in_names, out_names = ["f0", "p0", "p1"], ["p0", "p1", "d0"]
    config, names, norm = _config(in_names, out_names, residual_prediction=True, prescribed_prognostic_names=["p1"])
    info = ace_amd.DatasetInfo((H, W))
    torch.manual_seed(2)
    stepper = ace_amd.Stepper.from_config(config, info, device="cpu")
    ref = _reference_stepper(config, info, stepper)
    B, T = 1, 3
    g = torch.Generator().manual_seed(3)
    ic = {k: torch.randn(B, 1, H, W, generator=g) for k in ["p0", "p1"]}
    forcing = {k: torch.randn(B, 2 * T + 1, H, W, generator=g) for k in ["f0", "p1"]}       # p1 is prescribed from the data
    want, want_state = ref.predict(ic, forcing, n_forward_steps=2 * T)
    with fake_sfno():
        eng = RolloutEngine(stepper, batch=B, n_forward_steps=T, graph="step")
        first, s1 = eng.predict(ic, {k: v[:, : T + 1] for k, v in forcing.items()})
        first = {k: v.clone() for k, v in first.items()}
        second, s2 = eng.predict(s1, {k: v[:, T:] for k, v in forcing.items()})            # next window from the carried state
    for k in out_names:
        got = torch.cat([first[k], second[k]], dim=1)
        assert float((got - want[k]).abs().max()) <= 5e-6 * float(want[k].abs().max()), k
    assert torch.equal(second["p1"][:, -1], forcing["p1"][:, -1])                            # prescribed: overwritten from step s + 1


def test_engine_with_slab_ocean_hooks_as_torch_ops():
    """corrector (force-positive) + slab ocean through the engine's torch-op hook path vs Stepper.predict"""
    in_names = ["sst", "ocean_fraction", "mld", "qflux", "p0"]
    out_names = ["sst", "p0", "DLWRFsfc", "DSWRFsfc", "ULWRFsfc", "USWRFsfc", "LHTFLsfc", "SHTFLsfc"]
    ocean = {"surface_temperature_name": "sst", "ocean_fraction_name": "ocean_fraction",
             "slab": {"mixed_layer_depth_name": "mld", "q_flux_name": "qflux"}}
    from ace_amd.corrector import AtmosphereCorrectorConfig
    config, names, norm = _config(in_names, out_names, ocean=ocean, corrector=AtmosphereCorrectorConfig(force_positive_names=["p0"]))
    info = ace_amd.DatasetInfo((H, W), timestep=datetime.timedelta(hours=6))
    torch.manual_seed(4)
    stepper = ace_amd.Stepper.from_config(config, info, device="cpu")
    ref = _reference_stepper(config, info, stepper)
    B, T = 2, 3
    g = torch.Generator().manual_seed(5)
    ic = {"sst": 290.0 + torch.randn(B, 1, H, W, generator=g), "p0": torch.randn(B, 1, H, W, generator=g)}
    forcing = {"ocean_fraction": torch.rand(B, T + 1, H, W, generator=g), "mld": 20.0 + 30.0 * torch.rand(B, T + 1, H, W, generator=g),
               "qflux": 10.0 * torch.randn(B, T + 1, H, W, generator=g)}
    want, _ = ref.predict(ic, forcing)
    with fake_sfno():
        out, _ = RolloutEngine(stepper, batch=B, n_forward_steps=T, graph=None).predict(ic, forcing)
    for k in out_names:
        assert float((out[k] - want[k]).abs().max()) <= 5e-6 * max(float(want[k].abs().max()), 1.0), k
    assert float(out["p0"].min()) >= 0.0


def test_derived_insolation_through_engine_windows_and_run_inference(tmp_path):
    """A stepper whose configuration derives the insolation: RolloutEngine.predict(time=), EnginePredict on ForcingWindows that carry
    their time slices, run_inference over uneven windows - all equal to handing the precomputed field in."""
    from ace_amd.inference import EnginePredict, ForcingWindows, InferenceData, TensorFileWriter, run_inference
    from ace_amd.timeaxis import TimeAxis
    in_names, out_names = ["sun", "p0", "f1"], ["p0", "d0"]
    names = sorted(set(in_names) | set(out_names))
    norm = NormalizationConfig(means={k: (300.0 if k == "sun" else 0.2) for k in names}, stds={k: (400.0 if k == "sun" else 1.5) for k in names})
    config = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 8, "num_layers": 2, "operator_type": "dhconv"}),
        in_names=in_names, out_names=out_names, normalization=norm, next_step_forcing_names=["sun"])
    info = ace_amd.DatasetInfo((H, W), lat=torch.linspace(-78.75, 78.75, H), lon=torch.arange(W) * 22.5)
    derived = {"insolation": {"insolation_name": "sun", "solar_constant": {"value": 1360.0}}}
    torch.manual_seed(6)
    stepper = ace_amd.Stepper.from_config(config, info, device="cpu", derived_forcings=derived)
    plain = ace_amd.Stepper.from_config(config, info, device="cpu")
    plain._step_obj.module.torch_module.load_state_dict(stepper.modules[0].state_dict())
    B, total, T = 2, 5, 2
    g = torch.Generator().manual_seed(7)
    ic = {"p0": torch.randn(B, 1, H, W, generator=g)}
    record = {"f1": torch.randn(B, total + 1, H, W, generator=g)}
    time = TimeAxis.regular((2022, 3, 20, 12), datetime.timedelta(hours=6), total + 1, B)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sun = stepper.forcing_deriver(record, time)["sun"]
        assert float(sun.max()) > 800 and sun.shape == (B, total + 1, H, W)
        with fake_sfno():
            eng = RolloutEngine(stepper, batch=B, n_forward_steps=total, graph="step")
            out, _ = eng.predict(ic, record, time=time)
            out = {k: v.clone() for k, v in out.items()}
            with pytest.raises(ValueError, match="time axis"):
                eng.predict(ic, record)
            explicit, _ = RolloutEngine(plain, batch=B, n_forward_steps=total, graph="step").predict(ic, {**record, "sun": sun})
            for k in out_names:
                assert torch.equal(out[k], explicit[k]), k
            # the windowed driver: windows of 2, 2 and 1 steps, each with its own slice of the time axis
            windows = ForcingWindows(record, total, T, device="cpu", time=time)
            writer = TensorFileWriter(str(tmp_path))
            final = run_inference(EnginePredict(stepper, batch=B, graph="step"), InferenceData(ic, windows), writer=writer)
    series = torch.load(tmp_path / "autoregressive_predictions.pt")
    for k in out_names:
        assert float((series[k] - out[k]).abs().max()) <= 2e-6 * float(out[k].abs().max()), k
    assert torch.equal(final["p0"].cpu(), series["p0"][:, -1:])
    assert (float((out["p0"][:, 0] - ic["p0"][:, 0]).abs().max())) > 0


def test_engine_fills_nans_as_the_normaliser_does():
    """fill_nans_on_normalize / fill_nans_on_denormalize (fme/core/normalizer.py:212-242): a masked (NaN) region of a forcing enters
    the network as 0 in normalised space - same rollout as Stepper.predict."""
    in_names, out_names = ["f0", "p0"], ["p0", "d0"]
    names = sorted(set(in_names) | set(out_names))
    norm = NormalizationConfig(means={k: 0.3 * (i + 1) for i, k in enumerate(names)}, stds={k: 1.0 + 0.2 * i for i, k in enumerate(names)},
                               fill_nans_on_normalize=True, fill_nans_on_denormalize=True)
    config = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 8, "num_layers": 2, "operator_type": "dhconv"}),
        in_names=in_names, out_names=out_names, normalization=norm)
    info = ace_amd.DatasetInfo((H, W))
    torch.manual_seed(8)
    stepper = ace_amd.Stepper.from_config(config, info, device="cpu")
    ref = _reference_stepper(config, info, stepper)
    B, T = 2, 3
    g = torch.Generator().manual_seed(9)
    ic = {"p0": torch.randn(B, 1, H, W, generator=g)}
    forcing = {"f0": torch.randn(B, T + 1, H, W, generator=g)}
    forcing["f0"][:, :, :2, :5] = float("nan")                      # a "land" patch the data does not cover
    want, _ = ref.predict(ic, forcing)
    with fake_sfno():
        out, _ = RolloutEngine(stepper, batch=B, n_forward_steps=T, graph="step").predict(ic, forcing)
    for k in out_names:
        assert bool(torch.isfinite(out[k]).all()), k
        assert float((out[k] - want[k]).abs().max()) <= 5e-6 * float(want[k].abs().max()), k


@pytest.mark.parametrize("case,graph", [("ace2_like", None), ("ace2_like", "step"), ("residual_prescribed", "step"),
                                        ("ace2_like_override", None)])
def test_engine_host_logic_vs_the_real_reference_stepper_rollout(case, graph):
    """The CPU counterpart of the GPU suite's drop-in check: a state written by the REAL reference stepper
    (tests/golden/gen_checkpoint.pt) loaded by ace_amd.load_stepper and rolled by the RolloutEngine on the emulated C ABI (oracle
    network, torch-op hooks: ACE2-style corrector with its dry-air state, prescribed-SST ocean, next-step forcing; residual
    prediction with a prescribed prognostic; the inference-time override) against the reference's own per-step outputs - same
    tolerance rule as on the GPU (1e-5 of the field maximum per step, or 3 x the reference's own fp32 distance from exact arithmetic)."""
    from _util import checkpoint_case, checkpoint_override, conditioning_floor, load_golden
    g = checkpoint_case(load_golden("gen_checkpoint.pt"), case)
    loaded = ace_amd.load_stepper(g["state"], override_config=checkpoint_override(g), device="cpu")
    T = len(g["steps"])
    with fake_sfno():
        out, state = RolloutEngine(loaded.stepper, batch=2, n_forward_steps=T, graph=graph).predict(g["ic"], g["forcing"])
    assert set(out) == set(g["steps"][0])
    floor = conditioning_floor(g)
    for s, want_all in enumerate(g["steps"]):
        for k, want in want_all.items():
            err = float((out[k][:, s] - want).abs().max()) / float(want.abs().max())
            assert err <= max(1e-5 * (s + 1), 3.0 * floor[s][k]), (k, s, err)
    if case == "ace2_like":
        assert state.stepper_state.corrector_state.global_dry_air_mass is not None


def test_windowed_engine_inference_vs_the_reference_continuous_rollout(tmp_path):
    """run_inference over windows of 2 + 1 steps through EnginePredict == the reference stepper's continuous 3-step rollout: the
    corrector's dry-air reference rides on the prognostic state from window to window (CPU counterpart of the GPU test)."""
    from ace_amd.inference import EnginePredict, ForcingWindows, InferenceData, TensorFileWriter, run_inference
    from _util import conditioning_floor, load_golden
    g = load_golden("gen_checkpoint.pt")["ace2_like"]
    stepper = ace_amd.load_stepper(g["state"], device="cpu").stepper
    T = len(g["steps"])
    with fake_sfno():
        loader = ForcingWindows(g["forcing"], total_forward_steps=T, forward_steps_in_memory=2, device="cpu")
        state = run_inference(EnginePredict(stepper, batch=2, graph="step"), InferenceData(g["ic"], loader), writer=TensorFileWriter(str(tmp_path)))
        first = run_inference(EnginePredict(stepper, batch=2, graph="step"), InferenceData(g["ic"], ForcingWindows(g["forcing"], 1, 1, device="cpu")))
    series = torch.load(tmp_path / "autoregressive_predictions.pt", weights_only=True)
    floor = conditioning_floor(g)
    for s, want_all in enumerate(g["steps"]):
        for k, want in want_all.items():
            err = float((series[k][:, s] - want).abs().max()) / float(want.abs().max())
            assert err <= max(1e-5 * (s + 1), 3.0 * floor[s][k]), (k, s, err)
    cs = state.stepper_state.corrector_state
    assert cs is not None and torch.equal(cs.global_dry_air_mass, first.stepper_state.corrector_state.global_dry_air_mass)


def test_conditional_engine_with_labels_and_positional_context_vs_the_oracle():
    """NoiseConditionedSFNO behind the engine on the emulated C ABI: the host forms ONE conditioning field
    cat(noise, positional context + labels . label_pos_embed, label planes, ones) and uploads merged per-norm weights; the emulation's
    forward is the CPU oracle with exactly that single field.  Reference for the comparison: the oracle of the FULL model (noise,
    labels and positional context as the reference has them - bit for bit with the reference in fp32, tests/test_csfno_oracle.py)
    stepped by hand with the same noise draws."""
    from ace_amd.labels import BatchLabels
    from oracle.csfno import CSFNOConfig, CSFNOOracle
    in_names, out_names = ["f0", "p0"], ["p0", "d0"]
    names = sorted(set(in_names) | set(out_names))
    kw = {"embed_dim": 8, "noise_embed_dim": 3, "num_layers": 2, "pos_embed": False, "context_pos_embed_dim": 2, "label_embed_dim": 4,
          "affine_norms": True, "normalize_big_skip": True}
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="NoiseConditionedSFNO", conditional=True, config=kw), in_names=in_names, out_names=out_names,
        normalization=NormalizationConfig(means={k: 0.1 for k in names}, stds={k: 1.2 for k in names}))
    torch.manual_seed(0)
    stepper = ace_amd.Stepper.from_config(cfg, ace_amd.DatasetInfo((H, W), all_labels={"era5", "shield", "cm4"}), device="cpu")
    net = stepper.modules[0]
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if ".W_scale_" in k or ".W_bias_" in k or k in ("label_pos_embed",):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    B, T = 2, 3
    ic = {"p0": torch.randn(B, 1, H, W, generator=g)}
    forcing = {"f0": torch.randn(B, T + 1, H, W, generator=g)}
    enc = stepper._step_obj.module._label_encoding
    assert enc.names == ["cm4", "era5", "shield"]
    labels = enc.encode([{"era5"}, {"shield", "cm4"}], "cpu")
    with fake_sfno():
        eng = RolloutEngine(stepper, batch=B, n_forward_steps=T, graph=None)
        eng.set_labels(BatchLabels(labels.tensor[:, [2, 0, 1]], ["shield", "cm4", "era5"]))     # another column order: conformed
        torch.manual_seed(5)
        out, _ = eng.predict(ic, forcing)
        out = {k: v.clone() for k, v in out.items()}
        eng.set_labels(None)
        with pytest.raises(ValueError, match="labels must be provided"):
            eng.predict(ic, forcing)
    ocfg = CSFNOConfig(in_chans=2, out_chans=2, img_shape=(H, W), noise_type="gaussian", **kw)
    oracle = CSFNOOracle(ocfg, net.state_dict(), dtype=torch.float32)
    torch.manual_seed(5)
    state = ic["p0"][:, 0]
    for s in range(T):
        x = torch.stack([(forcing["f0"][:, s] - 0.1) / 1.2, (state - 0.1) / 1.2], dim=1)
        noise = torch.randn(B, 3, H, W)                                   # the engine's draw of this step (gaussian)
        y = oracle.forward(x, noise=noise, labels=labels.tensor) * 1.2 + 0.1
        for i, k in enumerate(out_names):
            # (fp32 both ways; the merged single-field sum and the reference's three separate sums round differently, free-running)
            assert float((out[k][:, s] - y[:, i]).abs().max()) <= 1e-5 * (s + 1) * float(y[:, i].abs().max()), (k, s)
        state = y[:, 0]
