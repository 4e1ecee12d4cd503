"""Step options pinned on the REAL reference stepper (tests/golden/make_golden_step_options.py): multi-call diagnostics
(fme/core/step/multi_call.py) and the secondary decoder (fme/core/step/secondary_decoder.py) built from the "MLP" registry network
(fme/core/models/mlp/mlp.py) - a reference stepper state with both, loaded by ace_amd.load_stepper, must reproduce the reference's
3-step rollout, names included.

CPU (``-m "not gpu"``): the network is the CPU oracle SFNO with the checkpoint's weights, the secondary decoder runs ace_amd/mlp.py on
the emulated k = 1 operator (tests/_fake_hpx.py) - configuration parsing, state loading, name handling and the step / multi-call /
decoder composition are what is checked.  GPU: the same through the real kernels."""
import os

import pytest
import torch

import ace_amd
from ace_amd.checkpoint import StepperOverrideConfig, load_stepper
from ace_amd.registry import Module
from _fake_hpx import fake_hpx
from _util import load_golden, rel_max


class _OracleModule(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self._net = net

    def forward(self, x):
        return self._net.forward(x)


def _with_oracle_network(stepper, n_in, n_out, img_shape, embed_dim):
    from oracle.sfno import SFNOConfig, SFNOOracle
    cfg = SFNOConfig(in_chans=n_in, out_chans=n_out, img_shape=img_shape, embed_dim=embed_dim, num_layers=2, operator_type="dhconv")
    stepper._step_obj.module = Module(_OracleModule(SFNOOracle(cfg, stepper.modules[0].state_dict(), dtype=torch.float32)), None)
    return stepper


def _rollout(stepper, g, **kw):
    out, state = stepper.predict(g["ic"], g["forcing"], **kw)
    return out, state


def test_mlp_on_the_emulated_operator_vs_reference():
    for name, g in load_golden("gen_step_options.pt")["mlp"].items():
        torch.manual_seed(g["seed"])
        net = ace_amd.ModuleSelector(type="MLP", config=g["config"]).build(g["n_in"], g["n_out"], ace_amd.DatasetInfo((4, 8))).torch_module
        assert list(net.state_dict()) == list(g["state_dict"])
        for k, v in net.state_dict().items():            # same construction order -> the reference's seeded initialisation
            assert torch.equal(v, g["state_dict"][k]), (name, k)
        with pytest.raises(RuntimeError, match="MI355X"):
            net(g["x"])
        with fake_hpx(), torch.no_grad():
            y = net(g["x"])
        assert y.shape == g["y"].shape and rel_max(y, g["y"]) <= 2e-6, (name, rel_max(y, g["y"]))
    with pytest.raises(ValueError, match="depth"):
        ace_amd.ModuleSelector(type="MLP", config={"depth": 0}).build(2, 2, ace_amd.DatasetInfo((4, 8)))


def test_stepper_with_multi_call_and_secondary_decoder_vs_reference_rollout():
    g = load_golden("gen_step_options.pt")["stepper"]
    loaded = load_stepper(g["state"], device="cpu")
    st = loaded.stepper
    assert loaded.ignored == [] or all("secondary" not in i and "multi_call" not in i for i in loaded.ignored)
    assert st.multi_call is not None and st._step_obj.secondary_decoder is not None
    assert set(st.out_names) == set(g["steps"][0])
    assert len(st.modules) == 2                                        # the network and the decoder's MLP
    _with_oracle_network(st, 4, 4, (8, 16), 12)
    with fake_hpx(), torch.no_grad():
        out, state = _rollout(st, g)
    assert set(out) == set(g["steps"][0])
    for k in out:
        want = torch.stack([s[k] for s in g["steps"]], dim=1)
        assert rel_max(out[k], want) <= 1e-5, (k, rel_max(out[k], want))
    assert set(state) == {"p0", "T_1"}
    # the override switches the multi-call diagnostics off, the decoder's stay
    off = load_stepper(g["state"], StepperOverrideConfig(multi_call=None), device="cpu").stepper
    _with_oracle_network(off, 4, 4, (8, 16), 12)
    with fake_hpx(), torch.no_grad():
        out2, _ = _rollout(off, g)
    assert set(out2) == {k for k in g["steps"][0] if "co2" not in k}
    for k in out2:
        assert torch.equal(out2[k], out[k]), k


def test_rollout_engine_with_secondary_decoder_host_logic():
    """the engine calls the decoder's module on its static network-output buffer and unpacks the result with a second pointer
    table - against Stepper.predict, on the emulations of both halves of the C ABI"""
    from ace_amd.rollout import RolloutEngine
    from _fake_sfno import fake_sfno
    g = load_golden("gen_step_options.pt")["stepper"]
    eng_st = load_stepper(g["state"], StepperOverrideConfig(multi_call=None), device="cpu").stepper
    ref_st = _with_oracle_network(load_stepper(g["state"], StepperOverrideConfig(multi_call=None), device="cpu").stepper, 4, 4, (8, 16), 12)
    with fake_hpx(), torch.no_grad():
        want, _ = _rollout(ref_st, g)
        with fake_sfno():
            eng = RolloutEngine(eng_st, batch=2, n_forward_steps=3, graph="step")
            out, state = eng.predict(g["ic"], g["forcing"])
            with pytest.raises(NotImplementedError, match="window"):
                RolloutEngine(eng_st, batch=2, n_forward_steps=3, graph="window")
    assert set(out) == set(want) and {"s0", "s1"} <= set(out)
    for k in out:
        assert rel_max(out[k], want[k]) <= 5e-6, (k, rel_max(out[k], want[k]))


@pytest.mark.parametrize("graph", [None, "step"])
def test_rollout_engine_with_multi_call_vs_reference_rollout(graph):
    """the engine's multi-call re-evaluations (scaled-forcing pointer tables, scratch planes, suffixed outputs) together with the
    secondary decoder, against the REAL reference stepper's rollout, on the emulations of both halves of the C ABI"""
    from ace_amd.rollout import RolloutEngine
    from _fake_sfno import fake_sfno
    g = load_golden("gen_step_options.pt")["stepper"]
    st = load_stepper(g["state"], device="cpu").stepper
    with fake_hpx(), fake_sfno(), torch.no_grad():
        eng = RolloutEngine(st, batch=2, n_forward_steps=3, graph=graph)
        out, state = eng.predict(g["ic"], g["forcing"])
        with pytest.raises(NotImplementedError, match="window"):
            RolloutEngine(st, batch=2, n_forward_steps=3, graph="window")
    assert set(out) == set(g["steps"][0])
    for k in out:
        want = torch.stack([s[k] for s in g["steps"]], dim=1)
        assert rel_max(out[k], want) <= 1e-5, (k, rel_max(out[k], want))


def test_rollout_engine_multi_call_with_corrector_and_ocean_hooks():
    """multi-call on an ACE2-like stepper (reference-written state: dry-air / moisture / energy corrector, prescribed-SST ocean,
    next-step forcing) with the scaled forcing being one the corrector itself reads: engine == Stepper.predict, over two windows
    (the re-evaluations' corrector state is seeded like the plain path's and carried)."""
    from ace_amd.rollout import RolloutEngine
    from _fake_sfno import fake_sfno
    g = load_golden("gen_checkpoint.pt")["ace2_like"]
    mc = {"forcing_name": "DSWRFtoa", "forcing_multipliers": {"_dim": 0.9, "_bright": 1.1}, "output_names": ["ULWRFtoa", "USWRFtoa"]}
    over = StepperOverrideConfig(multi_call=mc)
    eng_st = load_stepper(g["state"], over, device="cpu").stepper
    ref_st = load_stepper(g["state"], over, device="cpu").stepper
    cfg = ref_st._step_obj.config
    from oracle.sfno import SFNOConfig, SFNOOracle
    ocfg = SFNOConfig(in_chans=len(cfg.in_names), out_chans=len(cfg.out_names), img_shape=(8, 16), embed_dim=16, num_layers=2,
                      operator_type="dhconv")
    ref_st._step_obj.module = Module(_OracleModule(SFNOOracle(ocfg, ref_st.modules[0].state_dict(), dtype=torch.float32)), None)
    with torch.no_grad():
        want, _ = ref_st.predict(g["ic"], g["forcing"])
        with fake_sfno():
            eng = RolloutEngine(eng_st, batch=2, n_forward_steps=2, graph="step")
            first, s1 = eng.predict(g["ic"], {k: v[:, :3] for k, v in g["forcing"].items()})
            first = {k: v.clone() for k, v in first.items()}
            one = RolloutEngine(eng_st, batch=2, n_forward_steps=1, graph="step")
            second, _ = one.predict(s1, {k: v[:, 2:] for k, v in g["forcing"].items()})
    assert {"ULWRFtoa_dim", "ULWRFtoa_bright", "USWRFtoa_dim", "USWRFtoa_bright"} <= set(first)
    for k in want:
        got = torch.cat([first[k], second[k]], dim=1)
        scale = max(float(want[k].abs().max()), 1e-30)
        assert float((got - want[k]).abs().max()) <= 2e-5 * scale, (k, float((got - want[k]).abs().max()) / scale)
    assert float((want["ULWRFtoa_bright"] - want["ULWRFtoa"]).abs().max()) > 0           # the scaled forcing was seen


def test_secondary_decoder_configuration_errors():
    from ace_amd.step import NormalizationConfig
    norm = NormalizationConfig(means={k: 0.0 for k in "abcd"}, stds={k: 1.0 for k in "abcd"})
    base = dict(builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 4, "num_layers": 1}),
                in_names=["a", "b"], out_names=["b", "c"], normalization=norm)
    sd = lambda names: {"secondary_diagnostic_names": names, "network": {"type": "MLP", "config": {"hidden_dim": 4, "depth": 2}}}   # noqa: E731
    cfg = ace_amd.SingleModuleStepConfig(**base, secondary_decoder=sd(["d"]))
    assert cfg.output_names == ["b", "c", "d"] and set(cfg._normalize_names) == set("abcd")
    with pytest.raises(ValueError, match="input variable"):
        ace_amd.SingleModuleStepConfig(**base, secondary_decoder=sd(["a"]))
    with pytest.raises(ValueError, match="output variable"):
        ace_amd.SingleModuleStepConfig(**base, secondary_decoder=sd(["c"]))
    with pytest.raises(ValueError, match="surprise"):
        ace_amd.SingleModuleStepConfig(**base, secondary_decoder={**sd(["d"]), "surprise": 1})


@pytest.mark.gpu
def test_mlp_vs_reference_on_the_device():
    for name, g in load_golden("gen_step_options.pt")["mlp"].items():
        net = ace_amd.ModuleSelector(type="MLP", config=g["config"]).build(g["n_in"], g["n_out"], ace_amd.DatasetInfo((4, 8))).torch_module
        net.load_state_dict(g["state_dict"], strict=True)
        net = net.to("cuda")
        with torch.no_grad():
            y = net(g["x"].to("cuda"))
            assert torch.equal(y, net(g["x"].to("cuda")))
        assert rel_max(y, g["y"]) <= 2e-6, (name, rel_max(y, g["y"]))


@pytest.mark.gpu
def test_stepper_with_multi_call_and_secondary_decoder_on_the_device():
    """the same reference rollout through the real kernels (network, decoder MLP, the multi-call re-evaluations), per multiplier
    and as one batched step"""
    g = load_golden("gen_step_options.pt")["stepper"]
    st = load_stepper(g["state"], device="cuda").stepper
    ic = {k: v.to("cuda") for k, v in g["ic"].items()}
    forcing = {k: v.to("cuda") for k, v in g["forcing"].items()}
    with torch.no_grad():
        out, _ = st.predict(ic, forcing)
    for k in out:
        want = torch.stack([s[k] for s in g["steps"]], dim=1)
        assert rel_max(out[k], want) <= 1e-5, (k, rel_max(out[k], want))
    st.replace_multi_call(st.multi_call, batched=True)
    with torch.no_grad():
        batched, _ = st.predict(ic, forcing)
    for k in out:
        assert rel_max(batched[k], out[k]) <= 2e-6, k
    # the static-buffer engine produces both kinds of diagnostics too
    from ace_amd.rollout import RolloutEngine
    st.replace_multi_call(st.multi_call)
    eng_out, _ = RolloutEngine(st, batch=2, n_forward_steps=3, graph="step").predict(ic, forcing)
    assert set(eng_out) == set(out)
    for k in eng_out:
        assert rel_max(eng_out[k], out[k]) <= 5e-6, k


def test_rollout_engine_refuses_secondary_diagnostics_the_hooks_touch():
    """a secondary-decoder diagnostic that the corrector clamps (force_positive_names) would skip that hook in the engine's
    unpack-in-place layout: refused at construction with a pointer to Stepper.predict.  (The ocean's surface temperature cannot
    clash: it has to be an input and an output of the step, which a secondary diagnostic may not be.)"""
    from ace_amd.rollout import RolloutEngine
    from ace_amd.step import NormalizationConfig
    from _fake_sfno import fake_sfno
    names = ["f", "p", "q"]
    norm = NormalizationConfig(means={k: 0.0 for k in names}, stds={k: 1.0 for k in names})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 4, "num_layers": 1}),
        in_names=["f", "p"], out_names=["p"], normalization=norm,
        secondary_decoder={"secondary_diagnostic_names": ["q"], "network": {"type": "MLP", "config": {"hidden_dim": 4, "depth": 2}}},
        corrector={"force_positive_names": ["q"]})
    with fake_sfno():
        stepper = ace_amd.Stepper.from_config(cfg, ace_amd.DatasetInfo((4, 8)), device="cpu")
        with pytest.raises(NotImplementedError, match="q"):
            RolloutEngine(stepper, batch=1, n_forward_steps=2, graph=None)
