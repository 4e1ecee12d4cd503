"""Multi-call diagnostics (ace_amd/multi_call.py; fme/core/step/_multi_call.py, multi_call.py) against the reference's own code
(tests/golden/make_golden_multi_call.py: names, validation outcomes, MultiCall.step around a closed-form stand-in step), and the
Stepper / checkpoint plumbing with a stub network."""
import copy
import datetime
import os

import pytest
import torch

import ace_amd
from ace_amd.multi_call import MultiCall, MultiCallConfig, get_multi_call_name
from ace_amd.step import StepArgs, StepOutput, StepperState

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_multi_call.pt")


def _fake_step_outputs(inp, nxt):        # the stand-in step of the generator, restated
    return {"ULWRFtoa": 2.0 * inp["co2"] + inp["T_0"], "USWRFsfc": inp["co2"] * inp["T_0"] - nxt["co2"],
            "T_0": inp["T_0"] + 1.0, "h_3": inp["T_0"] * 0.5 + 3.0 * nxt["co2"]}


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


def test_names_and_validation_match_the_reference(gold):
    for name, suffix, expected in gold["names"]:
        assert get_multi_call_name(name, suffix) == expected
    cfg = MultiCallConfig.from_state(gold["config"])
    assert cfg.names == gold["config_names"]
    for in_names, out_names, message in gold["validations"]:
        if message is None:
            cfg.validate(in_names, out_names)
        else:
            with pytest.raises(ValueError) as e:
                cfg.validate(in_names, out_names)
            assert str(e.value) == message
    with pytest.raises(ValueError, match="surprise"):
        MultiCallConfig.from_state({**gold["config"], "surprise": 1})
    assert MultiCallConfig.from_state(None) is None


@pytest.mark.parametrize("batched", [False, True])
def test_multi_call_step_matches_the_reference(gold, batched):
    cfg = MultiCallConfig.from_state(gold["config"])
    seen = []

    def step_method(args, wrapper):
        seen.append(args)
        return StepOutput(output=_fake_step_outputs(args.input, args.next_step_input_data), stepper_state=args.stepper_state)

    multi = cfg.build(step_method, batched=batched)
    state = StepperState(corrector_state=ace_amd.corrector.CorrectorState(global_dry_air_mass=torch.tensor([[[1.0]], [[2.0]]])))
    res = multi.step(StepArgs(input=gold["input"], next_step_input_data=gold["next"], stepper_state=state))
    assert list(res.output) == list(gold["output"])
    for k, v in gold["output"].items():
        assert torch.equal(res.output[k], v), k
    if batched:      # ONE evaluation of batch 2 x 2, the per-sample state repeated for every multiplier
        assert len(seen) == 1 and seen[0].input["co2"].shape[0] == 4
        assert torch.equal(seen[0].stepper_state.corrector_state.global_dry_air_mass.flatten(), torch.tensor([1.0, 2.0, 1.0, 2.0]))
        assert torch.equal(seen[0].input["T_0"][:2], seen[0].input["T_0"][2:])            # only the named forcing is scaled
    else:            # one evaluation per multiplier, each seeing the INCOMING state
        assert len(seen) == 2 and all(a.stepper_state is state for a in seen)
    with pytest.raises(ValueError, match="not in input or next_step_input_data"):
        multi.step(StepArgs(input={"T_0": gold["input"]["T_0"]}, next_step_input_data={}))


def _stub_stepper(multi_call=None):
    from ace_amd.registry import Module
    from ace_amd.step import NormalizationConfig, SingleModuleStep

    class Net(torch.nn.Module):          # in: [co2, T_0] -> out: [T_0, ULWRFtoa]
        def forward(self, x):
            return torch.stack([0.9 * x[:, 1] + 0.1 * x[:, 0], x[:, 0] * x[:, 1]], dim=1)

    names = ["co2", "T_0", "ULWRFtoa"]
    norm = NormalizationConfig(means={"co2": 1.0, "T_0": 0.5, "ULWRFtoa": 2.0}, stds={"co2": 2.0, "T_0": 1.5, "ULWRFtoa": 4.0})
    cfg = ace_amd.SingleModuleStepConfig(
        builder=ace_amd.ModuleSelector(type="SphericalFourierNeuralOperatorNet", config={"embed_dim": 4, "num_layers": 1}),
        in_names=["co2", "T_0"], out_names=["T_0", "ULWRFtoa"], normalization=norm)
    info = ace_amd.DatasetInfo((4, 8))
    step = SingleModuleStep(cfg, info, cfg.normalization.build(names), device="cpu")
    step.module = Module(Net(), None)
    return ace_amd.Stepper(step, dataset_info=info, multi_call=multi_call)


@pytest.mark.parametrize("batched", [False, True])
def test_stepper_reports_the_multi_call_diagnostics(batched):
    """Stepper.predict with multi-call: the diagnostics are the step re-evaluated with the scaled forcing; the state that is fed
    back and the plain outputs are untouched (multi_call.py:296-312)."""
    mc = {"forcing_name": "co2", "forcing_multipliers": {"_doubled_co2": 2.0, "_halved_co2": 0.5}, "output_names": ["ULWRFtoa"]}
    plain, multi = _stub_stepper(), _stub_stepper()
    multi.replace_multi_call(mc, batched=batched)
    assert multi.out_names == ["T_0", "ULWRFtoa", "ULWRFtoa_doubled_co2", "ULWRFtoa_halved_co2"] and plain.out_names == ["T_0", "ULWRFtoa"]
    assert float(multi.normalizer.means["ULWRFtoa_halved_co2"]) == 2.0 and float(multi.normalizer.stds["ULWRFtoa_doubled_co2"]) == 4.0
    g = torch.Generator().manual_seed(1)
    ic = {"T_0": torch.randn(2, 1, 4, 8, generator=g)}
    forcing = {"co2": torch.rand(2, 4, 4, 8, generator=g) + 1.0}
    ref, ref_state = plain.predict(ic, forcing)
    out, state = multi.predict(ic, forcing)
    assert set(out) == set(multi.out_names)
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(state["T_0"], ref_state["T_0"])
    for suffix, factor in mc["forcing_multipliers"].items():
        scaled, _ = plain.predict(ic, {"co2": factor * forcing["co2"]})
        # only step 0 is comparable directly (later steps of `scaled` start from a state the scaled forcing produced)
        torch.testing.assert_close(out["ULWRFtoa" + suffix][:, 0], scaled["ULWRFtoa"][:, 0], rtol=1e-6, atol=1e-6)
    # later steps: re-evaluate the plain rollout's states with the scaled forcing
    for s in (1, 2):
        one, _ = plain.predict({"T_0": ref["T_0"][:, s - 1:s]}, {"co2": 2.0 * forcing["co2"][:, s:s + 2]})
        torch.testing.assert_close(out["ULWRFtoa_doubled_co2"][:, s], one["ULWRFtoa"][:, 0], rtol=1e-6, atol=1e-6)
    multi.replace_multi_call(None)
    assert multi.out_names == ["T_0", "ULWRFtoa"] and multi.multi_call is None
    with pytest.raises(ValueError, match="not in output names"):
        multi.replace_multi_call({**mc, "output_names": ["nope"]})


def test_checkpoint_with_multi_call_loads_and_overrides():
    from test_checkpoint_cpu import IN, OUT, _reference_style_checkpoint
    from ace_amd.checkpoint import StepperOverrideConfig, load_stepper
    ckpt, _ = _reference_style_checkpoint(wrap_multi_call=True)
    forcing_name = [n for n in IN if n not in OUT][0]
    out_name = [n for n in OUT if n not in IN][0] if [n for n in OUT if n not in IN] else OUT[0]
    mc = {"forcing_name": forcing_name, "forcing_multipliers": {"_x2": 2.0}, "output_names": [out_name]}
    ckpt["stepper"]["config"]["step"]["config"]["config"] = mc
    loaded = load_stepper(ckpt, device="cpu")
    assert loaded.stepper.multi_call == MultiCallConfig.from_state(mc)
    assert loaded.stepper.out_names[-1] == get_multi_call_name(out_name, "_x2")
    off = load_stepper(ckpt, StepperOverrideConfig(multi_call=None), device="cpu")
    assert off.stepper.multi_call is None and off.stepper.out_names == list(OUT)
    plain, _ = _reference_style_checkpoint(wrap_multi_call=True)
    on = load_stepper(plain, StepperOverrideConfig(multi_call=mc), device="cpu")
    assert on.stepper.multi_call is not None
    bad = copy.deepcopy(ckpt)
    bad["stepper"]["config"]["step"]["config"]["config"] = {**mc, "forcing_name": OUT[0]}
    with pytest.raises(ValueError):
        load_stepper(bad, device="cpu")


def test_rollout_engine_multi_call_needs_the_step_graph_mode():
    from ace_amd.rollout import RolloutEngine
    st = _stub_stepper({"forcing_name": "co2", "forcing_multipliers": {"_x2": 2.0}, "output_names": ["ULWRFtoa"]})
    with pytest.raises(NotImplementedError, match="window"):
        RolloutEngine(st, batch=1, n_forward_steps=2, graph="window")
