"""Emit tests/golden/gen_multi_call.pt from the reference's own multi-call code (fme/core/step/_multi_call.py, imported through
oracle/ref_loader.load_stepper_ref - build container only): the suffixed names, the validation outcomes, and the outputs of
``MultiCall.step`` around a deterministic stand-in step method (a closed-form function of the inputs, restated in the test), so
that the loop / scaling / naming / state semantics are pinned on the reference itself without a network.

Run:  python tests/golden/make_golden_multi_call.py"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_loader  # noqa: E402


def fake_step_outputs(inp, nxt):
    """the stand-in step: closed form in the (possibly scaled) inputs"""
    return {"ULWRFtoa": 2.0 * inp["co2"] + inp["T_0"], "USWRFsfc": inp["co2"] * inp["T_0"] - nxt["co2"],
            "T_0": inp["T_0"] + 1.0, "h_3": inp["T_0"] * 0.5 + 3.0 * nxt["co2"]}


def main():
    ref_loader.load_stepper_ref()
    import importlib
    mc = importlib.import_module("fme.core.step._multi_call")
    args_mod = importlib.import_module("fme.core.step.args")
    out_mod = importlib.import_module("fme.core.step.output")
    state_mod = importlib.import_module("fme.core.stepper_state")

    names = [(n, s, mc.get_multi_call_name(n, s)) for n, s in [("foo", "_with_quartered_co2"), ("bar_0", "_with_quartered_co2"),
                                                                ("air_temperature_7", "_x"), ("a_b", "_2x"), ("level_12_", "_q")]]
    cfg = mc.MultiCallConfig(forcing_name="co2", forcing_multipliers={"_quadrupled_co2": 4.0, "_halved_co2": 0.5},
                             output_names=["ULWRFtoa", "h_3"])
    validations = []
    for in_names, out_names in [(["co2", "T_0"], ["ULWRFtoa", "h_3", "T_0"]), (["T_0"], ["ULWRFtoa", "h_3"]),
                                (["co2", "T_0"], ["ULWRFtoa", "h_3", "co2"]), (["co2"], ["ULWRFtoa"]),
                                (["co2", "ULWRFtoa_halved_co2"], ["ULWRFtoa", "h_3"]),
                                (["co2"], ["ULWRFtoa", "h_3", "h_quadrupled_co2_3"])]:
        try:
            cfg.validate(in_names, out_names)
            validations.append((in_names, out_names, None))
        except ValueError as e:
            validations.append((in_names, out_names, str(e)))

    g = torch.Generator().manual_seed(0)
    inp = {"co2": torch.rand(2, 4, 8, generator=g) + 1.0, "T_0": torch.randn(2, 4, 8, generator=g)}
    nxt = {"co2": torch.rand(2, 4, 8, generator=g) + 1.0}
    calls = []

    def step_method(args, wrapper):
        calls.append(float(args.input["co2"].mean() / inp["co2"].mean()))
        return out_mod.StepOutput(output=fake_step_outputs(args.input, args.next_step_input_data), stepper_state=args.stepper_state)

    multi = cfg.build(step_method)
    sargs = args_mod.StepArgs(input=inp, next_step_input_data=nxt, labels=None)
    res = multi.step(sargs)
    out = os.path.join(HERE, "gen_multi_call.pt")
    torch.save({"names": names, "config": {"forcing_name": "co2", "forcing_multipliers": dict(cfg.forcing_multipliers),
                                           "output_names": list(cfg.output_names)},
                "config_names": list(cfg.names), "validations": validations, "input": inp, "next": nxt,
                "output": {k: v.clone() for k, v in res.output.items()}, "call_factors": calls}, out)
    print("names", names)
    print("config names", cfg.names, "calls", calls)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
