"""Window I/O around the rollout (ace_amd/inference.py) on CPU with a stand-in predict function: the forcing-window rule of
InferenceDataset (fme/ace/data_loading/inference.py:291-357), the Looper / run_inference call order
(fme/core/generics/inference.py:25-66, 117-166), the writer files and the state that rides on the prognostic state."""
import os

import pytest
import numpy as np
import torch

from ace_amd.inference import ForcingWindows, InferenceData, Looper, TensorFileWriter, run_inference
from ace_amd.stepper import PrognosticState


def fake_predict(ic, forcing, compute_derived_variables=False):
    """x_{t+1} = x_t + f_t ; diagnostic d_{t+1} = 2 f_{t+1}; counts the windows on the carried state."""
    x = ic["x"][:, 0]
    f = forcing["f"]
    outs = []
    for t in range(f.shape[1] - 1):
        x = x + f[:, t]
        outs.append(x)
    out = {"x": torch.stack(outs, dim=1), "d": 2 * f[:, 1:]}
    state = PrognosticState({"x": out["x"][:, -1:]})
    state.stepper_state = (getattr(ic, "stepper_state", None) or 0) + 1
    return out, state


def test_forcing_windows_follow_the_reference_rule():
    f = torch.arange(3 * 11, dtype=torch.float32).reshape(3, 11, 1, 1).expand(3, 11, 2, 4).contiguous()
    fw = ForcingWindows({"f": f}, total_forward_steps=10, forward_steps_in_memory=4, device="cpu")
    assert len(fw) == 3
    wins = list(fw)
    assert [w["f"].shape[1] for w in wins] == [5, 5, 3]            # T + 1, T + 1, cut at total + 1
    assert torch.equal(wins[0]["f"][:, -1], wins[1]["f"][:, 0])    # consecutive windows share one time level
    assert torch.equal(wins[2]["f"], f[:, 8:11])
    sub = list(ForcingWindows({"f": f}, 10, 4, device="cpu", members=[0, 2]))     # member g on rank g % world
    assert torch.equal(sub[1]["f"], f[[0, 2], 4:9])
    assert len(ForcingWindows({"f": f}, 8, 4, device="cpu")) == 2
    with pytest.raises(ValueError, match="number of forward inference steps"):
        ForcingWindows({"f": f}, 11, 4, device="cpu")
    with pytest.raises(ValueError):
        ForcingWindows({"f": f[:, :, 0]}, 4, 2, device="cpu")


def test_looper_chains_windows_through_the_prognostic_state():
    torch.manual_seed(0)
    f = torch.randn(2, 8, 3, 5)
    ic = {"x": torch.randn(2, 1, 3, 5)}
    one, _ = fake_predict(ic, {"f": f})
    looper = Looper(fake_predict, InferenceData(ic, ForcingWindows({"f": f}, 7, 3, device="cpu")))
    assert len(looper) == 3
    wins = list(looper)
    assert [w["x"].shape[1] for w in wins] == [3, 3, 1]
    torch.testing.assert_close(torch.cat([w["x"] for w in wins], dim=1), one["x"])
    assert torch.equal(torch.cat([w["d"] for w in wins], dim=1), one["d"])
    final = looper.get_prognostic_state()
    assert torch.equal(final["x"], wins[-1]["x"][:, -1:]) and final.stepper_state == 3


def test_looper_asks_for_derived_variables_when_told():
    seen = []

    def predict(ic, forcing, compute_derived_variables=False):
        seen.append(compute_derived_variables)
        return fake_predict(ic, forcing)

    data = lambda: InferenceData({"x": torch.zeros(1, 1, 2, 2)}, ForcingWindows({"f": torch.zeros(1, 3, 2, 2)}, 2, 1, device="cpu"))
    run_inference(predict, data())
    run_inference(predict, data(), compute_derived_variables=True)
    assert seen == [False, False, True, True]


def test_run_inference_call_order_and_files(tmp_path):
    calls = []

    class Agg:
        def record_initial_condition(self, initial_condition):
            calls.append("agg.ic")
            return ["ic"]

        def record_batch(self, data):
            calls.append(f"agg.batch{data['x'].shape[1]}")
            return ["b"]

    class Writer(TensorFileWriter):
        def write(self, data, filename):
            calls.append("write." + filename)
            super().write(data, filename)

        def append_batch(self, batch):
            calls.append("append")
            super().append_batch(batch)

    f = torch.randn(1, 6, 2, 2)
    ic = {"x": torch.zeros(1, 1, 2, 2)}
    logs = []
    state = run_inference(fake_predict, InferenceData(ic, ForcingWindows({"f": f}, 5, 2, device="cpu")), Agg(),
                          Writer(str(tmp_path), names=["x"]), logs.extend)
    assert calls == ["agg.ic", "write.initial_condition.nc", "append", "agg.batch2", "append", "agg.batch2", "append",
                     "agg.batch1", "write.restart.nc"]
    assert logs == ["ic", "b", "b", "b"]
    restart = torch.load(os.path.join(tmp_path, "restart.pt"), weights_only=True)
    assert set(restart) == {"x"} and torch.equal(restart["x"], state["x"]) and restart["x"].shape == (1, 1, 2, 2)
    series = torch.load(os.path.join(tmp_path, "autoregressive_predictions.pt"), weights_only=True)
    assert set(series) == {"x"} and series["x"].shape == (1, 5, 2, 2)
    torch.testing.assert_close(series["x"][:, -1:], restart["x"])
    assert torch.equal(torch.load(os.path.join(tmp_path, "initial_condition.pt"), weights_only=True)["x"], ic["x"])


def test_forcing_windows_tile_the_record_for_any_lengths():
    """property: for every (total, T) the windows are [iT, min(iT + T, total)] inclusive, consecutive windows share exactly
    one time level, and together they cover every time level once (plus the shared ones)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(total=st.integers(1, 40), T=st.integers(1, 12))
    def check(total, T):
        f = torch.arange(total + 1, dtype=torch.float32).reshape(1, total + 1, 1, 1)
        wins = [w["f"][0, :, 0, 0].tolist() for w in ForcingWindows({"f": f}, total, T, device="cpu")]
        assert len(wins) == -(-total // T)
        assert wins[0][0] == 0 and wins[-1][-1] == total
        for a, b in zip(wins, wins[1:]):
            assert a[-1] == b[0] and len(a) == T + 1
        steps = [x for w in wins for x in w[1:]]
        assert steps == list(map(float, range(1, total + 1)))

    check()


def test_forcing_windows_carry_their_time_slices():
    """A record with a time axis yields ForcingWindows whose times are the window's slice (consecutive windows share one level),
    member selection applies to the times as to the data."""
    import datetime

    from ace_amd.derived_forcings import ForcingWindow
    from ace_amd.timeaxis import TimeAxis
    total, T = 7, 3
    forcing = {"f": torch.arange(3 * 8, dtype=torch.float32).reshape(3, 8, 1, 1).expand(3, 8, 2, 4).contiguous()}
    time = TimeAxis.regular((2000, 1, 1), datetime.timedelta(hours=6), 8, 3)
    time = TimeAxis(time.calendar, time.us + np.arange(3)[:, None] * 86_400_000_000)      # member m starts m days later
    wins = list(ForcingWindows(forcing, total, T, device="cpu", members=[2, 0], time=time))
    assert len(wins) == 3 and all(isinstance(w, ForcingWindow) for w in wins)
    assert [w.time.shape for w in wins] == [(2, 4), (2, 4), (2, 2)]
    assert wins[1].time[:, :1] == wins[0].time[:, -1:]
    assert wins[0].time == time[np.array([2, 0])][:, 0:4]
    assert torch.equal(wins[2]["f"][:, :, 0, 0], forcing["f"][[2, 0]][:, 6:8, 0, 0])
    plain = list(ForcingWindows(forcing, total, T, device="cpu"))
    assert not isinstance(plain[0], ForcingWindow)
    with pytest.raises(ValueError, match="time must be"):
        ForcingWindows(forcing, total, T, device="cpu", time=time[:, :5])


def test_documented_switches_exist_in_the_code():
    """every ACE_* environment switch DESIGN.md documents is read somewhere in the sources (and vice versa for getenv)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    design = open(os.path.join(root, "DESIGN.md")).read()
    documented = set(re.findall(r"`(ACE_[A-Z0-9_]+)(?:=[^`]*)?`", design))
    src = ""
    for sub in ("ace_amd", os.path.join("ace_amd", "csrc")):
        d = os.path.join(root, sub)
        for f in os.listdir(d):
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src += open(os.path.join(d, f)).read()
    read_by_code = set(re.findall(r'getenv\("(ACE_[A-Z0-9_]+)"\)', src)) | set(re.findall(r'environ[^"\n]*"(ACE_[A-Z0-9_]+)"', src))
    compile_time = set(re.findall(r"#ifndef (ACE_[A-Z0-9_]+)", src))
    missing = {n for n in documented if n not in read_by_code and n not in compile_time and n not in src}
    assert not missing, f"documented but not in the sources: {sorted(missing)}"
    undocumented = {n for n in read_by_code if n not in design}
    assert not undocumented, f"read by the code but not documented in DESIGN.md: {sorted(undocumented)}"
