"""Window I/O around the rollout (second half): the forcing-window feeder, the looper that chains
windows through the prognostic state, and the inference driver with its writer / aggregator call order.

Mirrors, on plain ``name -> tensor`` dicts:

  * ``InferenceDataset._get_batch_data`` / ``__len__`` (fme/ace/data_loading/inference.py:291-357): window i is the time slice
    ``[i T, i T + T + 1)`` of the forcing record (consecutive windows share one time level), the last window is cut at
    ``total_forward_steps + 1``, there are ``ceil(total / T)`` windows; member g is processed by rank ``g % world``;
  * ``Looper`` and ``run_inference`` (fme/core/generics/inference.py:25-66, 117-166): initial condition -> aggregator and
    ``writer.write(ic, "initial_condition.nc")``, then per window ``predict -> writer.append_batch -> aggregator.record_batch``,
    finally ``writer.write(prognostic_state, "restart.nc")``.

netCDF / xarray are not on the path (and not in this image): ``TensorFileWriter`` keeps the reference's writer interface and file
stems but stores ``torch.save`` archives (``restart.pt`` holds exactly the prognostic names and shapes ``restart.nc`` would).
Derived variables (``compute_derived_variables``) are torch ops on the window's output series (ace_amd/derived_variables.py).

The feeder keeps one window ahead: while window i runs on the compute stream, window i + 1 is copied host -> HBM from pinned
memory on a side stream (85 MB for a 40-step window of 8 forcing fields at 1 degree: ~2 ms at PCIe rates against ~330 ms of
compute), so the PCIe transfer never sits on the step path."""
import math
import os
from typing import Any, Callable, Dict, Iterator, List, Mapping, Optional, Sequence, Tuple

import torch

TensorDict = Dict[str, torch.Tensor]


class ForcingWindows:
    """Iterable over the forcing windows of a rollout, with a one-window-ahead asynchronous upload.

    forcing: name -> (n_members, total_forward_steps + 1 [or more], H, W) host tensors (time level 0 = the initial time).
    members: the member indices this rank processes (``Distributed.local_members``); default: all.
    time: the record's ``TimeAxis`` (n_members, total_forward_steps + 1 [or more]) - every window then is a ``ForcingWindow``
    carrying its slice, which is what a stepper with derived forcings (the insolation) reads."""

    def __init__(self, forcing: Mapping[str, torch.Tensor], total_forward_steps: int, forward_steps_in_memory: int,
                 device=None, members: Optional[Sequence[int]] = None, pin_memory: Optional[bool] = None, time=None):
        if total_forward_steps < 1 or forward_steps_in_memory < 1:
            raise ValueError("total_forward_steps and forward_steps_in_memory must be positive")
        for name, t in forcing.items():
            if t.ndim != 4:
                raise ValueError(f"forcing '{name}' must be (members, time, lat, lon), got {tuple(t.shape)}")
            if t.shape[1] < total_forward_steps + 1:
                raise ValueError(
                    f"The number of forward inference steps ({total_forward_steps}) must be less than or equal to the number "
                    f"of possible steps in the forcing record of '{name}' ({t.shape[1] - 1})")
        self._T = int(forward_steps_in_memory)
        self._total = int(total_forward_steps)
        self._device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        cuda = self._device.type == "cuda"
        pin = cuda if pin_memory is None else (pin_memory and cuda)
        idx = None if members is None else torch.as_tensor(list(members), dtype=torch.long)
        self._host: TensorDict = {}
        for name, t in forcing.items():
            t = t.detach().to("cpu", torch.float32)
            if idx is not None:
                t = t.index_select(0, idx)
            t = t[:, : self._total + 1].contiguous()
            self._host[name] = t.pin_memory() if pin else t
        self._copy_stream = torch.cuda.Stream(device=self._device) if cuda else None
        self._time = None
        if time is not None:
            from .timeaxis import as_time_axis
            t = as_time_axis(time)
            if t.ndim != 2 or t.shape[1] < self._total + 1:
                raise ValueError(f"time must be (members, >= {self._total + 1} time levels), got {tuple(t.shape)}")
            if idx is not None:
                t = t[idx.numpy()]
            self._time = t[:, : self._total + 1]

    def _with_time(self, win: TensorDict, index: int):
        if self._time is None:
            return win
        from .derived_forcings import ForcingWindow
        return ForcingWindow(win, self._time[:, self.window_slice(index)])

    def __len__(self) -> int:
        return int(math.ceil(self._total / self._T))

    def window_slice(self, index: int) -> slice:
        start = index * self._T
        return slice(start, min(start + self._T + 1, self._total + 1))

    def _upload(self, index: int):
        sl = self.window_slice(index)
        if self._copy_stream is None:
            return {k: v[:, sl].to(self._device) for k, v in self._host.items()}, None
        with torch.cuda.stream(self._copy_stream):
            win = {k: v[:, sl].to(self._device, non_blocking=True) for k, v in self._host.items()}
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return win, done

    def __iter__(self) -> Iterator[TensorDict]:
        n = len(self)
        nxt = self._upload(0)
        for i in range(n):
            win, done = nxt
            nxt = self._upload(i + 1) if i + 1 < n else None      # next window's copy runs under this window's compute
            if done is not None:
                torch.cuda.current_stream(self._device).wait_event(done)
                for t in win.values():
                    t.record_stream(torch.cuda.current_stream(self._device))
            yield self._with_time(win, i)


class InferenceData:
    """``InferenceDataABC``: an initial condition and an iterable of aligned forcing windows."""

    def __init__(self, initial_condition: TensorDict, loader):
        self.initial_condition = initial_condition
        self.loader = loader


class Looper:
    """fme/core/generics/inference.py:25-66."""

    def __init__(self, predict: Callable, data: InferenceData, compute_derived_variables: bool = False):
        self._predict = predict
        self._derived = compute_derived_variables      # the reference always asks for them (inference.py:58-62)
        self._prognostic_state = data.initial_condition
        self._len = len(data.loader)
        self._loader = iter(data.loader)

    def __iter__(self):
        return self

    def __len__(self) -> int:
        return self._len

    def __next__(self) -> TensorDict:
        forcing = next(self._loader)
        if self._derived:
            output, self._prognostic_state = self._predict(self._prognostic_state, forcing, compute_derived_variables=True)
        else:
            output, self._prognostic_state = self._predict(self._prognostic_state, forcing)
        return output

    def get_prognostic_state(self) -> TensorDict:
        return self._prognostic_state


class NullDataWriter:
    def write(self, data: TensorDict, filename: str):
        pass

    def append_batch(self, batch: TensorDict):
        pass

    def flush(self):
        pass


class TensorFileWriter:
    """Writer with the reference's interface (fme/core/generics/writer.py): ``write(state, "restart.nc")`` stores
    ``<dir>/restart.pt``; ``append_batch`` collects the windows (optionally a subset of names) and ``flush`` stores them
    concatenated along time as ``<dir>/autoregressive_predictions.pt``."""

    def __init__(self, directory: str, names: Optional[Sequence[str]] = None):
        self._dir = directory
        self._names = None if names is None else list(names)
        self._windows: List[TensorDict] = []
        os.makedirs(directory, exist_ok=True)

    def write(self, data: TensorDict, filename: str):
        stem = os.path.splitext(filename)[0]
        torch.save({k: v.detach().cpu() for k, v in data.items()}, os.path.join(self._dir, stem + ".pt"))

    def append_batch(self, batch: TensorDict):
        names = self._names if self._names is not None else list(batch)
        self._windows.append({k: batch[k].detach().cpu() for k in names})

    def flush(self):
        if self._windows:
            out = {k: torch.cat([w[k] for w in self._windows], dim=1) for k in self._windows[0]}
            torch.save(out, os.path.join(self._dir, "autoregressive_predictions.pt"))


def run_inference(predict: Callable, data: InferenceData, aggregator=None, writer=None,
                  record_logs: Optional[Callable[[Any], None]] = None, compute_derived_variables: bool = False):
    """fme/core/generics/inference.py:117-166 (same call order; ``aggregator`` / ``record_logs`` optional here).
    ``compute_derived_variables``: ask ``predict`` for the derived output variables (ace_amd/derived_variables.py) as the
    reference's Looper always does.  Returns the final prognostic state."""
    if writer is None:
        writer = NullDataWriter()
    looper = Looper(predict=predict, data=data, compute_derived_variables=compute_derived_variables)
    if aggregator is not None:
        logs = aggregator.record_initial_condition(initial_condition=data.initial_condition)
        if record_logs is not None:
            record_logs(logs)
    writer.write(data.initial_condition, "initial_condition.nc")
    for batch in looper:
        writer.append_batch(batch=batch)
        if aggregator is not None:
            logs = aggregator.record_batch(data=batch)
            if record_logs is not None:
                record_logs(logs)
    state = looper.get_prognostic_state()
    writer.write(state, "restart.nc")
    if hasattr(writer, "flush"):
        writer.flush()
    return state


class EnginePredict:
    """``PredictFunction`` on the static-buffer ``RolloutEngine``: one engine per window length (the last window of a
    rollout may be shorter), built on first use.  Outputs are copied out of the engine's buffers (the next window reuses them)."""

    def __init__(self, stepper, batch: int, graph: Optional[str] = "step", labels=None):
        """``labels``: the batch's labels for a label-conditioned stepper (BatchLabels, or a (batch, n_labels) tensor in the module's
        encoding) - the reference takes them from the forcing batch (fme/core/generics/inference.py); every engine of this predict
        function gets them before its first window."""
        self._stepper = stepper
        self._batch = batch
        self._graph = graph
        self._labels = labels
        self._engines: Dict[int, Any] = {}

    def set_labels(self, labels) -> None:
        self._labels = labels
        for eng in self._engines.values():
            eng.set_labels(labels)

    def __call__(self, initial_condition: TensorDict, forcing: TensorDict,
                 compute_derived_variables: bool = False) -> Tuple[TensorDict, TensorDict]:
        from .rollout import RolloutEngine
        forcing = self._stepper.forcing_deriver(forcing)      # forcings computed from the window's time axis (the insolation)
        n_steps = next(iter(forcing.values())).shape[1] - 1
        eng = self._engines.get(n_steps)
        if eng is None:
            eng = self._engines[n_steps] = RolloutEngine(self._stepper, batch=self._batch, n_forward_steps=n_steps,
                                                         graph=self._graph)
            if self._labels is not None:
                eng.set_labels(self._labels)
        out, state = eng.predict(initial_condition, forcing)
        kept = type(state)({k: v.clone() for k, v in state.items()})
        kept.stepper_state = getattr(state, "stepper_state", None)
        out = {k: v.clone() for k, v in out.items()}
        if compute_derived_variables:
            from .stepper import derive_over_window
            out = derive_over_window(self._stepper.derive_func, out, initial_condition, forcing, 1, n_steps)
        return out, kept
