"""Forcings computed from the time axis instead of read from disk (fme/ace/stepper/derived_forcings.py:9-95): today the
insolation (ace_amd/insolation.py).  ``DerivedForcingsConfig`` is the ``StepperConfig.derived_forcings`` field of a checkpoint
(single_module.py:532-539); ``ForcingDeriver`` is what ``Stepper.predict`` calls on every forcing window before the rollout
(single_module.py:1202-1203).

A forcing window here is a plain ``name -> (samples, T + 1, lat, lon)`` dict; its times ride along as a ``TimeAxis`` of shape
(samples, T + 1), either passed explicitly or as the ``time`` attribute of a ``ForcingWindow``."""
import dataclasses
from typing import Any, Dict, List, Mapping, Optional

import torch

from .insolation import Insolation, InsolationConfig
from .timeaxis import TimeAxis, as_time_axis


class ForcingWindow(dict):
    """name -> tensor forcing window that carries its time axis (the reference's BatchData.time)."""

    time: Optional[TimeAxis] = None
    derived: bool = False     # set by ForcingDeriver: the derived forcings (the insolation) of this window have been computed from its times

    def __init__(self, data: Mapping[str, torch.Tensor], time=None, derived: bool = False):
        super().__init__(data)
        self.time = as_time_axis(time)
        self.derived = bool(derived)


@dataclasses.dataclass
class DerivedForcingsConfig:
    insolation: Optional[InsolationConfig] = None

    def __post_init__(self):
        if isinstance(self.insolation, Mapping):
            self.insolation = InsolationConfig(**self.insolation)

    @classmethod
    def from_state(cls, state: Any) -> "DerivedForcingsConfig":
        if state is None:
            return cls()
        if isinstance(state, cls):
            return state
        extra = set(state) - {"insolation"}
        if extra:
            raise ValueError(f'can not match {sorted(extra)} to any data class field of "DerivedForcingsConfig"')
        return cls(insolation=state.get("insolation"))

    def build(self, dataset_info) -> "ForcingDeriver":
        if self.insolation is None:
            return ForcingDeriver(None)
        if dataset_info is None:
            raise ValueError("derived forcings need the dataset_info (timestep and horizontal coordinates)")
        return ForcingDeriver(self.insolation.build(dataset_info.timestep, getattr(dataset_info, "horizontal_coordinates", None)))

    def update_names(self, names: List[str]) -> List[str]:
        """derived_forcings.py:34-42 on a plain name list: which forcing names still have to come from data."""
        return self.insolation.update_names(names) if self.insolation is not None else list(names)

    def validate_replacement(self, replacement: "DerivedForcingsConfig") -> None:
        """derived_forcings.py:44-62: a replacement must keep the name the network was trained on."""
        if self.insolation is not None and replacement.insolation is not None:
            original = self.insolation.insolation_name
            if original != replacement.insolation.insolation_name:
                raise ValueError(f"Replacement insolation_name should match the original insolation_name ({original!r}). Got "
                                 f"{replacement.insolation.insolation_name!r}.")


class ForcingDeriver:
    """derived_forcings.py:65-95."""

    def __init__(self, insolation: Optional[Insolation]):
        self.insolation = insolation

    @property
    def needs_time(self) -> bool:
        return self.insolation is not None

    def __call__(self, forcing: Mapping[str, torch.Tensor], time=None, device=None) -> Mapping[str, torch.Tensor]:
        if self.insolation is None:
            return forcing
        time = as_time_axis(time if time is not None else getattr(forcing, "time", None))
        if time is None:
            raise ValueError(f"the stepper derives '{self.insolation.config.insolation_name}' from the time axis: pass the times of the "
                             "forcing window (time=TimeAxis of shape (samples, time levels), or a ForcingWindow)")
        example = next((v for v in forcing.values() if isinstance(v, torch.Tensor)), None)
        if example is not None and tuple(time.shape) != tuple(example.shape[:2]):
            raise ValueError(f"time axis of shape {tuple(time.shape)} for forcings of (samples, time levels) = {tuple(example.shape[:2])}")
        return ForcingWindow(self.insolation.compute(time, forcing, device=device), time, derived=True)
