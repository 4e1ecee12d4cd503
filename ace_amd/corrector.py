"""Post-step corrector, the part that needs no dataset geometry.

Mirror of ``AtmosphereCorrectorConfig`` (fme/core/corrector/atmosphere.py:223-398): same field set and defaults, so
a reference step config round-trips.  ``force_positive_names`` (fme/core/corrector/utils.py:26-80) is implemented;
the conservation closures (dry air, moisture, energy: atmosphere.py:404-700) need area weights and a vertical
coordinate from the dataset and raise ``NotImplementedError`` when requested - they are never silently skipped.
"""
import dataclasses
from typing import Any, Dict, List, Mapping, Optional

import torch

TensorMapping = Mapping[str, torch.Tensor]
TensorDict = Dict[str, torch.Tensor]


def force_positive(data: TensorMapping, names: List[str]) -> TensorDict:
    """fme/core/corrector/utils.py:26-44 (inference: no straight-through gradient): only the clamped fields."""
    return {name: torch.clamp(data[name], min=0.0) for name in names}


@dataclasses.dataclass
class AtmosphereCorrectorConfig:
    conserve_dry_air: bool = False
    zero_global_mean_moisture_advection: bool = False
    moisture_budget_correction: Optional[str] = None
    force_positive_names: List[str] = dataclasses.field(default_factory=list)
    total_energy_budget_correction: Optional[Any] = None
    keep_gradient_through_clamps: bool = False
    clip_frozen_precipitation: bool = False

    @classmethod
    def from_state(cls, state: Optional[Mapping[str, Any]]) -> "AtmosphereCorrectorConfig":
        if state is None:
            return cls()
        state = dict(state)
        if set(state) == {"type", "config"}:           # CorrectorSelector form (fme/core/corrector/registry.py)
            if state["type"] != "atmosphere_corrector":
                raise NotImplementedError(f"corrector type '{state['type']}' is outside the accelerated hot path")
            state = dict(state["config"])
        unknown = set(state) - {f.name for f in dataclasses.fields(cls)}
        if unknown:
            raise ValueError(f"unknown corrector fields: {sorted(unknown)}")
        return cls(**state)

    def unsupported(self) -> List[str]:
        out = []
        if self.conserve_dry_air:
            out.append("conserve_dry_air")
        if self.zero_global_mean_moisture_advection:
            out.append("zero_global_mean_moisture_advection")
        if self.moisture_budget_correction is not None:
            out.append("moisture_budget_correction")
        if self.total_energy_budget_correction is not None:
            out.append("total_energy_budget_correction")
        return out

    def get_corrector(self, dataset_info=None, ignore_unsupported: bool = False) -> Optional["AtmosphereCorrector"]:
        missing = self.unsupported()
        if missing and not ignore_unsupported:
            raise NotImplementedError(
                "corrector options outside the accelerated hot path (need area weights / vertical coordinate): "
                + ", ".join(missing))
        if not self.force_positive_names:
            return None
        return AtmosphereCorrector(list(self.force_positive_names))


class AtmosphereCorrector:
    """CorrectionSequence with the one geometry-free correction (atmosphere.py:349-398: ForcePositive goes first)."""

    def __init__(self, force_positive_names: List[str]):
        self.force_positive_names = force_positive_names

    def __call__(self, input_data: TensorMapping, gen_data: TensorMapping, forcing_data: TensorMapping) -> TensorDict:
        return {**gen_data, **force_positive(gen_data, self.force_positive_names)}
