"""Static spatial masking around the step (fme/core/spatial_masking.py:11-167, fme/core/spatial_mask_provider.py:70-170,
fme/core/name_and_prefix_matcher.py): a dataset's time-invariant masks ("mask_<variable>", "mask_<level>", "mask_2d"), the
replacement of masked regions of the step INPUTS by a fill value (StepperConfig.input_masking, single_module.py:615-632) and of the
step OUTPUTS by NaN where the data has no valid points (the provider's output masker).  Elementwise torch ops on the dict of fields,
outside the network - used by ``Stepper.step``; the static-buffer ``RolloutEngine`` refuses a stepper that masks."""
import dataclasses
import re
from typing import Any, Dict, List, Mapping, Optional, Union

import torch

TensorMapping = Mapping[str, torch.Tensor]
_LEVEL = re.compile(r"_(\d+)$")


class NameMatcher:
    """name_and_prefix_matcher.py: 'thetao' matches thetao and thetao_<level>; 'thetao_' matches thetao_<level>; 'thetao_3' itself."""

    def __init__(self, names_and_prefixes: Optional[List[str]] = None):
        self._patterns = []
        for name in names_and_prefixes or []:
            if name.endswith("_"):
                self._patterns.append(re.compile(rf"^{name}\d+$"))
            elif re.match(r".+_\d+$", name):
                self._patterns.append(re.compile(rf"^{name}$"))
            else:
                self._patterns += [re.compile(rf"^{name}$"), re.compile(rf"^{name}_\d+$")]

    def match(self, name: str) -> bool:
        return any(p.match(name) for p in self._patterns)


class SpatialMaskProvider:
    """spatial_mask_provider.py:70-170: 2-D masks by name; lookup order variable-specific, level-specific, "mask_2d"."""

    def __init__(self, masks: Optional[TensorMapping] = None):
        self._masks: Dict[str, torch.Tensor] = dict(masks) if masks is not None else {}
        for key in self._masks:
            if not key.startswith("mask_"):
                raise ValueError("The 'mask' TensorDict passed to SpatialMaskProvider init has non-mask tensors, including "
                                 f"{key}. Expected all keys to start with the string 'mask_'.")

    @property
    def masks(self) -> TensorMapping:
        return self._masks

    def get_mask_tensor_for(self, name: str) -> Optional[torch.Tensor]:
        own = self._masks.get(f"mask_{name}")
        if own is not None:
            return own
        level = _LEVEL.search(name)
        if level:
            return self._masks.get(f"mask_{int(level.group(1))}")
        return self._masks.get("mask_2d")

    def to(self, device) -> "SpatialMaskProvider":
        return SpatialMaskProvider({k: v.to(device) for k, v in self._masks.items()})

    def build_output_spatial_masker(self) -> "StaticSpatialMasking":
        """NaN where the mask is 0 (no valid data)."""
        return StaticSpatialMasking(mask_value=0, fill_value=float("nan"), mask=self)

    def get_state(self) -> Dict[str, Any]:
        return {"masks": dict(self._masks)}

    @classmethod
    def from_state(cls, state: Optional[Mapping[str, Any]]) -> "SpatialMaskProvider":
        return cls(dict(state["masks"]) if state and state.get("masks") else None)

    def __bool__(self) -> bool:
        return bool(self._masks)


class StaticSpatialMasking:
    """spatial_masking.py:98-150: data[name] = fill where round(mask) == mask_value, per variable with a mask, unless excluded."""

    def __init__(self, mask_value: int, fill_value: Union[float, TensorMapping], mask: SpatialMaskProvider,
                 exclude: Optional[NameMatcher] = None):
        self._value = mask_value
        self._fill = fill_value
        self._mask = mask
        self._exclude = exclude or NameMatcher()
        self._on: Dict[str, SpatialMaskProvider] = {}

    def _provider(self, device) -> SpatialMaskProvider:
        key = str(device)
        if key not in self._on:
            self._on[key] = self._mask.to(device)
        return self._on[key]

    def _fill_for(self, name: str):
        if isinstance(self._fill, Mapping):
            if name not in self._fill:
                raise KeyError(f"StaticSpatialMasking was initialized with a fill_value mapping but the mapping is missing key '{name}'.")
            return self._fill[name]
        return self._fill

    def __call__(self, data: TensorMapping) -> Dict[str, torch.Tensor]:
        out = dict(data)
        for name, tensor in out.items():
            if self._exclude.match(name):
                continue
            mask = self._provider(tensor.device).get_mask_tensor_for(name)
            if mask is None:
                continue
            fill = self._fill_for(name)
            fill = fill.to(tensor.device, tensor.dtype) if isinstance(fill, torch.Tensor) else torch.tensor(fill, dtype=tensor.dtype, device=tensor.device)
            where = torch.round(mask).to(torch.int64).expand(tensor.shape) == self._value
            out[name] = torch.where(where, fill, tensor)
        return out


class NullSpatialMasking:
    def __call__(self, data: TensorMapping) -> Dict[str, torch.Tensor]:
        return dict(data)


@dataclasses.dataclass
class StaticSpatialMaskingConfig:
    """spatial_masking.py:44-95 (same fields)."""
    mask_value: int
    fill_value: Union[str, float] = 0.0
    exclude_names_and_prefixes: Optional[List[str]] = None

    def __post_init__(self):
        if self.mask_value not in (0, 1):
            raise ValueError(f"mask_value must be either 0 or 1, but got {self.mask_value}")
        if isinstance(self.fill_value, str) and self.fill_value != "mean":
            raise ValueError(f"fill_value must be a float or 'mean', got {self.fill_value!r}")

    @classmethod
    def from_state(cls, state) -> Optional["StaticSpatialMaskingConfig"]:
        if state is None or isinstance(state, cls):
            return state
        extra = set(state) - {"mask_value", "fill_value", "exclude_names_and_prefixes"}
        if extra:
            raise ValueError(f'can not match {sorted(extra)} to any data class field of "StaticSpatialMaskingConfig"')
        return cls(**state)

    def build(self, mask: SpatialMaskProvider, means: Optional[TensorMapping] = None) -> StaticSpatialMasking:
        exclude = NameMatcher(self.exclude_names_and_prefixes)
        if isinstance(self.fill_value, (int, float)) and not isinstance(self.fill_value, bool):
            return StaticSpatialMasking(self.mask_value, float(self.fill_value), mask, exclude)
        if means is None:
            raise ValueError("fill_values mapping required by build unless configured fill_value is a float.")
        return StaticSpatialMasking(self.mask_value, means, mask, exclude)
