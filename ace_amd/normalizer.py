"""StandardNormalizer (fme/core/normalizer.py:122-242): (x - mean) / std and x * std + mean per name."""

from typing import Dict, Mapping

import torch


class StandardNormalizer:
    def __init__(self, means: Mapping[str, torch.Tensor], stds: Mapping[str, torch.Tensor],
                 fill_nans_on_normalize: bool = False, fill_nans_on_denormalize: bool = False, device=None):
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.means = {k: torch.as_tensor(v, dtype=torch.float).to(dev) for k, v in means.items()}
        self.stds = {k: torch.as_tensor(v, dtype=torch.float).to(dev) for k, v in stds.items()}
        self._names = set(means).intersection(stds)
        self._fill_nans_on_normalize = fill_nans_on_normalize
        self._fill_nans_on_denormalize = fill_nans_on_denormalize

    @property
    def fill_nans_on_normalize(self):
        return self._fill_nans_on_normalize

    @property
    def fill_nans_on_denormalize(self):
        return self._fill_nans_on_denormalize

    def normalize(self, tensors: Mapping[str, torch.Tensor], apply_mean: bool = True) -> Dict[str, torch.Tensor]:
        filtered = {k: v for k, v in tensors.items() if k in self._names}
        if apply_mean:
            out = {k: (t - self.means[k]) / self.stds[k] for k, t in filtered.items()}
        else:
            out = {k: t / self.stds[k] for k, t in filtered.items()}
        if self._fill_nans_on_normalize:
            out = {k: torch.where(torch.isnan(v), torch.zeros_like(v), v) for k, v in out.items()}
        return out

    def denormalize(self, tensors: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        filtered = {k: v for k, v in tensors.items() if k in self._names}
        out = {k: t * self.stds[k] + self.means[k] for k, t in filtered.items()}
        if self._fill_nans_on_denormalize:
            out = {k: torch.where(torch.isnan(v), torch.full_like(v, fill_value=float(self.means[k])), v)
                   for k, v in out.items()}
        return out

    def get_state(self):
        return {
            "means": {k: float(v.cpu().numpy().item()) for k, v in self.means.items()},
            "stds": {k: float(v.cpu().numpy().item()) for k, v in self.stds.items()},
            "fill_nans_on_normalize": self._fill_nans_on_normalize,
            "fill_nans_on_denormalize": self._fill_nans_on_denormalize,
        }

    @classmethod
    def from_state(cls, state) -> "StandardNormalizer":
        means = {k: torch.tensor(v, dtype=torch.float) for k, v in state["means"].items()}
        stds = {k: torch.tensor(v, dtype=torch.float) for k, v in state["stds"].items()}
        return cls(means=means, stds=stds, fill_nans_on_normalize=state.get("fill_nans_on_normalize", False),
                   fill_nans_on_denormalize=state.get("fill_nans_on_denormalize", False))
