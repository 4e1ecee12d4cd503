"""Multi-call diagnostics (fme/core/step/_multi_call.py:22-188, fme/core/step/multi_call.py:67-318): the step is evaluated again
with one forcing (CO2) multiplied by each of a set of factors and selected outputs (radiative fluxes) of those evaluations are
reported under suffixed names - diagnostics only: the state that is fed back, the corrector state and the prognostic outputs are
those of the plain evaluation.

The reference wraps its step in a ``MultiCallStep``; here the ``Stepper`` holds the (optional) ``MultiCall`` beside its single-module
step, which keeps every other consumer of the step object (rollout engine, physics, checkpoint state) unchanged.  Two evaluation
orders, same numbers per sample:

  * sequential (default): one more ``step`` per multiplier, as the reference does;
  * ``batched=True``: the K perturbed inputs are concatenated along the batch dimension and evaluated by ONE step of batch K B -
    the network's weights (1.8 GB at the ACE2 shape) stream once instead of K times.  Every stage of the step is per sample
    (instance norms, corrector global means, ocean), so the results are the sequential ones up to the compensated-fp16 scaling,
    which is taken over the whole batch (differences at the 1e-7 level)."""
import dataclasses
import re
from typing import Callable, Dict, List, Mapping, Optional

import torch

from .step import StepArgs, StepOutput

_LEVEL = re.compile(r"_(\d+)$")


def get_multi_call_name(name: str, suffix: str) -> str:
    """'foo' + '_x' -> 'foo_x';  a vertical-level label stays last: 'bar_0' + '_x' -> 'bar_x_0' (_multi_call.py:22-46)."""
    m = _LEVEL.search(name)
    if m is None:
        return name + suffix
    return name[: m.start()] + suffix + m.group(0)


@dataclasses.dataclass
class MultiCallConfig:
    """_multi_call.py:49-127 (same fields)."""
    forcing_name: str
    forcing_multipliers: Dict[str, float]
    output_names: List[str]

    def __post_init__(self):
        self.forcing_multipliers = dict(self.forcing_multipliers)
        self.output_names = list(self.output_names)

    @classmethod
    def from_state(cls, state) -> Optional["MultiCallConfig"]:
        if state is None or isinstance(state, cls):
            return state
        extra = set(state) - {"forcing_name", "forcing_multipliers", "output_names"}
        if extra:
            raise ValueError(f'can not match {sorted(extra)} to any data class field of "MultiCallConfig"')
        missing = {"forcing_name", "forcing_multipliers", "output_names"} - set(state)
        if missing:
            raise ValueError(f'missing value for field(s) {sorted(missing)} of "MultiCallConfig"')
        return cls(**state)

    def get_multi_called_names(self, name: str) -> List[str]:
        return [get_multi_call_name(name, suffix) for suffix in self.forcing_multipliers]

    @property
    def names(self) -> List[str]:
        return [n for name in self.output_names for n in self.get_multi_called_names(name)]

    def validate(self, in_names: List[str], out_names: List[str]) -> None:
        if self.forcing_name not in in_names:
            raise ValueError(f"forcing name {self.forcing_name} not in input names. It is required as a forcing given provided "
                             "radiation multi call configuration.")
        if self.forcing_name in out_names:
            raise ValueError(f"forcing name {self.forcing_name} is in the output names, but it must be a forcing variable, not an "
                             "output.")
        for name in self.output_names:
            if name not in out_names:
                raise ValueError(f"{name} not in output names. It is required as an output given provided radiation multi call "
                                 "configuration.")
        for name in self.names:
            if name in in_names:
                raise ValueError(f"The multi-call output {name} is already in in_names. This will lead to a conflict--please rename "
                                 "the input or use a different multi-call suffix label.")
            if name in out_names:
                raise ValueError(f"The multi-call output {name} is already in out_names. This will lead to a conflict--please rename "
                                 "the output or use a different multi-call suffix label.")

    def build(self, step_method: Callable, batched: bool = False) -> "MultiCall":
        return MultiCall(self, step_method, batched=batched)

    def extend_normalizer(self, normalizer):
        """multi_call.py:227-250: the suffixed names are normalised like their base names."""
        from .normalizer import StandardNormalizer
        means, stds = dict(normalizer.means), dict(normalizer.stds)
        for name in self.output_names:
            if name not in means or name not in stds:
                raise ValueError(f"Normalizer does not contain {name} present in multi-call output names")
            for mc in self.get_multi_called_names(name):
                means[mc], stds[mc] = means[name], stds[name]
        return StandardNormalizer(means, stds, normalizer.fill_nans_on_normalize, normalizer.fill_nans_on_denormalize)


def _scaled(data: Mapping[str, torch.Tensor], name: str, factor: float) -> Dict[str, torch.Tensor]:
    if name not in data:
        return dict(data)
    return {**data, name: factor * data[name]}


def _repeat_state(state, k: int):
    """the per-sample stepper state for a batch made of k copies of the samples"""
    if state is None:
        return None
    cs = getattr(state, "corrector_state", None)
    if cs is None:
        return state
    fields = {}
    for f in dataclasses.fields(cs):
        v = getattr(cs, f.name)
        fields[f.name] = torch.cat([v] * k, dim=0) if isinstance(v, torch.Tensor) else v
    return dataclasses.replace(state, corrector_state=type(cs)(**fields))


class MultiCall:
    """_multi_call.py:130-188."""

    def __init__(self, config: MultiCallConfig, step_method: Callable, batched: bool = False):
        self.forcing_name = config.forcing_name
        self.forcing_multipliers = dict(config.forcing_multipliers)
        self.output_names = list(config.output_names)
        self._names = config.names
        self._step = step_method
        self.batched = batched

    @property
    def names(self) -> List[str]:
        return self._names

    def step(self, args: StepArgs, wrapper: Callable = lambda x: x) -> StepOutput:
        if self.forcing_name not in args.input and self.forcing_name not in args.next_step_input_data:
            raise ValueError(f"forcing name {self.forcing_name} not in input or next_step_input_data")
        # A seeded rollout (StepperState.random_state) draws one (real, imaginary) noise pair per CALL, in call order
        # (_multi_call.py:170-188 calls the module once per multiplier): the batched form would draw one k B-sized batch instead -
        # other numbers from the same generator - so it is only taken when no generator is carried.
        seeded = getattr(args.stepper_state, "random_state", None) is not None
        if self.batched and len(self.forcing_multipliers) > 1 and not seeded:
            return self._step_batched(args, wrapper)
        predictions: Dict[str, torch.Tensor] = {}
        state = args.stepper_state
        for suffix, factor in self.forcing_multipliers.items():
            scaled = args.apply_input_process_func(lambda d, f=factor: _scaled(d, self.forcing_name, f))
            result = self._step(scaled, wrapper)
            state = result.stepper_state
            for name in self.output_names:
                predictions[get_multi_call_name(name, suffix)] = result.output[name]
        return StepOutput(output=predictions, stepper_state=state)

    def _step_batched(self, args: StepArgs, wrapper: Callable) -> StepOutput:
        factors = list(self.forcing_multipliers.items())
        k = len(factors)

        def stack(data):
            per = [_scaled(data, self.forcing_name, f) for _, f in factors]
            return {name: torch.cat([p[name] for p in per], dim=0) for name in data}

        labels = args.labels
        if labels is not None:
            labels = type(labels)(torch.cat([labels.tensor] * k, dim=0), list(labels.names))
        if args.data_mask is not None:
            raise NotImplementedError("data masks are outside the accelerated hot path")
        big = StepArgs(input=stack(args.input), next_step_input_data=stack(args.next_step_input_data), labels=labels,
                       data_mask=None, stepper_state=_repeat_state(args.stepper_state, k))
        result = self._step(big, wrapper)
        b = next(iter(args.input.values())).shape[0]
        predictions = {}
        for i, (suffix, _) in enumerate(factors):
            for name in self.output_names:
                predictions[get_multi_call_name(name, suffix)] = result.output[name][i * b:(i + 1) * b]
        return StepOutput(output=predictions, stepper_state=None)      # (discarded by the caller, as the reference's is)
