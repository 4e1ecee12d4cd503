"""Stepper: the autoregressive loop (fme/ace/stepper/single_module.py:803, 1045-1075, 1124-1259).

`predict_generator` is the reference loop verbatim in behaviour: the state dict feeds
back prognostic names, input-only (forcing) names are taken at index `step` (or `step+1`
for `next_step_forcing_names`).  `predict` stacks the per-step outputs over time.
`ace_amd.rollout.RolloutEngine` is the same loop with static buffers and a captured
hipGraph per step for long rollouts."""

from collections.abc import Callable, Generator, Mapping
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .step import SingleModuleStep, SingleModuleStepConfig, StepArgs, StepOutput

TensorMapping = Mapping[str, torch.Tensor]
TensorDict = Dict[str, torch.Tensor]


class PrognosticState(dict):
    """name -> (B, 1, H, W) prognostic tensors plus the opaque per-sample ``stepper_state`` that rides with them from
    window to window (the reference keeps it on the BatchData of the prognostic state)."""

    stepper_state = None


def derive_over_window(derive_func, data: TensorDict, initial_condition: TensorMapping, forcing: TensorMapping,
                       n_ic: int, n_forward_steps: int) -> TensorDict:
    """single_module.py:1236-1245: prepend the initial condition (names it does not hold are NaN there,
    batch_data.py:889-913), compute the derived variables against the forcing of the same time levels, drop the initial
    time level again."""
    example = next(iter(initial_condition.values()))
    full = {}
    for k, v in data.items():
        head = initial_condition[k] if k in initial_condition else torch.full_like(example, float("nan"))
        full[k] = torch.cat([head.to(v.device), v], dim=1)
    f = {k: v[:, : n_ic + n_forward_steps] for k, v in forcing.items()}
    derived = derive_func(full, f)
    return {k: v[:, n_ic:] for k, v in {**full, **derived}.items() if k in full or k not in f}


class Stepper:
    TIME_DIM = 1
    CHANNEL_DIM = -3

    def __init__(self, step_obj: SingleModuleStep, derived_forcings=None, dataset_info=None, multi_call=None, input_masking=None):
        from .derived_forcings import DerivedForcingsConfig
        self._step_obj = step_obj
        # multi-call diagnostics (the reference's MultiCallStep wrapper, fme/core/step/multi_call.py): held beside the step
        self._multi_call_config = None
        self._multi_call = None
        self.replace_multi_call(multi_call)
        self._dataset_info = dataset_info
        # forcings computed from the time axis (StepperConfig.derived_forcings, single_module.py:532-539, 870)
        self._derived_forcings = DerivedForcingsConfig.from_state(derived_forcings)
        self.forcing_deriver = self._derived_forcings.build(dataset_info)
        # static spatial masking (StepperConfig.input_masking + the dataset's mask provider, single_module.py:615-632): inputs of every
        # step get a fill value in masked regions, outputs NaN where the data has no valid points
        from .masking import StaticSpatialMaskingConfig
        self._input_masking_config = StaticSpatialMaskingConfig.from_state(input_masking)
        provider = getattr(dataset_info, "mask_provider", None)
        self._input_process_func: Callable[[TensorMapping], TensorMapping] = lambda x: x
        self._output_masking: Callable[[TensorMapping], TensorDict] = lambda x: dict(x)
        self._masks = False
        if self._input_masking_config is not None:
            if provider is None:
                raise ValueError("input_masking needs the dataset's masks (dataset_info.mask_provider)")
            self._input_process_func = self._input_masking_config.build(mask=provider, means=step_obj.normalizer.means)
            self._masks = True
        if provider is not None and provider:
            self._output_masking = provider.build_output_spatial_masker()
            self._masks = True

    @classmethod
    def from_config(cls, config: SingleModuleStepConfig, dataset_info, device=None, derived_forcings=None,
                    multi_call=None, input_masking=None) -> "Stepper":
        normalizer = config.normalization.build(config._normalize_names, device=device)
        return cls(SingleModuleStep(config, dataset_info, normalizer, device=device), derived_forcings=derived_forcings,
                   dataset_info=dataset_info, multi_call=multi_call, input_masking=input_masking)

    # -- properties (single_module.py:960-1043)
    @property
    def prognostic_names(self) -> List[str]:
        return self._step_obj.prognostic_names

    @property
    def out_names(self) -> List[str]:
        """multi_call.py:175-177: the step's outputs, then the multi-call diagnostics."""
        return self._step_obj.output_names + (self._multi_call.names if self._multi_call is not None else [])

    @property
    def multi_call(self):
        return self._multi_call_config

    def replace_multi_call(self, multi_call, batched: bool = False) -> None:
        """single_module.py:1008-1018 / multi_call.py:206-207: a MultiCallConfig, its state dict, or None.  ``batched``: evaluate
        all multipliers in one step of K x batch (ace_amd/multi_call.py)."""
        from .multi_call import MultiCallConfig
        cfg = MultiCallConfig.from_state(multi_call)
        if cfg is not None:
            cfg.validate(self._step_obj.input_names, self._step_obj.output_names)
        self._multi_call_config = cfg
        self._multi_call = cfg.build(self._step_obj.step, batched=batched) if cfg is not None else None
        self._extended_normalizer = None      # rebuilt on first use (see `normalizer`)

    @property
    def n_ic_timesteps(self) -> int:
        return 1

    @property
    def modules(self) -> nn.ModuleList:
        return self._step_obj.modules

    @property
    def normalizer(self):
        if self._multi_call_config is not None:
            # extended once per multi-call configuration and per underlying normaliser, not on every access
            base = self._step_obj.normalizer
            cached = getattr(self, "_extended_normalizer", None)
            if cached is None or cached[0] is not base:
                cached = (base, self._multi_call_config.extend_normalizer(base))
                self._extended_normalizer = cached
            return cached[1]
        return self._step_obj.normalizer

    @property
    def _input_only_names(self) -> set:
        return set(self._step_obj.input_names).difference(self._step_obj.output_names)

    # -- inference-time replacements (single_module.py:967-996): same weights, new post-step behaviour
    def replace_ocean(self, ocean):
        """Replace the ocean model (an ``OceanConfig``, its state dict, or None)."""
        import dataclasses
        step = self._step_obj
        keep = getattr(step._config, "_ignore_unsupported", False)
        step._config = dataclasses.replace(step._config, ocean=ocean)      # __post_init__ validates / converts
        step._config._ignore_unsupported = keep
        cfg = step._config
        step._ocean = (cfg.ocean.build(list(cfg.in_names), list(cfg.out_names), step._timestep)
                       if cfg.ocean is not None else None)

    def replace_prescribed_prognostic_names(self, names: List[str]) -> None:
        import dataclasses
        step = self._step_obj
        keep = getattr(step._config, "_ignore_unsupported", False)
        step._config = dataclasses.replace(step._config, prescribed_prognostic_names=list(names))
        step._config._ignore_unsupported = keep

    def replace_derived_forcings(self, derived_forcings) -> None:
        """single_module.py:998-1006: new derived-forcing configuration (same insolation name as trained on)."""
        from .derived_forcings import DerivedForcingsConfig
        new = DerivedForcingsConfig.from_state(derived_forcings)
        self._derived_forcings.validate_replacement(new)
        self._derived_forcings = new
        self.forcing_deriver = new.build(self._dataset_info)

    @property
    def derived_forcings(self):
        return self._derived_forcings

    def forcing_names_from_data(self) -> List[str]:
        """The input-only names a forcing record has to hold (StepperConfig.get_forcing_window_data_requirements,
        single_module.py:555-573): a derived forcing is computed, a named solar constant is read instead."""
        return self._derived_forcings.update_names(sorted(self._input_only_names))

    def get_prescribed_prognostic_names(self) -> List[str]:
        return list(self._step_obj.config.prescribed_prognostic_names)

    def set_eval(self):
        for m in self.modules:
            m.eval()

    def step(self, args: StepArgs, wrapper: Callable[[nn.Module], nn.Module] = lambda x: x) -> StepOutput:
        """single_module.py:1045-1075."""
        from .rand import use_generator
        args = args.apply_input_process_func(self._input_process_func)
        # single_module.py:1063-1068: a seeded rollout's generator is active around every network call of the step (the
        # multi-call copies run inside the wrapped step there, i.e. under the same generator, in call order)
        random_state = args.stepper_state.random_state if args.stepper_state is not None else None
        with use_generator(None if random_state is None else random_state.generator):
            result = self._step_obj.step(args=args, wrapper=wrapper)
            output = result.output
            if self._multi_call is not None:     # multi_call.py:296-312: its own state and diagnostics are discarded
                output = {**self._multi_call.step(args=args, wrapper=wrapper).output, **output}
        # the wrapped step's corrector diagnostics travel with the output, masked the same way (single_module.py:1063-1075)
        diags = result.corrector_diagnostics
        return StepOutput(output=self._output_masking(output), stepper_state=result.stepper_state,
                          corrector_diagnostics=self._output_masking(dict(diags)) if diags else {})

    def predict_generator(self, ic_dict: TensorMapping, forcing_dict: TensorMapping, n_forward_steps: int,
                          labels=None, data_mask=None, stepper_state=None) -> Generator[StepOutput, None, None]:
        """single_module.py:1124-1167."""
        state = {k: ic_dict[k].squeeze(self.TIME_DIM) for k in ic_dict}
        for step in range(n_forward_steps):
            input_forcing = {
                k: (forcing_dict[k][:, step] if k not in self._step_obj.next_step_forcing_names
                    else forcing_dict[k][:, step + 1])
                for k in self._input_only_names
            }
            next_step_input_dict = {k: forcing_dict[k][:, step + 1] for k in self._step_obj.next_step_input_names}
            input_data = {**state, **input_forcing}
            result = self.step(StepArgs(input=input_data, next_step_input_data=next_step_input_dict, labels=labels,
                                        data_mask=data_mask, stepper_state=stepper_state))
            state = result.output
            stepper_state = result.stepper_state
            yield result

    @property
    def derive_func(self):
        """single_module.py:606-613, 895-897: the derived-variable function of the dataset's vertical coordinate."""
        from .derived_variables import AtmosphericDeriveFn
        return AtmosphericDeriveFn(self._step_obj._vertical_coordinate, self._step_obj._timestep)

    def predict(self, initial_condition: TensorMapping, forcing: TensorMapping,
                n_forward_steps: Optional[int] = None, compute_derived_variables: bool = False, labels=None,
                time=None, compute_derived_forcings: bool = True) -> Tuple[TensorDict, TensorDict]:
        """single_module.py:1169-1259 on plain dicts: initial_condition name -> (B, 1, H, W) prognostic state,
        forcing name -> (B, 1 + n_forward_steps, H, W).  Returns (output name -> (B, n_forward_steps, H, W),
        final prognostic state name -> (B, 1, H, W)).  The returned state is a ``PrognosticState`` (a dict) that carries
        the per-sample ``stepper_state`` (corrector dry-air reference) the way the reference's does
        (fme/ace/data_loading/batch_data.py:214-235): feed it back as the next window's initial condition.
        ``time``: the (B, 1 + n_forward_steps) ``TimeAxis`` of the forcing window (or ``forcing`` is a ``ForcingWindow`` carrying
        it) - needed only when the stepper derives forcings from it (single_module.py:1202-1203)."""
        if compute_derived_forcings:
            forcing = self.forcing_deriver(forcing, time)
        any_forcing = next(iter(forcing.values()))
        if n_forward_steps is None:
            n_forward_steps = any_forcing.shape[self.TIME_DIM] - self.n_ic_timesteps
        for k, v in initial_condition.items():
            if v.shape[self.TIME_DIM] != self.n_ic_timesteps:
                raise ValueError(f"Initial condition must have {self.n_ic_timesteps} timesteps, got {v.shape[1]}.")
        with torch.no_grad():
            outs = list(self.predict_generator(initial_condition, forcing, n_forward_steps, labels=labels,   # BatchLabels of a conditional module
                                               stepper_state=getattr(initial_condition, "stepper_state", None)))
        data = {k: torch.stack([o.output[k] for o in outs], dim=self.TIME_DIM) for k in outs[0].output}
        if compute_derived_variables:
            data = derive_over_window(self.derive_func, data, initial_condition, forcing, self.n_ic_timesteps, n_forward_steps)
        prognostic_state = PrognosticState({k: data[k][:, -1:] for k in self.prognostic_names})
        prognostic_state.stepper_state = outs[-1].stepper_state
        return data, prognostic_state

    def get_state(self):
        return {"step": self._step_obj.get_state()}

    def load_state(self, state):
        self._step_obj.load_state(state["step"])
