"""Where the random numbers of a stochastic module come from (third item).

The contract this restates (fme/core/rand.py:39-131, fme/core/random_state.py:23-92, fme/core/stepper_state.py:22-34,
fme/ace/stepper/single_module.py:1063-1068), in this module's own words:

* a draw has a SOURCE and a DESTINATION.  The destination is the ``device=`` the caller names (or the template tensor's); the
  source is decided by what is active: a CPU ``torch.Generator`` installed by ``use_generator`` (numbers come from it on the host
  and are then copied to the destination - a seeded rollout gives the same numbers on any device and is blind to other
  consumers of the global RNG), else the global CPU RNG under ``use_cpu_randn``, else the destination device's own RNG;
* ``RandomState`` carries that one generator through a rollout (on ``StepperState.random_state``); it advances in place, so the
  sequence does not depend on how the rollout is cut into windows, and its state dict is the ADVANCED generator state.

On this path the consumer is ``ace_amd.csfno.NoiseConditionedSFNO.draw_noise``; the draw stays OUTSIDE any captured hipGraph
(``ace_amd.rollout.RolloutEngine`` copies it into a static device buffer before each replay).
"""
import contextlib
from typing import Dict, Optional, Set

import torch


class _Source:
    """What is active right now.  One instance (`_src`); the context managers below swap single fields and put them back."""

    __slots__ = ("generator", "host_global")

    def __init__(self):
        self.generator: Optional[torch.Generator] = None   # set: draw from it, on the host
        self.host_global = False                           # set (and no generator): draw from the global CPU RNG


_src = _Source()


def __getattr__(name):   # `rand.USE_CPU_RANDN`: the reference exposes the flag as a module attribute; here it is a view of _src
    if name == "USE_CPU_RANDN":
        return _src.host_global
    raise AttributeError(name)


def active_generator() -> Optional[torch.Generator]:
    return _src.generator


def _draw(shape, dtype, destination, options, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The one place a normal deviate is made: host generator > global host RNG > the destination's RNG."""
    on_host = _src.generator is not None or _src.host_global
    if not on_host and like is not None:
        return torch.randn_like(like, dtype=dtype, device=destination, **options)   # keeps the template's memory format
    if not on_host:
        return torch.randn(shape, dtype=dtype, device=destination, **options)
    sample = torch.randn(shape, dtype=dtype, generator=_src.generator, device="cpu", **options)
    return sample if destination is None else sample.to(destination)


def randn(shape, **kwargs) -> torch.Tensor:
    """``torch.randn(shape, **kwargs)`` under the contract above (rand.py:55-63): `device` names where the result lives."""
    destination = kwargs.pop("device", None)
    return _draw(shape, kwargs.pop("dtype", None), destination, kwargs)


def randn_like(x: torch.Tensor, **kwargs) -> torch.Tensor:
    """``torch.randn_like(x, **kwargs)`` under the contract above (rand.py:39-52): shape, dtype and device default to x's."""
    destination = kwargs.pop("device", x.device)
    return _draw(x.shape, kwargs.pop("dtype", x.dtype), destination, kwargs, like=x)


@contextlib.contextmanager
def _swapped(field: str, value):
    previous = getattr(_src, field)
    setattr(_src, field, value)
    try:
        yield
    finally:
        setattr(_src, field, previous)


def use_generator(generator: Optional[torch.Generator]):
    """Context: draws come from `generator` (rand.py:82-104).  ``None`` leaves whatever is active in place; nesting restores."""
    return contextlib.nullcontext() if generator is None else _swapped("generator", generator)


def use_cpu_randn():
    """Context: draws without a generator come from the global CPU RNG (rand.py:107-120); restored on exceptions as well."""
    return _swapped("host_global", True)


def alternate_seed(seed: int) -> int:
    """A second seed derived from `seed`: the first 31-bit integer of a generator seeded with it (rand.py:123-131)."""
    return int(torch.randint(0, 2**31, (1,), generator=torch.Generator().manual_seed(seed)).item())


def _host_generator(seed: Optional[int] = None, state: Optional[torch.Tensor] = None) -> torch.Generator:
    g = torch.Generator()
    if seed is not None:
        g.manual_seed(seed)
    if state is not None:
        g.set_state(state)
    return g


class RandomState:
    """The rollout's one host generator (random_state.py:23-92).  A stepper-state component like the corrector's, but with no
    per-sample part: every placement / broadcast hook hands back the same, still advancing, object."""

    __slots__ = ("generator",)
    _STATE_KEY = "generator_state"

    def __init__(self, generator: torch.Generator):
        where = generator.device.type
        if where != "cpu":
            raise ValueError(f"RandomState requires a CPU torch.Generator, got device {generator.device}.")
        self.generator = generator

    def __repr__(self):
        return f"RandomState(generator=<{self.generator.device} generator, seed {self.generator.initial_seed()}>)"

    def __eq__(self, other):
        return isinstance(other, RandomState) and other.generator is self.generator

    __hash__ = None

    @classmethod
    def from_seed(cls, seed: int) -> "RandomState":
        return cls(_host_generator(seed=seed))

    @classmethod
    def from_state_dict(cls, state: Dict[str, torch.Tensor]) -> "RandomState":
        return cls(_host_generator(state=state[cls._STATE_KEY]))

    def to_state_dict(self) -> Dict[str, torch.Tensor]:
        # the Mersenne-Twister state as it is NOW (a CPU uint8 tensor), not the seed: a restart continues the sequence
        return {self._STATE_KEY: self.generator.get_state()}

    @staticmethod
    def per_sample_state_keys() -> Set[str]:
        return set()

    def sample_dim_size(self) -> Optional[int]:
        return None

    def _same(self, *_args, **_kwargs) -> "RandomState":
        return self

    to_device = to_cpu = pin_memory = broadcast_ensemble = _same
