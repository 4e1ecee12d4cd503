"""The reference's random-number contract for stochastic modules (third item).

fme/core/rand.py:39-104: every draw of a stochastic module goes through ``randn`` / ``randn_like``.  While a CPU
``torch.Generator`` is active (``use_generator``) the numbers are drawn from it ON THE CPU and moved to the requested device,
so a seeded rollout is reproducible across devices and independent of what else consumes the global RNG; with
``use_cpu_randn`` the global CPU RNG is used the same way; otherwise the draw happens directly on the device.
fme/core/random_state.py:23-92: ``RandomState`` wraps that one generator; it rides on ``StepperState.random_state``
(fme/core/stepper_state.py:22-34) and ``Stepper.step`` activates it around every network call
(fme/ace/stepper/single_module.py:1063-1068).  The generator advances in place, so the noise sequence does not depend on how
a rollout is cut into windows.

On this path the consumer is ``ace_amd.csfno.NoiseConditionedSFNO.draw_noise``; the draw stays OUTSIDE any captured hipGraph
(``ace_amd.rollout.RolloutEngine`` copies it into a static device buffer before each replay).
"""
import contextlib
import dataclasses
from typing import Dict, Optional, Set

import torch

USE_CPU_RANDN = False
_ACTIVE_GENERATOR: Optional[torch.Generator] = None


def active_generator() -> Optional[torch.Generator]:
    return _ACTIVE_GENERATOR


def randn(shape, **kwargs) -> torch.Tensor:
    """rand.py:55-63: same keyword handling - `device` is where the result ends up, not where it is drawn."""
    if _ACTIVE_GENERATOR is not None:
        device = kwargs.pop("device", None)
        result = torch.randn(shape, generator=_ACTIVE_GENERATOR, **kwargs)
        return result if device is None else result.to(device)
    if USE_CPU_RANDN:
        device = kwargs.pop("device", None)
        return torch.randn(shape, device="cpu", **kwargs).to(device)
    return torch.randn(shape, **kwargs)


def randn_like(x: torch.Tensor, **kwargs) -> torch.Tensor:
    """rand.py:39-52."""
    if _ACTIVE_GENERATOR is not None:
        device = kwargs.pop("device", x.device)
        dtype = kwargs.pop("dtype", x.dtype)
        return torch.randn(x.shape, generator=_ACTIVE_GENERATOR, dtype=dtype, **kwargs).to(device)
    if USE_CPU_RANDN:
        device = kwargs.pop("device", x.device)
        return torch.randn_like(x, device="cpu", **kwargs).to(device)
    return torch.randn_like(x, **kwargs)


@contextlib.contextmanager
def use_generator(generator: Optional[torch.Generator]):
    """rand.py:82-104: route randn / randn_like through `generator` (None: no-op); nested use restores the previous one."""
    global _ACTIVE_GENERATOR
    if generator is None:
        yield
        return
    old = _ACTIVE_GENERATOR
    _ACTIVE_GENERATOR = generator
    try:
        yield
    finally:
        _ACTIVE_GENERATOR = old


@contextlib.contextmanager
def use_cpu_randn():
    """rand.py:107-120 (restored on exceptions too)."""
    global USE_CPU_RANDN
    old = USE_CPU_RANDN
    USE_CPU_RANDN = True
    try:
        yield
    finally:
        USE_CPU_RANDN = old


def alternate_seed(seed: int) -> int:
    """rand.py:123-131."""
    g = torch.Generator()
    g.manual_seed(seed)
    return int(torch.randint(0, 2**31, (1,), generator=g).item())


@dataclasses.dataclass
class RandomState:
    """fme/core/random_state.py:23-92: one CPU generator for the whole batch, consumed in place; the device / ensemble
    transforms return the same advancing object."""

    generator: torch.Generator

    def __post_init__(self):
        if self.generator.device.type != "cpu":
            raise ValueError(f"RandomState requires a CPU torch.Generator, got device {self.generator.device}.")

    @classmethod
    def from_seed(cls, seed: int) -> "RandomState":
        generator = torch.Generator()
        generator.manual_seed(seed)
        return cls(generator=generator)

    def to_state_dict(self) -> Dict[str, torch.Tensor]:
        """the ADVANCED Mersenne-Twister state (a CPU uint8 tensor), not the seed: a restart continues the sequence"""
        return {"generator_state": self.generator.get_state()}

    @classmethod
    def from_state_dict(cls, state: Dict[str, torch.Tensor]) -> "RandomState":
        generator = torch.Generator()
        generator.set_state(state["generator_state"])
        return cls(generator=generator)

    @staticmethod
    def per_sample_state_keys() -> Set[str]:
        return set()

    def to_device(self) -> "RandomState":
        return self

    def to_cpu(self) -> "RandomState":
        return self

    def pin_memory(self) -> "RandomState":
        return self

    def broadcast_ensemble(self, n_ensemble: int) -> "RandomState":
        return self

    def sample_dim_size(self) -> Optional[int]:
        return None
