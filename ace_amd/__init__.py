"""ace_amd - MI355X-native SFNO rollout engine behind the fme.ace stepper / module-registry API.

Only the hot path named by BASELINE.json's north_star lives here: the SFNO forward
step (hand-written gfx950 HIP behind a C ABI, ace_amd/csrc + include/ace_sfno.h) and the
host-side mirror of the reference interfaces that call it (module registry, SFNO
builder, packer/normaliser, single-module step, stepper loop).  There is no CPU
fallback: without the built HIP library the ops raise.
"""

from .registry import Module, ModuleConfig, ModuleSelector, Registry  # noqa: F401
from .dataset_info import DatasetInfo  # noqa: F401
from .sht import InverseRealSHT, RealSHT  # noqa: F401
from .sfno import SphericalFourierNeuralOperatorBuilder, SphericalFourierNeuralOperatorNet  # noqa: F401
from .packer import Packer  # noqa: F401
from .normalizer import StandardNormalizer  # noqa: F401
from .step import SecondaryDecoderConfig, SingleModuleStep, SingleModuleStepConfig, StepArgs, StepOutput, StepperState  # noqa: F401
from .rand import RandomState, randn, randn_like, use_cpu_randn, use_generator  # noqa: F401
from .mlp import ColumnMLP, MLPConfig  # noqa: F401
from .checkpoint import LoadedStepper, StepperOverrideConfig, apply_stepper_override, load_stepper  # noqa: F401
from .csfno import NoiseConditionedSFNO, NoiseConditionedSFNOBuilder  # noqa: F401
from .healpix import CapturedHEALPixForward, HEALPixUNet, HEALPixUNetBuilder  # noqa: F401
from .corrector import AtmosphereCorrectorConfig  # noqa: F401
from .ocean import OceanConfig  # noqa: F401
from .derived_variables import AtmosphericDeriveFn, compute_derived_quantities  # noqa: F401
from .timeaxis import TimeAxis  # noqa: F401
from .insolation import InsolationConfig  # noqa: F401
from .derived_forcings import DerivedForcingsConfig, ForcingDeriver, ForcingWindow  # noqa: F401
from .multi_call import MultiCallConfig  # noqa: F401
from .stepper import PrognosticState, Stepper  # noqa: F401
from .inference import EnginePredict, ForcingWindows, InferenceData, Looper, TensorFileWriter, run_inference  # noqa: F401

__version__ = "0.1.0"
