"""Import the REAL reference hot-path modules under stubs (build container only).

Test infrastructure.  ``/root/reference`` exists only in the build container,
so this module is used exclusively by the fixture generators
``tests/golden/make_golden.py`` / ``tests/golden/make_golden_headline.py`` (to emit
golden vectors from the reference itself); the tests read the committed fixtures.
Nothing here travels to the GPU box in a usable form and nothing in the
product path imports it.

Recipe: ``fme`` cannot be imported normally here
(python 3.10 < 3.11, torch_harmonics / dacite / tensorly / tltorch / xarray
missing), so we (1) alias ``typing.Self``; (2) register empty namespace
packages whose ``__path__`` points into the reference tree so no package
``__init__`` runs; (3) stub the missing third-party modules - the
torch-harmonics quadrature/Legendre entry points are served by this oracle's
restatement, which the reference's own goldens then validate; (4) import the
real ``fme.sht_fix`` and ``fme.ace.models.modulus.sfnonet``.
"""

import importlib
import os
import sys
import types
import typing

REF = os.environ.get("ACE_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "fme"))


def _ns(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace with the reference classes (RealSHT, InverseRealSHT, SFNO)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present")
    import numpy as np
    import torch
    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self

    from . import legendre as _leg
    from . import quadrature as _quad

    # (2) skeleton packages
    for pkg in [
        "fme", "fme.core", "fme.core.models", "fme.core.distributed", "fme.core.benchmark",
        "fme.ace", "fme.ace.models", "fme.ace.models.modulus",
    ]:
        _ns(pkg, os.path.join(REF, *pkg.split(".")))

    # (3) third-party stubs
    th = _ns("torch_harmonics", "/nonexistent")
    thq = _ns("torch_harmonics.quadrature")
    thl = _ns("torch_harmonics.legendre")
    thd = _ns("torch_harmonics.distributed")

    def _as_t(f):
        def g(n, a=-1.0, b=1.0):
            x, w = f(n, a, b)
            return torch.as_tensor(np.ascontiguousarray(x)), torch.as_tensor(np.ascontiguousarray(w))
        return g

    thq.legendre_gauss_weights = _as_t(_quad.legendre_gauss_weights)
    thq.lobatto_weights = _as_t(_quad.lobatto_weights)
    thq.clenshaw_curtiss_weights = _as_t(_quad.clenshaw_curtiss_weights)

    def _precompute_legpoly(mmax, lmax, t, norm="ortho", inverse=False, csphase=True):
        # a copy: the oracle's table is a shared read-only array, the reference wraps what it gets in a tensor
        return np.array(_leg.precompute_legpoly(mmax, lmax, np.asarray(t), norm=norm, inverse=inverse, csphase=csphase))

    thl._precompute_legpoly = _precompute_legpoly

    class _Never(torch.nn.Module):
        pass

    thd.DistributedRealSHT = _Never
    thd.DistributedInverseRealSHT = type("DistributedInverseRealSHT", (_Never,), {})
    th.quadrature, th.legendre, th.distributed = thq, thl, thd
    th.RealFFT2 = type("RealFFT2", (_Never,), {})
    th.InverseRealFFT2 = type("InverseRealFFT2", (_Never,), {})

    tl = _ns("tensorly")
    tl.set_backend = lambda *_a, **_k: None
    tl.ndim = lambda t: t.ndim
    tl.einsum = torch.einsum
    _ns("tltorch", "/nonexistent")
    _ns("tltorch.factorized_tensors", "/nonexistent")
    core = _ns("tltorch.factorized_tensors.core")
    core.FactorizedTensor = type("FactorizedTensor", (), {})

    bench = _ns("fme.core.benchmark.benchmark")

    class BenchmarkABC:
        pass

    bench.BenchmarkABC = BenchmarkABC
    bench.register_benchmark = lambda name: (lambda cls: cls)

    timer = _ns("fme.core.benchmark.timer")
    timer.Timer = type("Timer", (), {})

    class NullTimer:      # fme/core/benchmark/timer.py: a no-op timer whose children are no-op timers
        def child(self, name):
            return self

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    timer.NullTimer = NullTimer
    timer.CUDATimer = type("CUDATimer", (), {})

    dev = _ns("fme.core.device")
    dev.get_device = lambda: torch.device("cpu")
    dev.using_gpu = lambda: False

    ty = _ns("fme.core.typing_")
    ty.TensorDict = dict
    ty.TensorMapping = typing.Mapping

    # (4) the real reference modules
    sht_fix = importlib.import_module("fme.sht_fix")
    sfnonet = importlib.import_module("fme.ace.models.modulus.sfnonet")

    ns = types.SimpleNamespace(
        RealSHT=sht_fix.RealSHT,
        InverseRealSHT=sht_fix.InverseRealSHT,
        SFNO=sfnonet.SphericalFourierNeuralOperatorNet,
        SpectralConvS2=importlib.import_module("fme.ace.models.modulus.s2convolutions").SpectralConvS2,
    )
    _loaded = ns
    return ns


_corrector = None


def load_corrector():
    """The REAL reference post-step corrector stack under stubs (build container only): returns a namespace with
    ``AtmosphereCorrectorConfig``, ``HybridSigmaPressureCoordinate``, ``LatLonCoordinates``, ``CorrectorState``.
    Extra stubs on top of ``load()``: ``dacite`` (plain ``data_class(**data)``), ``fme.core.device`` helpers, a
    non-distributed ``fme.core.distributed.Distributed`` (full slices, plain weighted mean), ``fme.core.dataset_info``
    and ``fme.core.registry.corrector`` (only imported for their names); ``fme.core.typing_`` is the real module."""
    global _corrector
    if _corrector is not None:
        return _corrector
    load()
    import torch

    fme = sys.modules["fme"]
    fme.get_device = lambda: torch.device("cpu")
    for pkg in ["fme.core.corrector", "fme.core.registry"]:
        if pkg not in sys.modules:
            _ns(pkg, os.path.join(REF, *pkg.split(".")))
    di = _ns("fme.core.dataset_info")
    di.DatasetInfo = type("DatasetInfo", (), {})
    rc = _ns("fme.core.registry.corrector")

    class CorrectorSelector:
        @classmethod
        def register(cls, name):
            return lambda c: c

    rc.CorrectorSelector = CorrectorSelector
    d = _ns("dacite")
    d.Config = type("Config", (), {"__init__": lambda self, **kw: None})
    d.from_dict = lambda data_class, data, config=None: data_class(**data)
    ex = _ns("dacite.exceptions")
    ex.DaciteError = type("DaciteError", (Exception,), {})
    d.exceptions = ex
    dv = _ns("fme.core.device")
    dv.get_device = lambda: torch.device("cpu")
    dv.using_gpu = lambda: False
    dv.in_dataloader_worker = lambda: False
    dv.using_srun = lambda: False
    dv.move_tensordict_to_device = lambda x: x

    class _Dist:
        @classmethod
        def get_instance(cls):
            return cls()

        def get_local_slices(self, shape, *a, **k):
            return tuple(slice(None) for _ in shape)

        def weighted_mean(self, data, weights, dim, keepdim=False):
            return importlib.import_module("fme.core.metrics").weighted_mean(data, weights, dim=dim, keepdim=keepdim)

        def spatial_reduce_sum(self, x):
            return x

        def zonal_mean(self, data):
            return data.nanmean(dim=-1)

    sys.modules["fme.core.distributed"].Distributed = _Dist
    if "fme.core.typing_" in sys.modules:
        del sys.modules["fme.core.typing_"]
    importlib.import_module("fme.core.typing_")
    coords = importlib.import_module("fme.core.coordinates")
    atm = importlib.import_module("fme.core.corrector.atmosphere")
    state = importlib.import_module("fme.core.corrector.state")
    _corrector = types.SimpleNamespace(
        AtmosphereCorrectorConfig=atm.AtmosphereCorrectorConfig, EnergyBudgetConfig=atm.EnergyBudgetConfig,
        HybridSigmaPressureCoordinate=coords.HybridSigmaPressureCoordinate, LatLonCoordinates=coords.LatLonCoordinates,
        CorrectorState=state.CorrectorState, module=atm)
    return _corrector


_csfno = None


def load_csfno():
    """The REAL NoiseConditionedSFNO (fme/ace/registry/stochastic_sfno.py + fme/core/models/conditional_sfno) under the
    stubs of ``load_corrector`` plus SHT factories on the stub ``Distributed`` (the reference's own
    ``fme.sht_fix.RealSHT/InverseRealSHT``).  Returns a namespace with ``Builder`` and ``DatasetInfo``-like helper."""
    global _csfno
    if _csfno is not None:
        return _csfno
    base = load()
    load_corrector()
    D = sys.modules["fme.core.distributed"].Distributed
    D.get_sht = lambda self, nlat, nlon, lmax=None, mmax=None, grid="equiangular": base.RealSHT(
        nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
    D.get_isht = lambda self, nlat, nlon, lmax=None, mmax=None, grid="equiangular": base.InverseRealSHT(
        nlat, nlon, lmax=lmax, mmax=mmax, grid=grid)
    D.set_seed = lambda self, seed: None
    for pkg in ["fme.core.models.conditional_sfno", "fme.ace.registry"]:
        if pkg not in sys.modules:
            _ns(pkg, os.path.join(REF, *pkg.split(".")))
    reg = importlib.import_module("fme.ace.registry.stochastic_sfno")

    class Info:
        def __init__(self, img_shape):
            self.img_shape = img_shape
            self.all_labels = set()

    _csfno = types.SimpleNamespace(Builder=reg.NoiseConditionedSFNOBuilder, Info=Info, module=reg)
    return _csfno


_hpx = None


def load_healpix():
    """The REAL HEALPix UNet (fme/ace/models/healpix/*, pure torch; its optional earth2grid padding backend is absent here, the
    'karlbauer' backend - documented there as giving the same result - is used).  Returns a namespace with the reference's
    configuration dataclasses, HEALPixUNet and HEALPixPadding."""
    global _hpx
    if _hpx is not None:
        return _hpx
    load()
    for pkg in ["fme.ace.models.healpix"]:
        if pkg not in sys.modules:
            _ns(pkg, os.path.join(REF, *pkg.split(".")))
    blocks = importlib.import_module("fme.ace.models.healpix.healpix_blocks")
    enc = importlib.import_module("fme.ace.models.healpix.healpix_encoder")
    dec = importlib.import_module("fme.ace.models.healpix.healpix_decoder")
    unet = importlib.import_module("fme.ace.models.healpix.healpix_unet")
    act = importlib.import_module("fme.ace.models.healpix.healpix_activations")
    pads = importlib.import_module("fme.ace.models.healpix.healpix_paddings")
    _hpx = types.SimpleNamespace(blocks=blocks, encoder=enc, decoder=dec, unet=unet, activations=act, paddings=pads)
    return _hpx


_stepper = None


def load_stepper_ref():
    """The REAL reference stepper stack (fme/ace/stepper/single_module.py ``Stepper`` / ``StepperConfig``,
    fme/core/step/{step,single_module,multi_call}.py, the real corrector / module registries, the real
    ``fme.core.dataset_info.DatasetInfo``) under stubs - build container only.  On top of ``load_csfno()``:
    ``dacite`` becomes oracle/_minidacite.py, the IO / logging third-party packages the stepper module imports for
    type names only (xarray, zarr, cftime, netCDF4, wandb, dask, h5netcdf) become permissive empty modules, the stub
    ``Distributed`` gains the non-distributed ``wrap_module`` (a holder whose state-dict keys carry the ``module.``
    prefix, fme/core/distributed/non_distributed.py:15-28), and every module that registers itself with a registry is
    re-imported against the real ``CorrectorSelector`` / ``ModuleSelector``.  Used by tests/golden/make_golden_checkpoint.py
    to emit a checkpoint state and a short rollout from the reference itself."""
    global _stepper
    if _stepper is not None:
        return _stepper
    load_csfno()
    import torch

    from . import _minidacite

    d = sys.modules["dacite"]
    ex = sys.modules["dacite.exceptions"]
    d.from_dict, d.Config = _minidacite.from_dict, _minidacite.Config
    for n in ["DaciteError", "UnexpectedDataError", "WrongTypeError", "MissingValueError", "UnionMatchError"]:
        setattr(ex, n, getattr(_minidacite, n))
        setattr(d, n, getattr(_minidacite, n))
    for pkg in ["fme.core.step", "fme.core.generics", "fme.core.dataset", "fme.ace.stepper"]:
        if pkg not in sys.modules:
            _ns(pkg, os.path.join(REF, *pkg.split(".")))

    class _Permissive(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            t = type(name, (), {"__init__": lambda self, *a, **k: None})
            setattr(self, name, t)
            return t

    for extra in ["xarray", "zarr", "cftime", "netCDF4", "wandb", "dask", "h5netcdf", "xarray.coding",
                  "xarray.coding.times", "dask.array"]:
        if extra not in sys.modules:
            m = _Permissive(extra)
            m.__path__ = []
            sys.modules[extra] = m
    for k in [k for k in sys.modules
              if k.startswith(("fme.core.registry.", "fme.core.corrector.", "fme.ace.registry."))
              or k in ("fme.core.dataset_info", "fme.core.ocean")]:
        del sys.modules[k]
    R = sys.modules["fme.core.registry"]
    R.CorrectorSelector = importlib.import_module("fme.core.registry.corrector").CorrectorSelector
    R.ModuleSelector = importlib.import_module("fme.core.registry.module").ModuleSelector

    class _Holder(torch.nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, *a, **k):
            return self.module(*a, **k)

    sys.modules["fme.core.distributed"].Distributed.wrap_module = lambda self, m: _Holder(m)
    di = importlib.import_module("fme.core.dataset_info")
    importlib.import_module("fme.core.corrector.atmosphere")
    importlib.import_module("fme.ace.registry.sfno")
    importlib.import_module("fme.ace.registry.stochastic_sfno")
    sm = importlib.import_module("fme.ace.stepper.single_module")
    coords = importlib.import_module("fme.core.coordinates")
    opt = importlib.import_module("fme.core.optimization")
    _stepper = types.SimpleNamespace(
        StepperConfig=sm.StepperConfig, Stepper=sm.Stepper, DatasetInfo=di.DatasetInfo,
        LatLonCoordinates=coords.LatLonCoordinates, HybridSigmaPressureCoordinate=coords.HybridSigmaPressureCoordinate,
        NullOptimization=opt.NullOptimization, module=sm)
    return _stepper


_insolation = None


def load_insolation():
    """The REAL insolation (fme/ace/stepper/insolation/cm4.py) under the stubs of ``load_corrector`` - build container only.
    cftime is not in this image; the reference only needs its datetimes to (a) subtract (-> timedelta), (b) carry a
    ``calendar`` attribute and (c) be constructible from components, so the stand-in is the STANDARD LIBRARY's
    ``datetime.datetime`` (a proleptic Gregorian calendar; identical to cftime's 'standard' for every date after 1582-10-15)
    subclassed to carry the attribute.  No calendar arithmetic is written here.  xarray's ``DataArray`` is a holder with
    ``to_numpy()``.  Returns a namespace with ``cm4`` (the real module), ``Datetime`` (the stand-in class, ``calendar`` set per
    subclass), ``TimeArray`` and the real ``LatLonCoordinates``."""
    global _insolation
    if _insolation is not None:
        return _insolation
    load_corrector()
    import datetime

    import numpy as np

    def make(calendar):
        return type("Datetime_" + calendar, (datetime.datetime,), {"calendar": calendar})

    std, pg = make("standard"), make("proleptic_gregorian")
    cf = _ns("cftime")
    cf.DatetimeGregorian, cf.DatetimeProlepticGregorian = std, pg
    for other in ["DatetimeNoLeap", "DatetimeJulian", "Datetime360Day", "DatetimeAllLeap"]:
        setattr(cf, other, type(other, (), {}))          # names the module's table needs at import; never instantiated here
    if "xarray" not in sys.modules:
        _ns("xarray")
    xr = sys.modules["xarray"]

    class TimeArray:
        def __init__(self, values):
            self._v = np.asarray(values, dtype=object)

        def to_numpy(self):
            return self._v

    if not hasattr(xr, "DataArray"):
        xr.DataArray = TimeArray
    for pkg in ["fme.ace.stepper", "fme.ace.stepper.insolation"]:
        if pkg not in sys.modules:
            _ns(pkg, os.path.join(REF, *pkg.split(".")))
    cm4 = importlib.import_module("fme.ace.stepper.insolation.cm4")
    coords = importlib.import_module("fme.core.coordinates")
    _insolation = types.SimpleNamespace(cm4=cm4, standard=std, proleptic_gregorian=pg, TimeArray=TimeArray,
                                        LatLonCoordinates=coords.LatLonCoordinates)
    return _insolation
