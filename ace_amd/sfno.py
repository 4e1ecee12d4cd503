"""SphericalFourierNeuralOperatorNet + its registry builder, MI355X-native.

Mirrors the reference's module tree so that `state_dict()` carries exactly the
reference's parameter names and shapes and reference
checkpoints load with a strict `load_state_dict`:

* builder / registration        fme/ace/registry/sfno.py:14-61
* network constructor / forward fme/ace/models/modulus/sfnonet.py:341-685, 713-749
* block                         fme/ace/models/modulus/sfnonet.py:123-252
* spectral filter               fme/ace/models/modulus/s2convolutions.py:55-197
* MLP                           fme/ace/models/modulus/layers.py:97-137
* weight init                   fme/ace/models/modulus/initialization.py:23-78

The torch sub-modules below only HOLD parameters (and consume the RNG in the
reference's order at construction); `forward` is one call into the HIP library.
"""

import ctypes
import dataclasses
import math
import os
from typing import Any, Literal, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .registry import ModuleConfig, ModuleSelector

_OPERATOR = {"diagonal": 0, "dhconv": 1}
_NORM = {"none": 0, "instance_norm": 1, "layer_norm": 3}   # (2: the conditional layer norm of csfno.py)
_ACT = {"gelu": 1, "relu": 2, "silu": 3}
_ACT_LAYER = {"gelu": nn.GELU, "relu": nn.ReLU, "silu": nn.SiLU}
_GRID = {"legendre-gauss": 0, "lobatto": 1, "equiangular": 2}
_PRECISION = {"fp32": 0, "f16x3": 1}
# "f16x3": contractions with K >= 32 on error-compensated fp16 MFMA (fp32-class accuracy: measured error against an fp64
# oracle is BELOW the fp32 CPU reference's own); "fp32": every contraction on exact-fp32 MFMA (bitwise k-ordered fma chains)
DEFAULT_PRECISION = "f16x3"


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """initialization.py:23-78: inverse-CDF truncated normal (uniform_ -> erfinv_ -> scale -> clamp)."""

    def norm_cdf(x):
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0

    with torch.no_grad():
        lo = norm_cdf((a - mean) / std)
        up = norm_cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * up - 1)
        tensor.erfinv_()
        tensor.mul_(std * math.sqrt(2.0))
        tensor.add_(mean)
        tensor.clamp_(min=a, max=b)
        return tensor


class SpectralConvS2(nn.Module):
    """Parameter holder of the spectral filter (s2convolutions.py:55-160): weight (Cin, Cout, L[, M], 2)
    = scale * randn, bias (1, Cout, 1, 1) zeros."""

    def __init__(self, in_channels, out_channels, modes_lat, modes_lon, operator_type):
        super().__init__()
        scale = 1 / (in_channels * out_channels)
        shape = [in_channels, out_channels, modes_lat]
        if operator_type == "diagonal":
            shape += [modes_lon]
        elif operator_type != "dhconv":
            raise ValueError(f"Unsupported operator type f{operator_type}")
        self.operator_type = operator_type
        self.weight = nn.Parameter(scale * torch.randn(*shape, 2))
        self.bias = nn.Parameter(scale * torch.zeros(1, out_channels, 1, 1))


class SpectralFilterLayer(nn.Module):
    """sfnonet.py:45-120 (filter_type 'linear' only)."""

    def __init__(self, embed_dim, modes_lat, modes_lon, operator_type):
        super().__init__()
        self.filter = SpectralConvS2(embed_dim, embed_dim, modes_lat, modes_lon, operator_type)


class MLP(nn.Module):
    """layers.py:97-137: fwd = Sequential(Conv1x1(C, hidden), act, Conv1x1(hidden, C))."""

    def __init__(self, in_features, hidden_features, act_layer):
        super().__init__()
        fc1 = nn.Conv2d(in_features, hidden_features, 1, bias=True)
        act = act_layer()
        fc2 = nn.Conv2d(hidden_features, in_features, 1, bias=True)
        self.fwd = nn.Sequential(fc1, act, fc2)


class FourierNeuralOperatorBlock(nn.Module):
    """sfnonet.py:123-215 (inner_skip='linear', outer_skip='identity' as the net builds them, :625-626)."""

    def __init__(self, embed_dim, modes_lat, modes_lon, operator_type, mlp_ratio, act_layer, norm_layer, use_mlp):
        super().__init__()
        self.norm0 = norm_layer[0]()
        self.filter = SpectralFilterLayer(embed_dim, modes_lat, modes_lon, operator_type)
        self.inner_skip = nn.Conv2d(embed_dim, embed_dim, 1, 1)
        self.act_layer = act_layer()
        self.norm1 = norm_layer[1]()
        if use_mlp:
            self.mlp = MLP(embed_dim, int(embed_dim * mlp_ratio), act_layer)
        self.outer_skip = nn.Identity()


class SphericalFourierNeuralOperatorNet(nn.Module):
    """Drop-in for fme.ace.models.modulus.sfnonet.SphericalFourierNeuralOperatorNet (inference forward)."""

    def __init__(
        self,
        params,
        spectral_transform: str = "sht",
        filter_type: str = "linear",
        operator_type: str = "diagonal",
        img_shape: Tuple[int, int] = (721, 1440),
        scale_factor: int = 1,
        residual_filter_factor: int = 1,
        in_chans: int = 2,
        out_chans: int = 2,
        embed_dim: int = 256,
        num_layers: int = 12,
        use_mlp: int = True,
        mlp_ratio: float = 2.0,
        activation_function: str = "gelu",
        encoder_layers: int = 1,
        pos_embed: bool = True,
        drop_rate: float = 0.0,
        drop_path_rate: float = 0.0,
        num_blocks: int = 16,
        sparsity_threshold: float = 0.0,
        normalization_layer: str = "instance_norm",
        hard_thresholding_fraction: float = 1.0,
        use_complex_kernels: bool = True,
        big_skip: bool = True,
        rank: float = 1.0,
        factorization: Any = None,
        separable: bool = False,
        complex_network: bool = True,
        complex_activation: str = "real",
        spectral_layers: int = 3,
        checkpointing: int = 0,
    ):
        super().__init__()

        def pick(name, default):  # sfnonet.py:380-462: `params.X if hasattr(params, "X") else X`
            return getattr(params, name) if hasattr(params, name) else default

        self.params = params
        self.spectral_transform = pick("spectral_transform", spectral_transform)
        self.filter_type = pick("filter_type", filter_type)
        self.operator_type = pick("operator_type", operator_type)
        if hasattr(params, "img_shape_x") and hasattr(params, "img_shape_y"):
            img_shape = (params.img_shape_x, params.img_shape_y)
        self.img_shape = (int(img_shape[0]), int(img_shape[1]))
        self.scale_factor = pick("scale_factor", scale_factor)
        self.residual_filter_factor = pick("residual_filter_factor", residual_filter_factor)
        self.in_chans = pick("N_in_channels", in_chans)
        self.out_chans = pick("N_out_channels", out_chans)
        self.embed_dim = self.num_features = pick("embed_dim", embed_dim)
        self.num_layers = pick("num_layers", num_layers)
        self.hard_thresholding_fraction = pick("hard_thresholding_fraction", hard_thresholding_fraction)
        self.normalization_layer = pick("normalization_layer", normalization_layer)
        self.use_mlp = bool(pick("use_mlp", use_mlp))
        self.mlp_ratio = mlp_ratio
        self.activation_function = pick("activation_function", activation_function)
        self.encoder_layers = pick("encoder_layers", encoder_layers)
        has_pos_embed = bool(pick("pos_embed", pos_embed))
        self.big_skip = bool(pick("big_skip", big_skip))
        self.factorization = pick("factorization", factorization)
        self.separable = pick("separable", separable)
        self.data_grid = pick("data_grid", "equiangular")

        # what the HIP path implements; anything else is rejected here, loudly
        if self.spectral_transform != "sht":
            raise NotImplementedError("only spectral_transform='sht' is implemented")
        if self.filter_type != "linear":
            raise NotImplementedError("only filter_type='linear' is implemented")
        if self.operator_type not in _OPERATOR:
            raise ValueError(f"Unsupported operator type f{self.operator_type}")
        if self.factorization is not None or self.separable:
            raise NotImplementedError("factorized / separable spectral weights are not implemented")
        if self.residual_filter_factor < 1 or self.scale_factor < 1:
            raise ValueError("scale_factor and residual_filter_factor must be >= 1")
        if self.normalization_layer not in _NORM:
            raise NotImplementedError(f"Error, normalization {self.normalization_layer} not implemented.")
        if self.activation_function not in _ACT:
            raise ValueError(f"Unknown activation function {self.activation_function}")
        if self.data_grid not in _GRID:
            raise NotImplementedError(f"data_grid {self.data_grid!r} is not implemented")
        if drop_rate > 0.0 or drop_path_rate > 0.0:
            raise NotImplementedError("dropout is a training feature; this is the inference path")

        self.h = int(self.img_shape[0] // self.scale_factor)
        self.w = int(self.img_shape[1] // self.scale_factor)
        modes_lat = int(self.h * self.hard_thresholding_fraction)
        modes_lon = int((self.w // 2 + 1) * self.hard_thresholding_fraction)
        self.modes_lat, self.modes_lon = modes_lat, modes_lon
        act_layer = _ACT_LAYER[self.activation_function]

        # encoder (sfnonet.py:566-577)
        current_dim = self.in_chans
        encoder_modules = []
        for _ in range(self.encoder_layers):
            encoder_modules.append(nn.Conv2d(current_dim, self.embed_dim, 1, bias=True))
            encoder_modules.append(act_layer())
            current_dim = self.embed_dim
        encoder_modules.append(nn.Conv2d(current_dim, self.embed_dim, 1, bias=False))
        self.encoder = nn.Sequential(*encoder_modules)

        if self.normalization_layer == "instance_norm":  # sfnonet.py:593-601
            def norm_layer():
                return nn.InstanceNorm2d(num_features=self.embed_dim, eps=1e-6, affine=True,
                                         track_running_stats=False)
            norm_layer0 = norm_layer1 = norm_layer
        elif self.normalization_layer == "layer_norm":   # sfnonet.py:584-592: over a whole grid, an affine of that shape shared by the channels
            def norm_layer0():      # on the data grid
                return nn.LayerNorm(normalized_shape=(self.img_shape[0], self.img_shape[1]), eps=1e-6)

            def norm_layer1():      # on the inner (img_shape // scale_factor) grid
                return nn.LayerNorm(normalized_shape=(self.h, self.w), eps=1e-6)
        else:
            norm_layer0 = norm_layer1 = nn.Identity

        def block_norms(i):         # sfnonet.py:616-623: a first block sees (data, inner), a last one (inner, data), the others (inner, inner)
            if i == 0:
                return norm_layer0, norm_layer1
            if i == self.num_layers - 1:
                return norm_layer1, norm_layer0
            return norm_layer1, norm_layer1

        self.blocks = nn.ModuleList(
            [
                FourierNeuralOperatorBlock(self.embed_dim, modes_lat, modes_lon, self.operator_type, mlp_ratio,
                                           act_layer, block_norms(i), self.use_mlp)
                for i in range(self.num_layers)
            ]
        )

        # decoder (sfnonet.py:660-671)
        current_dim = self.embed_dim + self.big_skip * self.in_chans
        decoder_modules = []
        for _ in range(self.encoder_layers):
            decoder_modules.append(nn.Conv2d(current_dim, self.embed_dim, 1, bias=True))
            decoder_modules.append(act_layer())
            current_dim = self.embed_dim
        decoder_modules.append(nn.Conv2d(current_dim, self.out_chans, 1, bias=False))
        self.decoder = nn.Sequential(*decoder_modules)

        if has_pos_embed:  # sfnonet.py:674-683
            self.pos_embed = nn.Parameter(torch.zeros(1, self.embed_dim, self.img_shape[0], self.img_shape[1]))
            trunc_normal_(self.pos_embed, std=0.02)

        self.apply(self._init_weights)

        # arithmetic of the 1x1 convolutions: "fp32" (exact fp32 MFMA, the reference's arithmetic) or "f16x3"
        # (compensated fp16 MFMA, fp32-class accuracy, ~5x the MFMA rate); not part of the reference API
        self.precision = os.environ.get("ACE_SFNO_PRECISION", DEFAULT_PRECISION)
        if self.precision not in _PRECISION:
            raise ValueError(f"ACE_SFNO_PRECISION must be one of {list(_PRECISION)}")
        self._native = None          # ace_sfno* handle
        self._native_key = None      # (device index, max_batch)
        self._uploaded = {}          # name -> (data_ptr, version) at last upload

    def _init_weights(self, m):
        """sfnonet.py:687-697."""
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ native plumbing
    def _config_struct(self, max_batch: int) -> _lib.AceSfnoConfig:
        return _lib.AceSfnoConfig(
            in_chans=self.in_chans, out_chans=self.out_chans, nlat=self.img_shape[0], nlon=self.img_shape[1],
            embed_dim=self.embed_dim, num_layers=self.num_layers, scale_factor=self.scale_factor,
            hard_thresholding_fraction=float(self.hard_thresholding_fraction),
            operator_type=_OPERATOR[self.operator_type], normalization_layer=_NORM[self.normalization_layer],
            activation_function=_ACT[self.activation_function], use_mlp=int(self.use_mlp),
            mlp_ratio=float(self.mlp_ratio), encoder_layers=self.encoder_layers,
            pos_embed=int(hasattr(self, "pos_embed")), big_skip=int(self.big_skip),
            data_grid=_GRID[self.data_grid], max_batch=max_batch, precision=_PRECISION[self.precision],
            residual_filter_factor=int(self.residual_filter_factor),
        )

    def set_precision(self, precision: str):
        """Switch the conv arithmetic ("fp32" | "f16x3"); rebuilds the native handle on next use."""
        if precision not in _PRECISION:
            raise ValueError(f"precision must be one of {list(_PRECISION)}")
        if precision != self.precision:
            self.precision = precision
            self._release_native()
        return self

    def _release_native(self):
        if self._native is not None:
            try:
                self.__dict__.get("_native_destroy", _lib.lib().ace_sfno_destroy)(self._native)
            except Exception:
                pass
        self._native, self._native_key, self._uploaded = None, None, {}

    def __del__(self):
        try:
            native = self.__dict__.get("_native")
            if native is not None:
                self.__dict__.get("_native_destroy", _lib.lib().ace_sfno_destroy)(native)
                self.__dict__["_native"] = None
        except Exception:  # interpreter shutdown
            pass

    def _ensure_native(self, device: torch.device, batch: int):
        key = (device.index, self._native_key[1] if self._native_key else 0)
        if self._native is None or self._native_key[0] != device.index or batch > self._native_key[1]:
            self._release_native()
            handle = ctypes.c_void_p()
            cfg = self._config_struct(max_batch=batch)
            with torch.cuda.device(device):
                _lib.check(_lib.lib().ace_sfno_create(ctypes.byref(cfg), ctypes.byref(handle)))
            self._native, self._native_key = handle, (device.index, batch)
            self.__dict__["_native_destroy"] = _lib.lib().ace_sfno_destroy      # freed by the library that made it
        del key

    def sync_weights(self, force: bool = False):
        """Upload parameters that changed since the last upload (load_state_dict, optimiser step, .to())."""
        L = _lib.lib()
        stream = _lib.current_stream()
        changed = 0
        for name, p in self.state_dict(keep_vars=True).items():
            stamp = (p.data_ptr(), p._version)
            if not force and self._uploaded.get(name) == stamp:
                continue
            t = p.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            _lib.check(L.ace_sfno_set_weight(self._native, name.encode(), _lib.ptr(t), t.numel(), stream))
            self._uploaded[name] = stamp
            changed += 1
        return changed

    def _prepare(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != self.in_chans or tuple(x.shape[-2:]) != self.img_shape:
            raise AssertionError(
                f"expected input (B, {self.in_chans}, {self.img_shape[0]}, {self.img_shape[1]}), got {tuple(x.shape)}"
            )
        if not x.is_cuda:
            raise RuntimeError("SphericalFourierNeuralOperatorNet (ace_amd) runs on an MI355X only: move the module "
                               "and its input to 'cuda'. There is no CPU fallback.")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and x.requires_grad:
            raise RuntimeError("ace_amd implements the inference forward only; call under torch.no_grad()")
        x = x.float().contiguous()
        self._ensure_native(x.device, x.shape[0])
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self._prepare(x)
        self.sync_weights()
        out = torch.empty(x.shape[0], self.out_chans, *self.img_shape, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ace_sfno_forward(self._native, _lib.ptr(x), _lib.ptr(out), x.shape[0],
                                               _lib.current_stream()))
        return out

    def forward_graph(self, x: torch.Tensor, out: torch.Tensor, sync: bool = True) -> torch.Tensor:
        """hipGraph replay of forward on STATIC buffers (same x/out storage every call).  ``sync``: upload parameters that
        changed since the last call first (version stamps; a changed weight drops the library's captured graphs) - a
        rollout loop passes False after syncing once per window."""
        assert x.is_contiguous() and out.is_contiguous() and x.dtype == out.dtype == torch.float32
        self._ensure_native(x.device, x.shape[0])
        if sync or not self._uploaded:
            self.sync_weights()
        _lib.check(_lib.lib().ace_sfno_forward_graph(self._native, _lib.ptr(x), _lib.ptr(out), x.shape[0],
                                                     _lib.current_stream()))
        return out

    # ------------------------------------------------------------------ test taps
    def forward_with_taps(self, x: torch.Tensor):
        """forward + the activation after the encoder and after every block (parity debugging)."""
        x = self._prepare(x)
        self.sync_weights()
        L = _lib.lib()
        _lib.check(L.ace_sfno_set_taps(self._native, 1))
        out = torch.empty(x.shape[0], self.out_chans, *self.img_shape, dtype=torch.float32, device=x.device)
        _lib.check(L.ace_sfno_forward(self._native, _lib.ptr(x), _lib.ptr(out), x.shape[0], _lib.current_stream()))
        taps = []
        for i in range(-1, self.num_layers):
            t = torch.empty(x.shape[0], self.embed_dim, *self.img_shape, dtype=torch.float32, device=x.device)
            _lib.check(L.ace_sfno_get_tap(self._native, i, _lib.ptr(t), x.shape[0], _lib.current_stream()))
            taps.append(t)
        _lib.check(L.ace_sfno_set_taps(self._native, 0))
        return out, taps


@ModuleSelector.register("SphericalFourierNeuralOperatorNet")
@dataclasses.dataclass
class SphericalFourierNeuralOperatorBuilder(ModuleConfig):
    """Same type string and field set as fme/ace/registry/sfno.py:14-42."""

    spectral_transform: str = "sht"
    filter_type: str = "linear"
    operator_type: str = "diagonal"
    scale_factor: int = 1
    residual_filter_factor: int = 1
    embed_dim: int = 256
    num_layers: int = 12
    hard_thresholding_fraction: float = 1.0
    normalization_layer: str = "instance_norm"
    use_mlp: bool = True
    activation_function: str = "gelu"
    encoder_layers: int = 1
    pos_embed: bool = True
    big_skip: bool = True
    rank: float = 1.0
    factorization: Optional[str] = None
    separable: bool = False
    complex_network: bool = True
    complex_activation: str = "real"
    spectral_layers: int = 1
    checkpointing: int = 0
    data_grid: Literal["legendre-gauss", "equiangular"] = "legendre-gauss"

    def build(self, n_in_channels: int, n_out_channels: int, dataset_info):
        if len(dataset_info.all_labels) > 0:
            raise ValueError("SphericalFourierNeuralOperatorNet does not support labels")
        return SphericalFourierNeuralOperatorNet(
            params=self, in_chans=n_in_channels, out_chans=n_out_channels, img_shape=dataset_info.img_shape
        )
