"""NoiseConditionedSFNO on the native library.

Mirror of the reference's plugin API for this model family:
  * ``NoiseConditionedSFNOBuilder`` - same registry type string "NoiseConditionedSFNO" and field set as
    fme/ace/registry/stochastic_sfno.py:183-397, ``build(n_in, n_out, dataset_info)`` returns
  * ``NoiseConditionedSFNO`` - an nn.Module that HOLDS the parameters under the reference's state_dict names
    (``conditional_model.…``: strict load_state_dict of reference checkpoints) and whose forward draws the conditioning
    noise as ``NoiseConditionedModel.forward`` does (stochastic_sfno.py:128-172: gaussian ``randn`` or isotropic via the
    inverse SHT of random spectral coefficients, stochastic_sfno.py:21-47) and runs the C-ABI forward
    (ace_sfno_forward_conditioned).  ``forward(x, noise=...)`` takes the noise from the caller instead (parity tests).

Supported: filter_type "linear", dhconv, gaussian / isotropic noise, affine_norms, normalize_big_skip,
filter_num_groups, use_mlp, activation, encoder_layers, pos_embed, big_skip, data_grid, and the label / positional context
(dataset_info.all_labels, label_embed_dim, context_pos_embed_dim: stochastic_sfno.py:88-175, layers.py:160-318) - the host
forms ONE conditioning field cat(noise, positional context, label planes, ones) and merges the per-norm weights accordingly, so
the native conditional norm is unchanged.  Everything else the reference builder accepts (LoRA, spectral_ratio < 1, local blocks,
filter_residual / filter_output, global_layer_norm, clip_latent_global_means, filter_preserves_global_mean) raises at build
time.
"""
import ctypes
import dataclasses
import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from .rand import randn
from .registry import ModuleConfig, ModuleSelector
from .sfno import _ACT, _ACT_LAYER, _GRID, _PRECISION, DEFAULT_PRECISION, trunc_normal_
from .sht import InverseRealSHT


class _CondNorm(nn.Module):
    """parameter holder of ConditionalLayerNorm (conditional_sfno/layers.py:143-243) in the reference's registration order:
    label, noise and positional conditioning (embed_dim_scalar is 0 in this family), then the channel norm's affine"""

    def __init__(self, n_channels: int, noise_dim: int, affine: bool, label_dim: int = 0, pos_dim: int = 0):
        super().__init__()
        if label_dim > 0:
            self.W_scale_labels = nn.Linear(label_dim, n_channels)
            self.W_bias_labels = nn.Linear(label_dim, n_channels)
            for lin in (self.W_scale_labels, self.W_bias_labels):
                nn.init.constant_(lin.weight, 0.0)
                nn.init.constant_(lin.bias, 0.0)
        if noise_dim > 0:
            self.W_scale_2d = nn.Conv2d(noise_dim, n_channels, 1, bias=False)
            self.W_bias_2d = nn.Conv2d(noise_dim, n_channels, 1, bias=False)
            nn.init.constant_(self.W_scale_2d.weight, 0.0)
            nn.init.constant_(self.W_bias_2d.weight, 0.0)
        if pos_dim > 0:
            self.W_scale_pos = nn.Conv2d(pos_dim, n_channels, 1, bias=False)
            self.W_bias_pos = nn.Conv2d(pos_dim, n_channels, 1, bias=False)
            nn.init.constant_(self.W_scale_pos.weight, 0.0)
            nn.init.constant_(self.W_bias_pos.weight, 0.0)
        if affine:
            self.norm = nn.Module()
            self.norm.weight = nn.Parameter(torch.ones(n_channels))
            self.norm.bias = nn.Parameter(torch.zeros(n_channels))

    def merged(self, which: str) -> torch.Tensor:
        """[C, J_total] weight of ONE 1 x 1 convolution over the merged conditioning field cat(noise, positional context, label
        planes, ones): [W_2d | W_pos | W_labels.weight | W_labels.bias] - the native conditional norm knows one conditioning
        field; labels enter as constant planes (their value per sample) and the label bias as an all-ones plane."""
        parts = []
        for name in (f"W_{which}_2d", f"W_{which}_pos"):
            if hasattr(self, name):
                parts.append(getattr(self, name).weight.detach()[:, :, 0, 0])
        if hasattr(self, f"W_{which}_labels"):
            lin = getattr(self, f"W_{which}_labels")
            parts += [lin.weight.detach(), lin.bias.detach()[:, None]]
        return torch.cat(parts, dim=1).float().contiguous()


class _Filter(nn.Module):
    """SpectralFilterLayer -> SpectralConvS2 parameters (conditional_sfno/s2convolutions.py:226-270)"""

    def __init__(self, C: int, L: int, G: int):
        super().__init__()
        self.filter = nn.Module()
        scale = math.sqrt(1 / C) * torch.ones(L, 1, 1, 2)
        scale[0, :] *= math.sqrt(2.0)
        self.filter.weight = nn.Parameter(scale * torch.randn(G, L, C // G, C // G, 2))
        self.filter.bias = nn.Parameter(torch.zeros(1, C, 1, 1))
        self._shape = (G, L, C // G)
        # checkpoints written before the grouped layout load as the reference loads them (s2convolutions.py:276-365)
        self.filter._register_load_state_dict_pre_hook(self._legacy_weight_layouts)

    def _legacy_weight_layouts(self, state_dict, prefix, *_):
        """ungrouped (C, C, L, 2) -> a singleton group; then the old (G, in / G, out / G, L, 2) order -> (G, L, out / G, in / G, 2)"""
        key = prefix + "weight"
        w = state_dict.get(key)
        if w is None:
            return
        G, L, Cg = self._shape
        if tuple(w.shape) == (G * Cg, G * Cg, L, 2):
            w = w.view(1, *w.shape)
        if w.ndim == 5 and tuple(w.shape) == (G, Cg, Cg, L, 2):
            w = w.permute(0, 3, 2, 1, 4)
        state_dict[key] = w


class _Block(nn.Module):
    def __init__(self, C, L, G, noise_dim, affine, use_mlp, mlp_ratio, act_layer, label_dim=0, pos_dim=0):
        super().__init__()
        self.norm0 = _CondNorm(C, noise_dim, affine, label_dim, pos_dim)
        self.filter = _Filter(C, L, G)
        self.inner_skip = nn.Conv2d(C, C, 1, 1)
        self.norm1 = _CondNorm(C, noise_dim, affine, label_dim, pos_dim)
        if use_mlp:
            hid = int(C * mlp_ratio)
            self.mlp = nn.Module()
            self.mlp.fwd = nn.Sequential(nn.Conv2d(C, hid, 1, bias=True), act_layer(), nn.Conv2d(hid, C, 1, bias=True))


class _ConditionalNet(nn.Module):
    """parameter tree of the conditional SphericalFourierNeuralOperatorNet (conditional_sfno/sfnonet.py:496-768)"""

    def __init__(self, cfg: "NoiseConditionedSFNOBuilder", in_chans, out_chans, img_shape, label_dim: int = 0):
        super().__init__()
        C = cfg.embed_dim
        pos_dim = cfg.context_pos_embed_dim
        L = int(img_shape[0] * 1.0)
        act_layer = _ACT_LAYER[cfg.activation_function]
        enc, cur = [], in_chans
        for _ in range(cfg.encoder_layers):
            enc += [nn.Conv2d(cur, C, 1, bias=True), act_layer()]
            cur = C
        enc.append(nn.Conv2d(cur, C, 1, bias=False))
        self.encoder = nn.Sequential(*enc)
        self.blocks = nn.ModuleList([_Block(C, L, cfg.filter_num_groups, cfg.noise_embed_dim, cfg.affine_norms, cfg.use_mlp,
                                            cfg.mlp_ratio, act_layer, label_dim, pos_dim) for _ in range(cfg.num_layers)])
        dec, cur = [], C + int(cfg.big_skip) * in_chans
        for _ in range(cfg.encoder_layers):
            dec += [nn.Conv2d(cur, C, 1, bias=True), act_layer()]
            cur = C
        dec.append(nn.Conv2d(cur, out_chans, 1, bias=False))
        self.decoder = nn.Sequential(*dec)
        if cfg.pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, C, img_shape[0], img_shape[1]))
            trunc_normal_(self.pos_embed, std=0.02)
        if cfg.normalize_big_skip and cfg.big_skip:
            self.norm_big_skip = _CondNorm(in_chans, cfg.noise_embed_dim, cfg.affine_norms, label_dim, pos_dim)


class NoiseConditionedSFNO(nn.Module):
    """NoiseConditionedModel (stochastic_sfno.py:50-180) around the conditional network: draws the noise, embeds the labels,
    forms the positional context (pos_embed + labels . label_pos_embed) and hands the native network ONE conditioning field
    cat(noise, positional context, label planes, ones) with the per-norm weights merged accordingly (_CondNorm.merged)."""

    def __init__(self, cfg: "NoiseConditionedSFNOBuilder", in_chans: int, out_chans: int, img_shape, n_labels: int = 0):
        super().__init__()
        self.cfg = cfg
        self.in_chans, self.out_chans = in_chans, out_chans
        self.img_shape = (int(img_shape[0]), int(img_shape[1]))
        self.embed_dim = cfg.noise_embed_dim                       # the reference wrapper's attribute (noise channels)
        if cfg.label_embed_dim > 0 and n_labels == 0:
            raise ValueError("label_embed_dim > 0 requires n_labels > 0")
        self.n_labels = n_labels
        self.label_dim = cfg.label_embed_dim if cfg.label_embed_dim > 0 else n_labels      # effective_label_dim
        self.pos_dim = cfg.context_pos_embed_dim
        # registration order as the reference's wrapper: conditional_model, label_embedding, pos_embed, label_pos_embed
        self.conditional_model = _ConditionalNet(cfg, in_chans, out_chans, self.img_shape, self.label_dim)
        if cfg.label_embed_dim > 0:
            self.label_embedding = nn.Linear(n_labels, cfg.label_embed_dim)
        if self.pos_dim > 0:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.pos_dim, *self.img_shape))
            trunc_normal_(self.pos_embed, std=0.02)
            if self.label_dim > 0:
                self.label_pos_embed = nn.Parameter(torch.zeros(self.label_dim, self.pos_dim, *self.img_shape))
                trunc_normal_(self.label_pos_embed, std=0.02)
        # conditioning channels the native network sees: noise | positional context | label planes | ones (label bias)
        self.cond_dim = cfg.noise_embed_dim + self.pos_dim + (self.label_dim + 1 if self.label_dim > 0 else 0)
        self._lmax = self.img_shape[0]
        self._mmax = self.img_shape[1] // 2 + 1
        self._isht: Optional[InverseRealSHT] = None
        self.precision = os.environ.get("ACE_SFNO_PRECISION", DEFAULT_PRECISION)
        self._native = None
        self._native_key = None
        self._uploaded = {}

    # ------------------------------------------------------------------ native plumbing
    def _config_struct(self, max_batch: int) -> _lib.AceSfnoConfig:
        c = self.cfg
        return _lib.AceSfnoConfig(
            in_chans=self.in_chans, out_chans=self.out_chans, nlat=self.img_shape[0], nlon=self.img_shape[1],
            embed_dim=c.embed_dim, num_layers=c.num_layers, scale_factor=1, hard_thresholding_fraction=1.0,
            operator_type=1, normalization_layer=2, activation_function=_ACT[c.activation_function],
            use_mlp=int(c.use_mlp), mlp_ratio=float(c.mlp_ratio), encoder_layers=c.encoder_layers,
            pos_embed=int(c.pos_embed), big_skip=int(c.big_skip), data_grid=_GRID[c.data_grid], max_batch=max_batch,
            precision=_PRECISION[self.precision], noise_embed_dim=self.cond_dim, affine_norms=int(c.affine_norms),
            normalize_big_skip=int(c.normalize_big_skip), filter_num_groups=c.filter_num_groups)

    def set_precision(self, precision: str):
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
        if self._native is None or self._native_key[0] != device.index or batch > self._native_key[1]:
            self._release_native()
            handle = ctypes.c_void_p()
            cfg = self._config_struct(max_batch=batch)
            with torch.cuda.device(device):
                _lib.check(_lib.lib().ace_sfno_create(ctypes.byref(cfg), ctypes.byref(handle)))
            self._native, self._native_key = handle, (device.index, batch)
            self.__dict__["_native_destroy"] = _lib.lib().ace_sfno_destroy      # freed by the library that made it

    def sync_weights(self, force: bool = False):
        L = _lib.lib()
        stream = _lib.current_stream()
        changed = 0
        merged_context = self.pos_dim > 0 or self.label_dim > 0
        if merged_context:
            # the conditional norms' label / positional weights ride on the native W_scale_2d / W_bias_2d (merged along the
            # conditioning channels); the wrapper's own parameters are consumed on the host (forward)
            for mname, mod in self.conditional_model.named_modules():
                if not isinstance(mod, _CondNorm):
                    continue
                for which in ("scale", "bias"):
                    parts = [p for n, p in mod.named_parameters() if n.startswith(f"W_{which}_")]
                    stamp = tuple((p.data_ptr(), p._version) for p in parts)
                    key = f"{mname}.W_{which}_2d.weight#merged"
                    if not force and self._uploaded.get(key) == stamp:
                        continue
                    t = mod.merged(which)
                    _lib.check(L.ace_sfno_set_weight(self._native, f"{mname}.W_{which}_2d.weight".encode(), _lib.ptr(t), t.numel(), stream))
                    self._uploaded[key] = stamp
                    changed += 1
        for name, p in self.state_dict(keep_vars=True).items():
            if merged_context and (".W_scale_" in name or ".W_bias_" in name or not name.startswith("conditional_model.")):
                continue
            stamp = (p.data_ptr(), p._version)
            if not force and self._uploaded.get(name) == stamp:
                continue
            t = p.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            native_name = name[len("conditional_model."):] if name.startswith("conditional_model.") else name
            _lib.check(L.ace_sfno_set_weight(self._native, native_name.encode(), _lib.ptr(t), t.numel(), stream))
            self._uploaded[name] = stamp
            changed += 1
        return changed

    # ------------------------------------------------------------------ noise (stochastic_sfno.py:21-47, 128-146)
    def draw_noise(self, batch: int, device: torch.device) -> torch.Tensor:
        J = self.cfg.noise_embed_dim
        if self.cfg.noise_type == "isotropic":
            shape = (batch, J, self._lmax, self._mmax)
            real = randn(shape, dtype=torch.float32, device=device)      # fme.core.rand: the active CPU generator of a seeded
            imag = randn(shape, dtype=torch.float32, device=device)      # rollout, else the device's global RNG
            imag[..., :, 0] = 0.0
            real[..., :, 1:] /= math.sqrt(2.0)
            imag[..., :, 1:] /= math.sqrt(2.0)
            alm = (real + 1j * imag) * (math.sqrt(4.0 * math.pi) / self._lmax)
            if self._isht is None:
                self._isht = InverseRealSHT(self.img_shape[0], self.img_shape[1], self._lmax, self._mmax, self.cfg.data_grid)
            return self._isht(alm)
        return randn(torch.Size([batch, J, *self.img_shape]), device=device, dtype=torch.float32)

    def conditioning_field(self, batch: int, device: torch.device, labels: Optional[torch.Tensor] = None,
                           noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The (batch, cond_dim, H, W) field every conditional norm of the native network reads: the noise (drawn here unless
        given) and - with a label / positional context - cat(noise, pos_embed + labels . label_pos_embed, label planes, ones)
        (stochastic_sfno.py:128-175)."""
        if labels is not None and self.label_dim == 0:
            raise ValueError("labels were provided but the model was built without labels (dataset_info.all_labels is empty)")
        if labels is None and self.label_dim > 0:
            raise ValueError("labels must be provided")
        if noise is None:
            noise = self.draw_noise(batch, device)
        noise = noise.to(device=device, dtype=torch.float32).contiguous()
        if tuple(noise.shape) != (batch, self.cfg.noise_embed_dim, *self.img_shape):
            raise ValueError(f"noise must have shape {(batch, self.cfg.noise_embed_dim, *self.img_shape)}, got {tuple(noise.shape)}")
        if self.pos_dim == 0 and self.label_dim == 0:
            return noise
        fields = [noise]
        lab = None
        if self.label_dim > 0:
            lab = labels.to(device=device, dtype=torch.float32)
            if tuple(lab.shape) != (batch, self.n_labels):
                raise ValueError(f"labels must have shape {(batch, self.n_labels)}, got {tuple(lab.shape)}")
            if hasattr(self, "label_embedding"):
                lab = torch.nn.functional.linear(lab, self.label_embedding.weight.detach(), self.label_embedding.bias.detach())
        if self.pos_dim > 0:
            pos = self.pos_embed.detach().repeat(batch, 1, 1, 1)
            if lab is not None:
                pos = pos + torch.einsum("bl,lpxy->bpxy", lab, self.label_pos_embed.detach())
            fields.append(pos)
        if lab is not None:
            fields.append(lab[:, :, None, None].expand(batch, self.label_dim, *self.img_shape))
            fields.append(torch.ones(batch, 1, *self.img_shape, dtype=torch.float32, device=device))
        return torch.cat(fields, dim=1).contiguous()

    def forward(self, x: torch.Tensor, labels: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        if labels is not None and self.label_dim == 0:
            raise ValueError("labels were provided but the model was built without labels (dataset_info.all_labels is empty)")
        if labels is None and self.label_dim > 0:
            raise ValueError("labels must be provided")
        x = x.reshape(-1, *x.shape[-3:])
        if x.shape[1] != self.in_chans or tuple(x.shape[-2:]) != self.img_shape:
            raise AssertionError(f"expected input (B, {self.in_chans}, {self.img_shape[0]}, {self.img_shape[1]}), "
                                 f"got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("NoiseConditionedSFNO (ace_amd) runs on an MI355X only: move the module and its input to "
                               "'cuda'. There is no CPU fallback.")
        x = x.float().contiguous()
        B = x.shape[0]
        noise = self.conditioning_field(B, x.device, labels=labels, noise=noise)
        self._ensure_native(x.device, B)
        self.sync_weights()
        out = torch.empty(B, self.out_chans, *self.img_shape, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ace_sfno_forward_conditioned(self._native, _lib.ptr(x), _lib.ptr(noise), _lib.ptr(out), B,
                                                           _lib.current_stream()))
        return out


@ModuleSelector.register("NoiseConditionedSFNO")
@dataclasses.dataclass
class NoiseConditionedSFNOBuilder(ModuleConfig):
    """Same type string and field set as fme/ace/registry/stochastic_sfno.py:183-305."""

    spectral_transform: str = "sht"
    filter_type: str = "linear"
    operator_type: str = "dhconv"
    residual_filter_factor: int = 1
    embed_dim: int = 256
    noise_embed_dim: int = 256
    context_pos_embed_dim: int = 0
    label_embed_dim: int = 0
    noise_type: str = "gaussian"
    global_layer_norm: bool = False
    num_layers: int = 12
    use_mlp: bool = True
    mlp_ratio: float = 2.0
    activation_function: str = "gelu"
    encoder_layers: int = 1
    pos_embed: bool = True
    big_skip: bool = True
    rank: float = 1.0
    factorization: None = None
    separable: bool = False
    complex_network: bool = True
    complex_activation: str = "real"
    spectral_layers: int = 1
    checkpointing: int = 0
    data_grid: str = "legendre-gauss"
    filter_residual: bool = False
    filter_output: bool = False
    local_blocks: Optional[List[int]] = None
    normalize_big_skip: bool = False
    affine_norms: bool = False
    filter_num_groups: int = 1
    lora_rank: int = 0
    lora_alpha: Optional[float] = None
    spectral_lora_rank: int = 0
    spectral_lora_alpha: Optional[float] = None
    filter_preserves_global_mean: bool = False
    spectral_ratio: float = 1.0
    clip_latent_global_means: bool = False

    def __post_init__(self):
        # the reference's own checks (stochastic_sfno.py:307-327)
        if self.context_pos_embed_dim > 0 and self.pos_embed:
            raise ValueError("context_pos_embed_dim and pos_embed should not both be set")
        if self.factorization is not None:
            raise ValueError("The 'factorization' parameter is no longer supported.")
        if self.separable:
            raise ValueError("The 'separable' parameter is no longer supported.")
        if self.operator_type != "dhconv":
            raise ValueError("Only 'dhconv' operator_type is supported for NoiseConditionedSFNO models.")
        if not (0.0 < self.spectral_ratio <= 1.0):
            raise ValueError("spectral_ratio must be in (0, 1]")

    def _unsupported(self) -> List[str]:
        out = []
        if self.filter_type != "linear":
            out.append(f"filter_type='{self.filter_type}'")
        for name, default in (("global_layer_norm", False),
                              ("filter_residual", False), ("filter_output", False), ("lora_rank", 0),
                              ("spectral_lora_rank", 0), ("filter_preserves_global_mean", False), ("spectral_ratio", 1.0),
                              ("clip_latent_global_means", False), ("residual_filter_factor", 1)):
            if getattr(self, name) != default:
                out.append(name)
        if self.local_blocks:
            out.append("local_blocks")
        return out

    def build(self, n_in_channels: int, n_out_channels: int, dataset_info) -> nn.Module:
        n_labels = len(getattr(dataset_info, "all_labels", ()) or ())
        bad = self._unsupported()
        if bad:
            raise NotImplementedError("NoiseConditionedSFNO options outside the accelerated hot path: " + ", ".join(bad))
        if self.noise_type not in ("isotropic", "gaussian"):
            raise ValueError(f"unknown noise_type {self.noise_type}")
        if self.activation_function not in _ACT:
            raise ValueError(f"Unknown activation function {self.activation_function}")
        if self.embed_dim % self.filter_num_groups != 0:
            raise ValueError("embed_dim must be divisible by filter_num_groups")
        return NoiseConditionedSFNO(self, n_in_channels, n_out_channels, dataset_info.img_shape, n_labels=n_labels)
