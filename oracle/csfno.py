"""NoiseConditionedSFNO forward on CPU (oracle; test infrastructure only) - groundwork for SURVEY 8(f) rank 1.

Functional restatement - weights come in as the reference module's ``state_dict`` - of the inference path of the
configuration family ACE ships today (configs/baselines/era5/ace-train-config-1-step-pretrain.yaml:93-108:
NoiseConditionedSFNO, filter_type linear, dhconv, isotropic or gaussian noise, affine_norms, normalize_big_skip):

* ``isotropic_noise`` / ``NoiseConditionedModel.forward``     fme/ace/registry/stochastic_sfno.py:21-47, 128-172
* ``ChannelLayerNorm`` / ``ConditionalLayerNorm.forward``     fme/core/models/conditional_sfno/layers.py:95-141, 245-318
* ``SpectralConvS2.forward`` (grouped dhconv, ``_contract_dhconv``)   .../s2convolutions.py:119-135, 367-433
* ``FourierNeuralOperatorBlock.forward``                      .../sfnonet.py:388-437
* ``SphericalFourierNeuralOperatorNet.forward``               .../sfnonet.py:770-824

Not covered (raise): labels / scalar / positional context embeddings, LoRA, spectral_ratio < 1, local (DISCO) blocks,
filter_residual / filter_output, global_layer_norm, clip_latent_global_means, filter_preserves_global_mean.
Pinned against the real reference module (imported under stubs) by tests/golden/gen_csfno_*.pt.
"""
import dataclasses
import math
from typing import Optional

import torch
import torch.nn.functional as F

from .sht import InverseRealSHT, RealSHT


@dataclasses.dataclass
class CSFNOConfig:
    """The fields of NoiseConditionedSFNOBuilder (stochastic_sfno.py:266-305) this oracle honours."""
    in_chans: int
    out_chans: int
    img_shape: tuple
    embed_dim: int = 256
    noise_embed_dim: int = 256
    noise_type: str = "gaussian"
    num_layers: int = 12
    use_mlp: bool = True
    mlp_ratio: float = 2.0
    activation_function: str = "gelu"
    encoder_layers: int = 1
    pos_embed: bool = True
    big_skip: bool = True
    data_grid: str = "legendre-gauss"
    normalize_big_skip: bool = False
    affine_norms: bool = False
    filter_num_groups: int = 1
    hard_thresholding_fraction: float = 1.0
    context_pos_embed_dim: int = 0     # learned positional context (stochastic_sfno.py:105-125)
    label_embed_dim: int = 0           # > 0: Linear(n_labels, label_embed_dim) in front of the label conditioning


_ACT = {"gelu": F.gelu, "relu": F.relu, "silu": F.silu}


def isotropic_noise(leading_shape, lmax: int, mmax: int, isht, dtype=torch.float32, generator=None) -> torch.Tensor:
    """stochastic_sfno.py:21-47: a_lm ~ CN(0, 1) (real for m = 0), scaled so that the field has unit variance; draws
    in the reference's order (real parts, then imaginary parts, both fp32) from torch's global RNG or - fme/core/rand.py:55-63
    under ``use_generator`` - from the given CPU generator."""
    shape = (*leading_shape, lmax, mmax)
    real = torch.randn(shape, dtype=torch.float32, generator=generator)
    imag = torch.randn(shape, dtype=torch.float32, generator=generator)
    imag[..., :, 0] = 0.0
    sqrt2 = math.sqrt(2.0)
    real[..., :, 1:] /= sqrt2
    imag[..., :, 1:] /= sqrt2
    scale = math.sqrt(4.0 * math.pi) / lmax
    alm = (real + 1j * imag) * scale
    return isht(alm.to(torch.complex128 if dtype == torch.float64 else torch.complex64))


class CSFNOOracle:
    def __init__(self, cfg: CSFNOConfig, state: dict, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        strip = lambda k: k.removeprefix("module.").removeprefix("conditional_model.")
        inner = lambda k: k.removeprefix("module.").startswith("conditional_model.")
        # parameters of the conditional network (prefix stripped) and of the NoiseConditionedModel wrapper around it
        # (stochastic_sfno.py:88-125: label_embedding.*, pos_embed = the positional CONTEXT, label_pos_embed)
        self.p = {strip(k): v.detach().to("cpu").to(dtype) for k, v in state.items() if torch.is_floating_point(v) and inner(k)}
        self.wrap = {k.removeprefix("module."): v.detach().to("cpu").to(dtype) for k, v in state.items()
                     if torch.is_floating_point(v) and not inner(k)}
        if not self.p:      # a state without the wrapper prefix: everything belongs to the conditional network
            self.p, self.wrap = self.wrap, {}
        for k in self.p:
            if "lora" in k or "W_scale." in k or k.endswith("_gm_min"):
                raise NotImplementedError(f"parameter {k}: outside the oracle's configuration family")
        h, w = cfg.img_shape
        L = int(h * cfg.hard_thresholding_fraction)
        M = int((w // 2 + 1) * cfg.hard_thresholding_fraction)
        self.L, self.M = L, M
        mk = lambda cls, grid: cls(h, w, lmax=L, mmax=M, grid=grid, dtype=dtype)
        self.trans_down, self.itrans_up = mk(RealSHT, cfg.data_grid), mk(InverseRealSHT, cfg.data_grid)
        self.trans, self.itrans = mk(RealSHT, "legendre-gauss"), mk(InverseRealSHT, "legendre-gauss")
        self.act = _ACT[cfg.activation_function]

    # ---- layers.py:95-141 + 245-318: noise, label and positional conditioning (embed_dim_scalar = 0 in this family)
    def _cln(self, prefix: str, x: torch.Tensor, ctx, eps: float = 1e-5) -> torch.Tensor:
        p = self.p
        noise, labels, pos = ctx if isinstance(ctx, tuple) else (ctx, None, None)
        mean = x.mean(dim=-3, keepdim=True)
        var = x.var(dim=-3, keepdim=True, unbiased=False)
        y = (x - mean) * torch.rsqrt(var + eps)
        if prefix + "norm.weight" in p:
            y = y * p[prefix + "norm.weight"].view(1, -1, 1, 1) + p[prefix + "norm.bias"].view(1, -1, 1, 1)
        scale = torch.ones(list(x.shape[:-2]) + [1, 1], dtype=x.dtype)
        bias = torch.zeros(list(x.shape[:-2]) + [1, 1], dtype=x.dtype)
        if prefix + "W_scale_2d.weight" in p:
            scale = scale + F.conv2d(noise, p[prefix + "W_scale_2d.weight"])
            bias = bias + F.conv2d(noise, p[prefix + "W_bias_2d.weight"])
        if prefix + "W_scale_labels.weight" in p:
            if labels is None:
                raise ValueError("labels must be provided")
            scale = scale + F.linear(labels, p[prefix + "W_scale_labels.weight"], p[prefix + "W_scale_labels.bias"])[:, :, None, None]
            bias = bias + F.linear(labels, p[prefix + "W_bias_labels.weight"], p[prefix + "W_bias_labels.bias"])[:, :, None, None]
        if prefix + "W_scale_pos.weight" in p:
            if pos is None:
                raise ValueError("embedding_pos must be provided")
            scale = scale + F.conv2d(pos, p[prefix + "W_scale_pos.weight"])
            bias = bias + F.conv2d(pos, p[prefix + "W_bias_pos.weight"])
        return y * scale + bias

    # ---- s2convolutions.py:367-433 (spectral_ratio 1, no LoRA)
    def _filter(self, i: int, x: torch.Tensor):
        cfg, p = self.cfg, self.p
        fwd = self.trans_down if i == 0 else self.trans
        inv = self.itrans_up if i == cfg.num_layers - 1 else self.itrans
        residual = x
        xs = fwd(x)
        if (fwd.grid != inv.grid):
            residual = inv(xs)
        B, C, H, W = xs.shape
        G = cfg.filter_num_groups
        w = p[f"blocks.{i}.filter.filter.weight"]                      # (G, L, O/G, I/G, 2)
        wc = torch.view_as_complex(w.contiguous())
        xg = xs.reshape(B, G, C // G, H, W)
        out = torch.einsum("bgixy,gxoi->bgoxy", xg, wc).reshape(B, C, H, W)
        y = inv(out)
        bias = p.get(f"blocks.{i}.filter.filter.bias")
        if bias is not None:
            y = y + bias
        return y, residual

    def _block(self, i: int, x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        p = self.p
        pre = f"blocks.{i}."
        xn = self._cln(pre + "norm0.", x, noise)
        y, residual = self._filter(i, xn)
        y = y + F.conv2d(residual, p[pre + "inner_skip.weight"], p[pre + "inner_skip.bias"])
        y = self.act(y)
        y = self._cln(pre + "norm1.", y, noise)
        if self.cfg.use_mlp:
            y = F.conv2d(y, p[pre + "mlp.fwd.0.weight"], p[pre + "mlp.fwd.0.bias"])
            y = self.act(y)
            y = F.conv2d(y, p[pre + "mlp.fwd.2.weight"], p[pre + "mlp.fwd.2.bias"])
        return y + residual                                            # outer skip = identity on the filter's residual

    def forward(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """x: (B, in_chans, H, W).  ``noise`` (B, noise_embed_dim, H, W): if None it is drawn from torch's global RNG
        exactly as the reference does (seed with torch.manual_seed to reproduce), or from ``generator`` - the CPU generator a seeded
        rollout's ``StepperState.random_state`` carries (fme/ace/stepper/single_module.py:1063-1068).  ``labels`` (B, n_labels): one-hot (or
        soft) label encoding, stochastic_sfno.py:128-175."""
        cfg, p = self.cfg, self.p
        x = x.reshape(-1, *x.shape[-3:]).to(self.dtype)
        if noise is None:
            if cfg.noise_type == "isotropic":
                noise = isotropic_noise((x.shape[0], cfg.noise_embed_dim), self.L, self.M, self.itrans_up, self.dtype, generator)
            else:
                noise = torch.randn(torch.Size([x.shape[0], cfg.noise_embed_dim, *x.shape[-2:]]), dtype=torch.float32,
                                    generator=generator)
        noise = noise.to(self.dtype)
        w = self.wrap
        if labels is not None:
            labels = labels.to(self.dtype)
            if "label_embedding.weight" in w:
                labels = F.linear(labels, w["label_embedding.weight"], w["label_embedding.bias"])
        pos = None
        if cfg.context_pos_embed_dim > 0:
            pos = w["pos_embed"].repeat(noise.shape[0], 1, 1, 1)
            if "label_pos_embed" in w and labels is not None:
                pos = pos + torch.einsum("bl,lpxy->bpxy", labels, w["label_pos_embed"])
        noise = (noise, labels, pos)       # the Context every conditional norm reads
        if cfg.big_skip:
            residual = self._cln("norm_big_skip.", x, noise) if cfg.normalize_big_skip else x
        h = x
        for j in range(cfg.encoder_layers):
            h = self.act(F.conv2d(h, p[f"encoder.{2 * j}.weight"], p[f"encoder.{2 * j}.bias"]))
        h = F.conv2d(h, p[f"encoder.{2 * cfg.encoder_layers}.weight"])
        if cfg.pos_embed:
            h = h + p["pos_embed"]
        for i in range(cfg.num_layers):
            h = self._block(i, h, noise)
        if cfg.big_skip:
            h = torch.cat((h, residual), dim=1)
        for j in range(cfg.encoder_layers):
            h = self.act(F.conv2d(h, p[f"decoder.{2 * j}.weight"], p[f"decoder.{2 * j}.bias"]))
        return F.conv2d(h, p[f"decoder.{2 * cfg.encoder_layers}.weight"])
