"""SFNO network forward on CPU (oracle; test infrastructure only).

Functional restatement - weights come in as a ``state_dict`` with the
reference's parameter names - of:

* ``SpectralConvS2.forward``      fme/ace/models/modulus/s2convolutions.py:162-197
* ``_contract_dhconv/_diagonal``  fme/ace/models/modulus/contractions.py:183-195 / 169-180
* ``MLP``                         fme/ace/models/modulus/layers.py:97-137
* ``FourierNeuralOperatorBlock``  fme/ace/models/modulus/sfnonet.py:217-252
* ``SphericalFourierNeuralOperatorNet.__init__/forward``  sfnonet.py:341-685, 713-749

using the same torch-CPU op sequence (conv2d / instance_norm / einsum / GELU),
so it doubles as the timed CPU baseline ("port" of the reference CPU path).
"""

import dataclasses
from typing import Optional

import torch
import torch.nn.functional as F

from .sht import InverseRealSHT, RealSHT


@dataclasses.dataclass
class SFNOConfig:
    """Field set of SphericalFourierNeuralOperatorBuilder (fme/ace/registry/sfno.py:21-42)
    plus the constructor arguments the builder passes (sfno.py:54-59)."""

    in_chans: int
    out_chans: int
    img_shape: tuple
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
    mlp_ratio: float = 2.0
    activation_function: str = "gelu"
    encoder_layers: int = 1
    pos_embed: bool = True
    big_skip: bool = True
    data_grid: str = "legendre-gauss"


_ACT = {"gelu": F.gelu, "relu": F.relu, "silu": F.silu}


class SFNOOracle:
    def __init__(self, cfg: SFNOConfig, state: dict, dtype=torch.float32):
        assert cfg.spectral_transform == "sht" and cfg.filter_type == "linear"
        self.cfg = cfg
        self.dtype = dtype
        self.p = {k.removeprefix("module."): v.detach().to("cpu").to(dtype) for k, v in state.items()
                  if isinstance(v, torch.Tensor)}
        H, W = cfg.img_shape
        # sfnonet.py:467-472
        self.h = int(H // cfg.scale_factor)
        self.w = int(W // cfg.scale_factor)
        modes_lat = int(self.h * cfg.hard_thresholding_fraction)
        modes_lon = int((self.w // 2 + 1) * cfg.hard_thresholding_fraction)
        # sfnonet.py:498-515
        kw = dict(lmax=modes_lat, mmax=modes_lon, dtype=dtype)
        # sfnonet.py:473-497: the big skip's input is band-limited on the data grid when residual_filter_factor != 1
        if cfg.residual_filter_factor != 1:
            rl, rm = int(H // cfg.residual_filter_factor), int(W // cfg.residual_filter_factor // 2 + 1)
            self.residual_filter_down = RealSHT(H, W, lmax=rl, mmax=rm, grid=cfg.data_grid, dtype=dtype)
            self.residual_filter_up = InverseRealSHT(H, W, lmax=rl, mmax=rm, grid=cfg.data_grid, dtype=dtype)
        self.trans_down = RealSHT(H, W, grid=cfg.data_grid, **kw)
        self.itrans_up = InverseRealSHT(H, W, grid=cfg.data_grid, **kw)
        self.trans = RealSHT(self.h, self.w, grid="legendre-gauss", **kw)
        self.itrans = InverseRealSHT(self.h, self.w, grid="legendre-gauss", **kw)
        self.act = _ACT[cfg.activation_function]

    # -- pieces ---------------------------------------------------------
    def _norm(self, x, prefix):
        # sfnonet.py:593-601: InstanceNorm2d(eps=1e-6, affine=True, no running stats)
        if self.cfg.normalization_layer == "none":
            return x
        if self.cfg.normalization_layer == "layer_norm":   # sfnonet.py:584-592: nn.LayerNorm((H, W), eps=1e-6), elementwise affine
            return F.layer_norm(x, tuple(x.shape[-2:]), weight=self.p[prefix + ".weight"], bias=self.p[prefix + ".bias"], eps=1e-6)
        assert self.cfg.normalization_layer == "instance_norm"
        return F.instance_norm(x, weight=self.p[prefix + ".weight"], bias=self.p[prefix + ".bias"], eps=1e-6)

    def _filter(self, x, i, fwd, inv):
        # s2convolutions.py:162-197
        p = f"blocks.{i}.filter.filter."
        residual = x
        scale_residual = (fwd.nlat != inv.nlat) or (fwd.nlon != inv.nlon) or (fwd.grid != inv.grid)
        xs = fwd(x)
        if scale_residual:
            residual = inv(xs.contiguous())
        wc = torch.view_as_complex(self.p[p + "weight"].contiguous())
        L, M = inv.lmax, inv.mmax
        xp = torch.zeros_like(xs)
        if self.cfg.operator_type == "dhconv":
            xp[..., :L, :M] = torch.einsum("bixy,iox->boxy", xs[..., :L, :M], wc)
        elif self.cfg.operator_type == "diagonal":
            xp[..., :L, :M] = torch.einsum("bixy,ioxy->boxy", xs[..., :L, :M], wc)
        else:
            raise ValueError(self.cfg.operator_type)
        y = inv(xp.contiguous())
        y = y + self.p[p + "bias"]
        return y, residual

    def _block(self, x, i):
        # sfnonet.py:217-252 (inner_skip="linear", outer_skip="identity", :625-626)
        n = self.cfg.num_layers
        fwd = self.trans_down if i == 0 else self.trans
        inv = self.itrans_up if i == n - 1 else self.itrans
        p = f"blocks.{i}."
        x_norm = self._norm(x, p + "norm0")
        x, residual = self._filter(x_norm, i, fwd, inv)
        x = x + F.conv2d(residual, self.p[p + "inner_skip.weight"], self.p[p + "inner_skip.bias"])
        x = self.act(x)
        x = self._norm(x, p + "norm1")
        if self.cfg.use_mlp:
            x = F.conv2d(x, self.p[p + "mlp.fwd.0.weight"], self.p[p + "mlp.fwd.0.bias"])
            x = self.act(x)
            x = F.conv2d(x, self.p[p + "mlp.fwd.2.weight"], self.p[p + "mlp.fwd.2.bias"])
        x = x + residual
        return x

    def _mlp_stack(self, x, prefix):
        # encoder/decoder: sfnonet.py:566-577, 660-671
        nl = self.cfg.encoder_layers
        for j in range(nl):
            x = F.conv2d(x, self.p[f"{prefix}.{2 * j}.weight"], self.p[f"{prefix}.{2 * j}.bias"])
            x = self.act(x)
        return F.conv2d(x, self.p[f"{prefix}.{2 * nl}.weight"], None)

    # -- network --------------------------------------------------------
    def forward(self, x: torch.Tensor, return_blocks: bool = False):
        # sfnonet.py:713-749
        x = x.detach().to("cpu").to(self.dtype)
        residual = x
        if self.cfg.residual_filter_factor != 1:   # sfnonet.py:715-716
            residual = self.residual_filter_up(self.residual_filter_down(x))
        x = self._mlp_stack(x, "encoder")
        if self.cfg.pos_embed:
            x = x + self.p["pos_embed"]
        taps = []
        for i in range(self.cfg.num_layers):
            x = self._block(x, i)
            if return_blocks:
                taps.append(x)
        if self.cfg.big_skip:
            x = torch.cat((x, residual), dim=1)
        x = self._mlp_stack(x, "decoder")
        return (x, taps) if return_blocks else x

    __call__ = forward


def init_state(cfg: SFNOConfig, seed: Optional[int] = 0) -> dict:
    """Random-init parameters with the reference's shapes and distributions
    (sfnonet.py:687-697 trunc_normal std 0.02 / zero bias; s2convolutions.py:72-73,148
    scale*randn filter weights; InstanceNorm gamma=1, beta=0).  Generated on CPU
    ("must initialize on CPU to get the same results on GPU", test_sfnonet.py:28).
    NOT draw-order compatible with the reference constructor; parity tests load
    the same tensors into both implementations instead."""
    g = torch.Generator().manual_seed(seed)
    C, H, W = cfg.embed_dim, *cfg.img_shape
    h, w = H // cfg.scale_factor, W // cfg.scale_factor
    L = int(h * cfg.hard_thresholding_fraction)
    M = int((w // 2 + 1) * cfg.hard_thresholding_fraction)

    def tn(*shape):
        t = torch.empty(*shape)
        torch.nn.init.trunc_normal_(t, std=0.02, a=-2.0, b=2.0, generator=g)
        return t

    st = {}
    cur = cfg.in_chans
    for j in range(cfg.encoder_layers):
        st[f"encoder.{2 * j}.weight"] = tn(C, cur, 1, 1)
        st[f"encoder.{2 * j}.bias"] = torch.zeros(C)
        cur = C
    st[f"encoder.{2 * cfg.encoder_layers}.weight"] = tn(C, cur, 1, 1)
    if cfg.pos_embed:
        st["pos_embed"] = tn(1, C, H, W)
    scale = 1.0 / (C * C)
    hid = int(C * cfg.mlp_ratio)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        if cfg.normalization_layer == "instance_norm":
            for nm in ("norm0", "norm1"):
                # non-trivial affine so the gamma/beta paths are exercised
                st[p + nm + ".weight"] = 1.0 + 0.1 * torch.randn(C, generator=g)
                st[p + nm + ".bias"] = 0.1 * torch.randn(C, generator=g)
        elif cfg.normalization_layer == "layer_norm":
            # sfnonet.py:616-623: norm0 on the block's input grid, norm1 on its output grid (a first block's: the inner one)
            shapes = {"norm0": (H, W) if i == 0 else (h, w), "norm1": (h, w) if (i == 0 or i + 1 < cfg.num_layers) else (H, W)}
            for nm in ("norm0", "norm1"):
                st[p + nm + ".weight"] = 1.0 + 0.1 * torch.randn(*shapes[nm], generator=g)
                st[p + nm + ".bias"] = 0.1 * torch.randn(*shapes[nm], generator=g)
        wshape = (C, C, L, 2) if cfg.operator_type == "dhconv" else (C, C, L, M, 2)
        st[p + "filter.filter.weight"] = scale * torch.randn(*wshape, generator=g)
        st[p + "filter.filter.bias"] = 0.01 * torch.randn(1, C, 1, 1, generator=g)
        st[p + "inner_skip.weight"] = tn(C, C, 1, 1)
        st[p + "inner_skip.bias"] = 0.01 * torch.randn(C, generator=g)
        if cfg.use_mlp:
            st[p + "mlp.fwd.0.weight"] = tn(hid, C, 1, 1)
            st[p + "mlp.fwd.0.bias"] = 0.01 * torch.randn(hid, generator=g)
            st[p + "mlp.fwd.2.weight"] = tn(C, hid, 1, 1)
            st[p + "mlp.fwd.2.bias"] = 0.01 * torch.randn(C, generator=g)
    cur = C + (cfg.in_chans if cfg.big_skip else 0)
    for j in range(cfg.encoder_layers):
        st[f"decoder.{2 * j}.weight"] = tn(C, cur, 1, 1)
        st[f"decoder.{2 * j}.bias"] = 0.01 * torch.randn(C, generator=g)
        cur = C
    st[f"decoder.{2 * cfg.encoder_layers}.weight"] = tn(cfg.out_chans, cur, 1, 1)
    return st
