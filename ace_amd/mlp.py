"""Column-local MLP (registry type "MLP": fme/core/models/mlp/mlp.py:9-57) - Conv2d(1 x 1) / GELU chains, what the reference's
secondary decoder (fme/core/step/secondary_decoder.py) is normally built from - on the native fused 1 x 1 convolution
(``ace_hpx_conv`` with k = 1: one (tap, channel) contraction in the compensated-fp16 MFMA mode with bias and GELU in the epilogue;
the operator is grid-agnostic: [images][channels][rows][pitch % 4 == 0]).

The module IS an ``nn.Sequential`` of the reference's layers (same construction order -> same seeded initialisation, same
state-dict keys "0.weight", "0.bias", "2.weight", ...); only ``forward`` is replaced.  No CPU path."""
import ctypes
import dataclasses
from typing import Dict, List, Tuple

import torch
from torch import nn

from . import _lib
from .registry import ModuleConfig, ModuleSelector


def _round4(n: int) -> int:
    return (n + 3) & ~3


class ColumnMLP(nn.Sequential):
    def __init__(self, in_dim: int, out_dim: int, n_hidden: int, depth: int):
        if depth < 1:
            raise ValueError(f"depth must be >= 1, got {depth}")
        layers: List[nn.Module] = []
        if depth == 1:
            layers.append(nn.Conv2d(in_dim, out_dim, kernel_size=1))
        else:
            dims = [in_dim] + [n_hidden] * (depth - 1)
            for a, b in zip(dims[:-1], dims[1:]):
                layers += [nn.Conv2d(a, b, kernel_size=1), nn.GELU()]
            layers.append(nn.Conv2d(n_hidden, out_dim, kernel_size=1))
        super().__init__(*layers)
        self._prepared: Dict[int, Tuple[Tuple[int, int], int, object]] = {}     # layer index -> (stamp, handle, destroyer)

    def __del__(self):
        try:
            for _, h, destroy in self.__dict__.get("_prepared", {}).values():
                destroy(ctypes.c_void_p(h))
        except Exception:
            pass

    def _weight(self, i: int, conv: nn.Conv2d) -> ctypes.c_void_p:
        from .healpix import _check
        w = conv.weight
        stamp = (w.data_ptr(), w._version)
        cur = self._prepared.get(i)
        if cur is None or cur[0] != stamp:
            h = ctypes.c_void_p()
            t = w.detach().reshape(w.shape[0], w.shape[1]).contiguous().float()
            _check(_lib.lib().ace_hpx_weight_create(t.data_ptr(), t.shape[0], t.shape[1], _lib.current_stream(), ctypes.byref(h)))
            if cur is not None:
                cur[2](ctypes.c_void_p(cur[1]))
            self._prepared[i] = (stamp, h.value, _lib.lib().ace_hpx_weight_destroy)
        return ctypes.c_void_p(self._prepared[i][1])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4:
            raise ValueError(f"expected (batch, channels, rows, columns), got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("the MLP (ace_amd) runs on an MI355X only: move the module and its input to 'cuda'. There is no CPU "
                               "fallback.")
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError("ace_amd implements the inference forward only; call under torch.no_grad()")
        return self._run(x)

    def _run(self, x: torch.Tensor) -> torch.Tensor:
        from .healpix import ACT_GELU, ACT_NONE, _INF, _check
        B, C, H, W = x.shape
        pitch = _round4(W)
        dev = x.device
        if pitch != W:
            cur = torch.zeros(B, C, H, pitch, dtype=torch.float32, device=dev)
            cur[..., :W] = x
        else:
            cur = x.float().contiguous()
        L = _lib.lib()
        st = _lib.current_stream()
        slots = torch.zeros(64 * (len(self) + 1), dtype=torch.int32, device=dev)       # bound slots: the input's, then one per layer
        _check(L.ace_hpx_absmax(cur.data_ptr(), cur.numel(), slots.data_ptr(), st))
        k = 0
        mods = list(self)
        keep = []                                  # the bias tensors handed over by pointer
        for i, m in enumerate(mods):
            if not isinstance(m, nn.Conv2d):
                continue
            act = ACT_GELU if i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU) else ACT_NONE
            y = torch.empty(B, m.out_channels, H, pitch, dtype=torch.float32, device=dev)
            bias = m.bias.detach().float().contiguous() if m.bias is not None else None
            keep.append(bias)
            _check(L.ace_hpx_conv(cur.data_ptr(), None, m.in_channels, 0, self._weight(i, m), None,
                                  bias.data_ptr() if bias is not None else None, None, y.data_ptr(), B,
                                  m.out_channels, H, W, pitch, 1, 1, act, _INF, slots[64 * k:].data_ptr(), None,
                                  slots[64 * (k + 1):].data_ptr(), st))
            cur, k = y, k + 1
        return cur[..., :W].contiguous() if pitch != W else cur


@ModuleSelector.register("MLP")
@dataclasses.dataclass
class MLPConfig(ModuleConfig):
    """mlp.py:9-37 (same type string, same fields)."""
    hidden_dim: int = 256
    depth: int = 2

    def build(self, n_in_channels: int, n_out_channels: int, dataset_info) -> nn.Module:
        return ColumnMLP(in_dim=n_in_channels, out_dim=n_out_channels, n_hidden=self.hidden_dim, depth=self.depth)
