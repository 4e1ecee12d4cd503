"""Goldens of scale_factor != 1 (fme/ace/models/modulus/sfnonet.py:467-515: the blocks between the first filter's inverse transform and
the last filter's forward transform work on the (H // sf) x (W // sf) Gauss-Legendre grid; the first and the last block change grids
and their residual is the spectrally round-tripped input, s2convolutions.py:165-172), produced HERE by the real reference net imported
from /root/reference under stubs (oracle/ref_loader.py).  Build container only:  python tests/golden/make_golden_scale_factor.py

Pure data: configurations, the seeds the tests regenerate weights / inputs from (oracle.sfno.init_state; checksums guard against RNG
drift) and the reference's outputs.
"""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import checksum, ref_net_from_cfg  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle.sfno import SFNOConfig, init_state  # noqa: E402


def main():
    ns = ref_loader.load()
    base = dict(in_chans=5, out_chans=7, embed_dim=16, operator_type="dhconv")
    cases = {
        "sf2_3blocks": (SFNOConfig(img_shape=(12, 24), num_layers=3, scale_factor=2, **base), 2, 61),
        "sf2_2blocks": (SFNOConfig(img_shape=(12, 24), num_layers=2, scale_factor=2, **base), 2, 62),
        "sf2_1block": (SFNOConfig(img_shape=(12, 24), num_layers=1, scale_factor=2, **base), 1, 63),
        "sf3_equiangular": (SFNOConfig(img_shape=(9, 18), num_layers=3, scale_factor=3, data_grid="equiangular", **base), 3, 64),
        "sf2_layer_norm": (SFNOConfig(img_shape=(12, 24), num_layers=3, scale_factor=2, normalization_layer="layer_norm", **base), 2, 65),
        "sf2_diagonal_no_norm": (SFNOConfig(img_shape=(12, 24), num_layers=3, scale_factor=2, in_chans=5, out_chans=7, embed_dim=16,
                                            operator_type="diagonal", normalization_layer="none"), 2, 66),
    }
    out = {}
    for name, (cfg, batch, seed) in cases.items():
        state = init_state(cfg, seed=seed)
        x = torch.randn(batch, cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(seed + 1000))
        net = ref_net_from_cfg(ns, cfg, state)
        with torch.no_grad():
            y = net(x)
        out[name] = {"cfg": dataclasses.asdict(cfg), "seed": seed, "batch": batch,
                     "state_checksum": sum(checksum(v) for v in state.values()), "x_checksum": checksum(x), "y": y}
        print(name, tuple(y.shape), float(y.abs().max()))
    torch.save(out, os.path.join(HERE, "gen_sfno_scale_factor.pt"))


if __name__ == "__main__":
    main()
