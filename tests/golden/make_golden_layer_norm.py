"""Golden of the reference's "layer_norm" normalisation (fme/ace/models/modulus/sfnonet.py:584-592: nn.LayerNorm over (H, W) with an
(H, W) elementwise affine in front of the filter and of the MLP of every block), produced HERE by the real reference net imported
from /root/reference under stubs (oracle/ref_loader.py).  Build container only:  python tests/golden/make_golden_layer_norm.py

Pure data: the configuration, the seeds the tests regenerate weights / input from (oracle.sfno.init_state; checksums guard against RNG
drift) and the reference's output.
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
    cases = {
        "gen_sfno_layer_norm_12x24.pt": (SFNOConfig(in_chans=5, out_chans=7, img_shape=(12, 24), embed_dim=16, num_layers=3,
                                                     operator_type="dhconv", normalization_layer="layer_norm"), 2, 31),
        "gen_sfno_layer_norm_equiangular_9x18.pt": (SFNOConfig(in_chans=2, out_chans=3, img_shape=(9, 18), embed_dim=8, num_layers=2,
                                                                operator_type="dhconv", data_grid="equiangular",
                                                                normalization_layer="layer_norm"), 3, 32),
    }
    for fname, (cfg, batch, seed) in cases.items():
        state = init_state(cfg, seed=seed)
        x = torch.randn(batch, cfg.in_chans, *cfg.img_shape, generator=torch.Generator().manual_seed(seed + 1000))
        net = ref_net_from_cfg(ns, cfg, state)
        with torch.no_grad():
            y = net(x)
        torch.save({"cfg": dataclasses.asdict(cfg), "seed": seed, "batch": batch,
                    "state_checksum": sum(checksum(v) for v in state.values()), "x_checksum": checksum(x), "y": y},
                   os.path.join(HERE, fname))
        print(fname, tuple(y.shape), float(y.abs().max()))


if __name__ == "__main__":
    main()
