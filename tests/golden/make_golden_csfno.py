"""Golden vectors for the NoiseConditionedSFNO oracle, emitted by the REAL reference module imported under stubs
(oracle/ref_loader.load_csfno) - build container only.  Small configurations of the family ACE ships today; each case
stores the builder kwargs, the reference state_dict (random init + randomised conditioning weights, which the
reference initialises to zero), the input, the RNG seed of the forward call and the reference output."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402

CASES = {
    "isotropic_affine_bigskipnorm": dict(embed_dim=16, noise_embed_dim=8, noise_type="isotropic", num_layers=2,
                                         affine_norms=True, normalize_big_skip=True),
    "gaussian_groups2": dict(embed_dim=16, noise_embed_dim=4, noise_type="gaussian", num_layers=3, filter_num_groups=2,
                             activation_function="silu", encoder_layers=2),
    "equiangular_nomlp": dict(embed_dim=8, noise_embed_dim=8, noise_type="isotropic", num_layers=2, use_mlp=False,
                              data_grid="equiangular", big_skip=False, pos_embed=False),
}


# label / positional context (stochastic_sfno.py:88-175, conditional_sfno/layers.py:160-318): written to their own file so that
# gen_csfno.pt stays byte-identical
CONTEXT_CASES = {
    "labels3_pos4": (dict(embed_dim=16, noise_embed_dim=4, noise_type="gaussian", num_layers=2, affine_norms=True,
                          normalize_big_skip=True, pos_embed=False, context_pos_embed_dim=4), ["a", "b", "c"]),
    "labels3_embed2_pos2_isotropic": (dict(embed_dim=8, noise_embed_dim=4, noise_type="isotropic", num_layers=2, pos_embed=False,
                                           context_pos_embed_dim=2, label_embed_dim=2), ["a", "b", "c"]),
    "labels2_nopos": (dict(embed_dim=8, noise_embed_dim=4, noise_type="gaussian", num_layers=1), ["x", "y"]),
}


def context_cases(ref):
    out = {}
    for name, (kw, all_labels) in CONTEXT_CASES.items():
        torch.manual_seed(0)
        info = ref.Info((12, 24))
        info.all_labels = set(all_labels)
        model = ref.Builder(**kw).build(5, 4, info)
        g = torch.Generator().manual_seed(2)
        with torch.no_grad():      # every conditioning weight starts at 0 (layers.py:222-243): make them matter
            for k, p in model.named_parameters():
                if any(t in k for t in ("W_scale_2d", "W_bias_2d", "W_scale_pos", "W_bias_pos", "W_scale_labels", "W_bias_labels")):
                    p.copy_(0.3 * torch.randn(p.shape, generator=g))
                if ".norm.weight" in k:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                if ".norm.bias" in k or k.endswith("filter.filter.bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                if k in ("pos_embed", "label_pos_embed"):
                    p.copy_(0.5 * torch.randn(p.shape, generator=g))
        model.eval()
        x = torch.randn(3, 5, 12, 24, generator=g)
        labels = torch.zeros(3, len(all_labels))
        labels[0, 0] = 1.0
        labels[1, -1] = 1.0
        labels[2] = torch.tensor([0.25, 0.75, 0.0][: len(all_labels)])       # a soft encoding
        torch.manual_seed(4321)
        with torch.no_grad():
            y = model(x, labels=labels)
        out[name] = {"kwargs": kw, "all_labels": sorted(all_labels), "state": {k: v.clone() for k, v in model.state_dict().items()},
                     "x": x, "labels": labels, "forward_seed": 4321, "y": y}
        print(name, tuple(y.shape), float(y.abs().max()), len(out[name]["state"]), "tensors")
    path = os.path.join(HERE, "gen_csfno_context.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    ref = ref_loader.load_csfno()
    context_cases(ref)
    out = {}
    for name, kw in CASES.items():
        torch.manual_seed(0)
        model = ref.Builder(**kw).build(5, 4, ref.Info((12, 24)))
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():      # the conditioning weights and affine norms start at 0 / 1: make them matter
            for k, p in model.named_parameters():
                if "W_scale_2d" in k or "W_bias_2d" in k:
                    p.copy_(0.3 * torch.randn(p.shape, generator=g))
                if ".norm.weight" in k:
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                if ".norm.bias" in k or k.endswith("filter.filter.bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
        model.eval()
        x = torch.randn(2, 5, 12, 24, generator=g)
        torch.manual_seed(1234)
        with torch.no_grad():
            y = model(x)
        out[name] = {"kwargs": kw, "state": {k: v.clone() for k, v in model.state_dict().items()}, "x": x,
                     "forward_seed": 1234, "y": y}
        print(name, tuple(y.shape), float(y.abs().max()), len(out[name]["state"]), "tensors")
    path = os.path.join(HERE, "gen_csfno.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
