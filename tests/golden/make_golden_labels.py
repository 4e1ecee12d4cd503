"""Golden vectors for the label plumbing (fme/core/labels.py: BatchLabels.conform_to_encoding, LabelEncoding.encode /
conform_to_state / append_missing_labels), emitted by the REAL reference module - build container only."""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader  # noqa: E402


def main():
    ref_loader.load()
    lab = importlib.import_module("fme.core.labels")
    out = {"conform": [], "encode": [], "state": []}
    g = torch.Generator().manual_seed(0)
    for names, enc in [(["b", "a", "c"], ["a", "b", "c"]), (["a", "b"], ["b", "x", "a"]), ([], ["p", "q"]), (["z", "a"], ["a"])]:
        t = torch.rand(3, len(names), generator=g)
        r = lab.BatchLabels(t.clone(), list(names)).conform_to_encoding(lab.LabelEncoding(list(enc)))
        out["conform"].append({"names": names, "tensor": t, "encoding": enc, "out_names": list(r.names), "out": r.tensor.clone()})
    e = lab.LabelEncoding(["era5", "shield", "cm4"])
    sets = [{"era5"}, {"cm4", "shield"}, set()]
    out["encode"].append({"encoding": e.names, "sets": [sorted(s) for s in sets], "out": e.encode(sets, torch.device("cpu")).tensor.clone()})
    for mine, state in [(["a", "b", "c"], ["c", "a"]), (["a"], ["a", "b"]), (["x", "y"], ["y", "x"])]:
        enc = lab.LabelEncoding(list(mine))
        enc.conform_to_state({"labels": list(state)})
        out["state"].append({"mine": mine, "state": state, "names": list(enc.names),
                             "appended": list(lab.LabelEncoding(list(mine)).append_missing_labels(list(state) + ["zz"]).names)})
    dst = os.path.join(HERE, "gen_labels.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
