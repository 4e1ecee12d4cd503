"""ace_amd/labels.py (BatchLabels, LabelEncoding: the label plumbing of conditional modules) against results emitted by the
reference's own fme/core/labels.py (tests/golden/make_golden_labels.py)."""
import os

import pytest
import torch

from ace_amd.labels import BatchLabels, InvalidLabelError, LabelEncoding

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gen_labels.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


def test_conform_to_encoding_matches_reference(gold):
    for c in gold["conform"]:
        r = BatchLabels(c["tensor"].clone(), list(c["names"])).conform_to_encoding(LabelEncoding(list(c["encoding"])))
        assert r.names == c["out_names"]
        assert torch.equal(r.tensor, c["out"])


def test_encode_and_state_match_reference(gold):
    e = gold["encode"][0]
    enc = LabelEncoding(list(e["encoding"]))
    got = enc.encode([set(s) for s in e["sets"]], torch.device("cpu"))
    assert got.names == e["encoding"] and torch.equal(got.tensor, e["out"])
    with pytest.raises(InvalidLabelError):
        enc.encode([{"not-a-label"}], torch.device("cpu"))
    for s in gold["state"]:
        mine = LabelEncoding(list(s["mine"]))
        mine.conform_to_state({"labels": list(s["state"])})
        assert mine.names == s["names"]
        assert LabelEncoding(list(s["mine"])).append_missing_labels(list(s["state"]) + ["zz"]).names == s["appended"]
        assert LabelEncoding.from_state({"labels": list(s["state"])}).get_state() == {"labels": list(s["state"])}
    with pytest.raises(ValueError):
        LabelEncoding("abc")
    with pytest.raises(ValueError):
        BatchLabels(torch.zeros(2, 3), ["a"])
