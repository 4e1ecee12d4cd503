"""Label conditioning plumbing of the module registry (fme/core/labels.py): ``BatchLabels`` (a (batch, n_labels) tensor with
column names) and ``LabelEncoding`` (the ordered label names a conditional module was built with; one-hot encoding of label
sets; state as stored in checkpoints).  Plain torch on whatever device the tensors live on."""
import logging
from typing import Any, Dict, List, Set

import torch


class BatchLabels:
    """labels.py:7-106."""

    def __init__(self, tensor: torch.Tensor, names: List[str]):
        self.tensor = tensor
        self.names = names
        if len(names) != tensor.shape[1]:
            raise ValueError(f"Number of names ({len(names)}) must match number of columns in tensor ({tensor.shape[1]}).")
        self._names_set = set(names)

    def to(self, device) -> "BatchLabels":
        return BatchLabels(self.tensor.to(device), self.names)

    def __repr__(self) -> str:
        return f"BatchLabels(names={self.names}, tensor={self.tensor})"

    def conform_to_encoding(self, encoding: "LabelEncoding") -> "BatchLabels":
        """A new BatchLabels in the encoding's column order: a column this batch holds is copied, a name it does not hold becomes a
        zero column, a name the encoding does not know is dropped (logged) - labels.py:35-89."""
        target = encoding.names
        n = self.tensor.shape[0]
        if not self.names:
            return BatchLabels(torch.zeros((n, len(target)), device=self.tensor.device), names=target)
        column = {name: j for j, name in enumerate(self.names)}
        out = self.tensor.new_zeros((n, len(target)))
        held = [(j, column[name]) for j, name in enumerate(target) if name in column]
        if held:
            dst, src = zip(*held)
            out[:, list(dst)] = self.tensor[:, list(src)]
        unknown = self._names_set.difference(target)
        if unknown:
            logging.warning(f"Dropping labels not present in new encoding: {unknown}")
        return BatchLabels(out, target)

    def __eq__(self, other: Any) -> bool:
        return isinstance(other, BatchLabels) and self.names == other.names and torch.equal(self.tensor, other.tensor)

    @classmethod
    def new_from_set(cls, label_set: Set[str], n_samples: int, device) -> "BatchLabels":
        names = sorted(list(label_set))
        return cls(tensor=torch.ones((n_samples, len(names)), dtype=torch.float32).to(device), names=names)


class InvalidLabelError(ValueError):
    pass


class LabelEncoding:
    """labels.py:117-190."""

    def __init__(self, labels: List[str]):
        if not isinstance(labels, list):
            raise ValueError("Labels must be an ordered list of strings")
        self.names = labels.copy()

    def encode(self, labels: List[Set[str]], device) -> BatchLabels:
        """one row per batch member: 1 where the member carries the encoding's label (labels.py:131-160)"""
        known = set(self.names)
        for member in labels:
            if not member <= known:
                raise InvalidLabelError(f"Invalid labels: at least one of {member} is not in {self.names}")
        onehot = torch.tensor([[float(name in member) for name in self.names] for member in labels], dtype=torch.float32, device=device)
        return BatchLabels(tensor=onehot.reshape(len(labels), len(self.names)), names=self.names)

    def get_state(self) -> Dict[str, Any]:
        return {"labels": self.names}

    @classmethod
    def from_state(cls, state: Dict[str, Any]) -> "LabelEncoding":
        encoder = cls(list(state["labels"]))
        encoder.conform_to_state(state)
        return encoder

    def append_missing_labels(self, labels: List[str]) -> "LabelEncoding":
        missing = set(labels).difference(self.names)
        return LabelEncoding(self.names + sorted(list(missing))) if missing else self

    def conform_to_state(self, state: Dict[str, Any]) -> None:
        """The loaded weights need the state's labels first, in the state's order; labels only this encoding has follow."""
        state_labels = list(state["labels"])
        additional = set(self.names).difference(state_labels)
        self.names = state_labels + sorted(list(additional))
