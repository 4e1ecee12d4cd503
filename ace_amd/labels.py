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
        """Columns re-ordered to the encoding's names; names the batch does not have become zero columns, names the encoding
        does not have are dropped (with a warning)."""
        if len(self.names) == 0:
            return BatchLabels(torch.zeros((self.tensor.shape[0], len(encoding.names)), device=self.tensor.device), names=encoding.names)
        old_index = {name: i for i, name in enumerate(self.names)}
        new_names = encoding.names
        idx = torch.tensor([old_index.get(name, -1) for name in new_names], dtype=torch.long, device=self.tensor.device)
        new_mask = idx == -1
        safe_idx = idx.clone()
        safe_idx[new_mask] = 0
        gathered = self.tensor[:, safe_idx]
        if bool(new_mask.any()):
            gathered[:, new_mask] = 0
        dropped = self._names_set.difference(new_names)
        if dropped:
            logging.warning(f"Dropping labels not present in new encoding: {dropped}")
        return BatchLabels(gathered, new_names)

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
        rows = []
        for batch_labels in labels:
            if not batch_labels.issubset(self.names):
                raise InvalidLabelError(f"Invalid labels: at least one of {batch_labels} is not in {self.names}")
            rows.append([1 if label in batch_labels else 0 for label in self.names])
        return BatchLabels(tensor=torch.tensor(rows, dtype=torch.float32, device=device), names=self.names)

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
