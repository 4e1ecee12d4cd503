"""The slice of DatasetInfo the hot path reads (fme/core/dataset_info.py:151-163, 225-229):
`img_shape`, `all_labels` and `timestep`.  The reference object (duck-typed: anything
with these attributes) can be passed instead."""

import datetime
from typing import Optional, Set, Tuple


class DatasetInfo:
    def __init__(self, img_shape: Tuple[int, int], all_labels: Optional[Set[str]] = None,
                 timestep: datetime.timedelta = datetime.timedelta(hours=6)):
        self._img_shape = (int(img_shape[-2]), int(img_shape[-1]))
        self._all_labels = set(all_labels) if all_labels else set()
        self._timestep = timestep

    @property
    def img_shape(self) -> Tuple[int, int]:
        return self._img_shape

    @property
    def all_labels(self) -> Set[str]:
        return self._all_labels

    @property
    def timestep(self) -> datetime.timedelta:
        return self._timestep

    def __repr__(self):
        return f"DatasetInfo(img_shape={self._img_shape}, all_labels={self._all_labels}, timestep={self._timestep})"
