"""The slice of DatasetInfo the hot path reads (fme/core/dataset_info.py:151-163, 225-229):
`img_shape`, `all_labels` and `timestep` (plus the coordinates the post-step physics and the derived forcings read).  The reference object (duck-typed: anything
with these attributes) can be passed instead."""

import datetime
from typing import Optional, Set, Tuple


class DatasetInfo:
    def __init__(self, img_shape: Tuple[int, int], all_labels: Optional[Set[str]] = None,
                 timestep: datetime.timedelta = datetime.timedelta(hours=6), lat=None, lon=None, ak=None, bk=None,
                 area_weights=None, mask_provider=None):
        """``lat``/``lon`` (degrees; LatLonCoordinates, fme/core/coordinates.py:608-709) give the area weights of the
        conservation correctors, ``ak``/``bk`` the hybrid sigma-pressure interfaces
        (HybridSigmaPressureCoordinate, coordinates.py:150-280); all optional."""
        self._img_shape = (int(img_shape[-2]), int(img_shape[-1]))
        self._all_labels = set(all_labels) if all_labels else set()
        self._timestep = timestep
        self._area_weights = None
        self._vertical_coordinate = None
        self._horizontal_coordinates = None
        self._mask_provider = mask_provider          # ace_amd.masking.SpatialMaskProvider or None
        if lat is not None and lon is not None:
            from .insolation import LatLonGrid
            import torch
            self._horizontal_coordinates = LatLonGrid(torch.as_tensor(lat).detach().cpu(), torch.as_tensor(lon).detach().cpu())
        if area_weights is not None:
            import torch
            self._area_weights = torch.as_tensor(area_weights).detach().cpu()
        elif lat is not None:
            import torch
            from .atmosphere import spherical_area_weights
            lat = torch.as_tensor(lat).detach().cpu()
            if len(lat) != self._img_shape[0]:
                raise ValueError(f"{len(lat)} latitudes for an image of {self._img_shape[0]} rows")
            nlon = len(lon) if lon is not None else self._img_shape[1]
            self._area_weights = spherical_area_weights(lat, nlon)
        if ak is not None and bk is not None:
            import torch
            from .atmosphere import HybridSigmaPressureCoordinate
            self._vertical_coordinate = HybridSigmaPressureCoordinate(torch.as_tensor(ak).detach().cpu(),
                                                                      torch.as_tensor(bk).detach().cpu())

    @property
    def area_weights(self):
        """(nlat, nlon) weights summing to 1, or None (fme/core/metrics.py:14-32)."""
        return self._area_weights

    @property
    def horizontal_coordinates(self):
        """1-D lat / lon in degrees with a ``meshgrid`` (what the derived insolation reads), or None."""
        return self._horizontal_coordinates

    @property
    def mask_provider(self):
        """the dataset's static masks (fme/core/spatial_mask_provider.py), or None"""
        return self._mask_provider

    @property
    def vertical_coordinate(self):
        return self._vertical_coordinate

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
