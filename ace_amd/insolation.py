"""Top-of-atmosphere insolation derived from the time axis (the reference's only derived forcing:
fme/ace/stepper/insolation/config.py:59-175, cm4.py:216-509 - GFDL CM4 / FMS astronomy): the mean over the model timestep ENDING
at each time level of  S0 * (a / r)^2 * max(cos zenith, 0), per grid cell.

The algorithm is that of the astronomy / time_manager modules of GFDL's Flexible Modeling System (FMS, Apache License 2.0), which the
reference's cm4.py is derived from; nothing of either is copied here - the formulas are re-derived from the geometry below.

Where it sits: once per forcing window, in front of the rollout (``Stepper.predict`` / ``EnginePredict`` / ``ForcingWindows``), on
the device the forcings live on - (samples, T + 1, lat, lon) cells of elementwise arithmetic, a few ATen launches per window, never
inside the per-step graphs.  Times come as an ``ace_amd.timeaxis.TimeAxis`` (no cftime / xarray on the path).

The geometry is restated from the astronomy, not from the reference's case table:
  * orbital position: the year since the autumnal equinox of 1998-09-23T05:37 (calendar-specific year length, cm4.py:196-214),
    mapped to the true anomaly through a 3600-entry table integrated with RK4 from Kepler's second law (dtheta / dtau =
    sqrt(1 - e^2) (a / r)^2 over mean time tau).  As in the reference the table is built in fp32 and its first entry is the angle
    after ONE step, so the looked-up angle runs 1/3600 of a year ahead of FMS's; kept, because parity is with the reference.
    (At the last 1/3600 of the orbital year the reference indexes one past its table - an IndexError there; here the
    recurrence simply continues one more step.)
  * declination from the obliquity, half-day length h from  cos h = -tan(lat) tan(decl)  (polar day h = pi, polar night h = 0);
  * the timestep is the hour-angle interval [t, t + dt], dt < pi; daylight is |hour angle| <= h, i.e. the window [-h, h] and
    its image one revolution later; the mean of  sin(lat) sin(decl) + cos(lat) cos(decl) cos(hour angle)  over the daylight
    part of the interval has the closed form  aa + bb (sum of sine differences) / (daylight length), and the daylight fraction
    is (daylight length) / dt.  (The reference walks through FMS's eight branch cases; they are this overlap, case by case.)"""
import dataclasses
import datetime
import math
import warnings
from typing import Any, Dict, List, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from .timeaxis import TimeAxis, US_PER_DAY, as_time_axis, _timedelta_us

TensorMapping = Mapping[str, torch.Tensor]

AUTUMNAL_EQUINOX = (1998, 9, 23, 5, 37, 0)
NUM_ANGLES = 3600
# FMS time_manager year lengths (cm4.py:196-214), microseconds
_YEAR_US = {
    "noleap": 365 * US_PER_DAY,
    "standard": 365 * US_PER_DAY + 20952 * 1_000_000,
    "proleptic_gregorian": 365 * US_PER_DAY + 20952 * 1_000_000,
    "julian": 365 * US_PER_DAY + 21600 * 1_000_000,
    "360_day": 360 * US_PER_DAY,
    "all_leap": 366 * US_PER_DAY,
}
MAXIMUM_TIMESTEP = datetime.timedelta(hours=12)     # FMS averages over less than half a day (cm4.py:216-218)
_TWO_PI = 2.0 * math.pi


@dataclasses.dataclass
class NameConfig:
    """Solar constant read from the forcing data (possibly time-varying); the insolation takes its dtype (config.py:14-29)."""
    name: str

    def get(self, tensors: TensorMapping) -> torch.Tensor:
        return tensors[self.name]


@dataclasses.dataclass
class ValueConfig:
    """One solar constant for all time (config.py:32-56)."""
    value: float
    dtype: str = "float32"

    @property
    def torch_dtype(self) -> torch.dtype:
        dt = getattr(torch, self.dtype, None)
        if not isinstance(dt, torch.dtype):
            raise ValueError(f"Invalid dtype '{self.dtype}'")
        return dt

    def get(self, tensors: TensorMapping) -> torch.Tensor:
        return torch.tensor(self.value, dtype=self.torch_dtype)


def _inverse_square_distance(angle: torch.Tensor, ecc: torch.Tensor, perihelion_deg: torch.Tensor) -> torch.Tensor:
    """(a / r)^2 on the Kepler ellipse r / a = (1 - e^2) / (1 + e cos(angle - perihelion))."""
    r = (1 - ecc ** 2) / (1 + ecc * torch.cos(angle - torch.deg2rad(perihelion_deg)))
    return r ** (-2)


_TABLES: Dict[Tuple[float, float], torch.Tensor] = {}


def orbital_angle_table(ecc: torch.Tensor, perihelion_deg: torch.Tensor, device=None) -> torch.Tensor:
    """theta[k], k = 0 .. NUM_ANGLES + 1: orbital angle after k steps of mean time 2 pi / NUM_ANGLES from the equinox (theta[0] = 0),
    RK4 on  dtheta = sqrt(1 - e^2) (a / r)^2 dtau  in fp32 (the reference's arithmetic, cm4.py:326-353).  Built once per orbit
    (3601 sequential steps on the host, ~1 s) and kept."""
    ecc = ecc.to(torch.float32).cpu()
    per = perihelion_deg.to(torch.float32).cpu()
    key = (float(ecc), float(per))
    if key not in _TABLES:
        _TABLES[key] = _integrate_orbit(ecc, per)
    theta = _TABLES[key]
    return theta.to(device) if device is not None else theta


def _integrate_orbit(ecc: torch.Tensor, per: torch.Tensor) -> torch.Tensor:
    step = (2 * torch.pi / NUM_ANGLES) * torch.sqrt(1 - ecc ** 2)
    theta = torch.zeros(NUM_ANGLES + 2, dtype=torch.float32)
    cur = theta[0]
    for k in range(1, NUM_ANGLES + 2):
        k1 = step * _inverse_square_distance(cur, ecc, per)
        k2 = step * _inverse_square_distance(cur + 0.5 * k1, ecc, per)
        k3 = step * _inverse_square_distance(cur + 0.5 * k2, ecc, per)
        k4 = step * _inverse_square_distance(cur + k3, ecc, per)
        cur = cur + (k1 / 6.0 + k2 / 3.0 + k3 / 3.0 + k4 / 6.0)
        theta[k] = cur
    return theta


class CM4Insolation:
    """cm4.py:216-243: callable (time, timestep, lat, lon, solar_constant) -> insolation."""

    def __init__(self, obliquity: float, eccentricity: float, longitude_of_perhelion: float):
        self.obliquity = torch.as_tensor(obliquity)
        self.eccentricity = torch.as_tensor(eccentricity)
        self.longitude_of_perhelion = torch.as_tensor(longitude_of_perhelion)
        self._table = orbital_angle_table(self.eccentricity, self.longitude_of_perhelion)
        self._table_on: Dict[str, torch.Tensor] = {}
        self._radians: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]] = {}

    def table(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._table_on:
            self._table_on[key] = self._table.to(device)
        return self._table_on[key]

    def geometry(self, time: TimeAxis, timestep: datetime.timedelta, lat_deg: torch.Tensor, lon_deg: torch.Tensor
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (mean cos zenith over the daylight part, daylight fraction, (a / r)^2), each time.shape + lat.shape.
        lat_deg / lon_deg: the grid's coordinates in degrees, fully broadcast over the horizontal dims (a meshgrid)."""
        if timestep >= MAXIMUM_TIMESTEP:
            raise NotImplementedError(f"Computing insolation via the CM4 implementation is not implemented for a model timestep "
                                      f"greater than or equal to 12 hours. Timestep is {timestep!r}.")
        dev, dtype = lat_deg.device, lat_deg.dtype
        # the units check reads two ranges back to the host: once per grid, not once per window
        key = (lat_deg.data_ptr(), lon_deg.data_ptr())
        hit = self._radians.get(key)
        if hit is None or hit[0] is not lat_deg or hit[1] is not lon_deg:
            hit = self._radians[key] = (lat_deg, lon_deg) + degrees_to_radians(lat_deg, lon_deg)
        lat, lon = hit[2], hit[3]
        begin = time - timestep                                     # the averaging interval ENDS at the given time
        tshape = begin.shape
        expand = tshape + (1,) * lat.ndim

        def on_device(a: np.ndarray) -> torch.Tensor:
            return torch.as_tensor(a.astype(np.float64), device=dev).to(dtype).reshape(expand)

        # time of day as an angle and position in the orbital year as an angle, both exact integer arithmetic up to the last division
        day_angle = on_device(_TWO_PI * (begin.microseconds_of_day() / US_PER_DAY))
        year = _YEAR_US[begin.calendar]
        since = begin.microseconds_since(AUTUMNAL_EQUINOX)
        orbital_time = on_device(_TWO_PI * ((since % year) / year))

        # table lookup (linear interpolation); entry n of the reference's table is theta[n + 1]
        theta = self.table(dev)
        pos = orbital_time * NUM_ANGLES / _TWO_PI
        whole = torch.floor(pos)
        n = whole.to(torch.int64) % NUM_ANGLES
        frac = pos - whole
        angle = ((1.0 - frac) * theta[n + 1] + frac * theta[n + 2]) % _TWO_PI

        obliq = torch.deg2rad(self.obliquity.to(dev))
        decl = torch.arcsin(-torch.sin(obliq) * torch.sin(angle))
        rr = _inverse_square_distance(angle, self.eccentricity.to(dev), self.longitude_of_perhelion.to(dev))

        aa = torch.sin(lat) * torch.sin(decl)
        bb = torch.cos(lat) * torch.cos(decl)
        h = half_day(lat, decl)

        t = day_angle + lon - torch.pi                              # local hour angle at the start of the interval, in [-pi, pi)
        t = torch.where(t >= torch.pi, t - _TWO_PI, t)
        t = torch.where(t < -torch.pi, t + _TWO_PI, t)
        dt = _TWO_PI * (_timedelta_us(timestep) / US_PER_DAY)
        tt = t + dt

        # overlap of [t, tt] with the daylight windows [-h, h] and [2 pi - h, 2 pi + h]  (tt < 2 pi + h always: dt < pi)
        sin_t, sin_tt, sin_h = torch.sin(t), torch.sin(tt), torch.sin(h)
        lo1, hi1 = torch.maximum(t, -h), torch.minimum(tt, h)
        len1 = torch.clamp(hi1 - lo1, min=0.0)
        sines1 = torch.where(tt < h, sin_tt, sin_h) - torch.where(t > -h, sin_t, -sin_h)
        len2 = torch.clamp(tt - (_TWO_PI - h), min=0.0)               # (t < pi <= 2 pi - h: the interval never STARTS in the second window)
        sines2 = sin_tt + sin_h
        zero = torch.zeros((), dtype=dtype, device=dev)
        sines = torch.where(len1 > 0, sines1, zero) + torch.where(len2 > 0, sines2, zero)
        daylight = len1 + len2
        cosz = torch.where(daylight > 0, aa + bb * sines / torch.where(daylight > 0, daylight, torch.ones_like(daylight)), zero)
        cosz = torch.clamp(cosz, min=0.0)
        return cosz, daylight / dt, rr

    def __call__(self, time: TimeAxis, timestep: datetime.timedelta, lat_deg: torch.Tensor, lon_deg: torch.Tensor,
                 solar_constant: torch.Tensor) -> torch.Tensor:
        cosz, fracday, rr = self.geometry(time, timestep, lat_deg, lon_deg)
        s0 = solar_constant.to(lat_deg.device)
        return (s0 * rr * fracday * cosz).to(s0.dtype)


def degrees_to_radians(lat: torch.Tensor, lon: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """cm4.py:246-260: coordinates are taken to be degrees; warn when their ranges look like radians."""
    lat_range = lat.max() - lat.min()
    lon_range = lon.max() - lon.min()
    if lat_range < torch.pi or lon_range < 2 * torch.pi:
        warnings.warn(f"Range of latitude and/or longitude coordinates is smaller than expected for units of degrees. Latitude range "
                      f"= {lat_range:0.2f}; longitude range = {lon_range:0.2f}. Computing insolation assumes latitude and longitude "
                      f"start out in units of degrees instead of radians.")
    return torch.deg2rad(lat), torch.deg2rad(lon)


def half_day(lat: torch.Tensor, decl: torch.Tensor) -> torch.Tensor:
    """Hour angle of sunset: arccos(-tan lat tan decl), pi where the sun never sets, 0 where it never rises; the poles themselves
    are nudged by 1e-5 rad as FMS does (cm4.py:372-383)."""
    half_pi = 0.5 * torch.pi
    lat = torch.where(lat == half_pi, lat - 1.0e-5, lat)
    lat = torch.where(lat == -half_pi, lat + 1.0e-5, lat)
    c = -torch.tan(lat) * torch.tan(decl)
    inside = (c > -1.0) & (c < 1.0)
    h = torch.where(inside, torch.arccos(torch.where(inside, c, torch.zeros_like(c))), torch.zeros_like(c))
    return torch.where(c <= -1.0, torch.full_like(c, torch.pi), h)


@dataclasses.dataclass
class InsolationConfig:
    """config.py:59-141 (same fields, same defaults)."""
    insolation_name: str
    solar_constant: Union[NameConfig, ValueConfig]
    obliquity: float = 23.439
    eccentricity: float = 0.0167
    longitude_of_perhelion: float = 102.932

    def __post_init__(self):
        sc = self.solar_constant
        if isinstance(sc, Mapping):           # serialised form: the reference's dacite Union resolves on the field names
            sc = dict(sc)
            if "name" in sc and "value" not in sc:
                self.solar_constant = NameConfig(**sc)
            elif "value" in sc:
                self.solar_constant = ValueConfig(**sc)
            else:
                raise ValueError(f"solar_constant needs a 'name' or a 'value', got {sorted(sc)}")

    def build(self, timestep: datetime.timedelta, horizontal_coordinates) -> "Insolation":
        return Insolation(self, timestep, horizontal_coordinates)

    def build_insolation_function(self) -> CM4Insolation:
        return CM4Insolation(self.obliquity, self.eccentricity, self.longitude_of_perhelion)

    def update_names(self, names: List[str]) -> List[str]:
        """config.py:124-141 on a plain name list (the reference's DataRequirements.names): the insolation is not read from disk,
        a named solar constant is."""
        names = list(names)
        if self.insolation_name in names:
            names.remove(self.insolation_name)
            if isinstance(self.solar_constant, NameConfig) and self.solar_constant.name not in names:
                names.append(self.solar_constant.name)
        return names


class LatLonGrid:
    """The slice of fme.core.coordinates.LatLonCoordinates this needs: 1-D lat / lon in degrees -> ``meshgrid`` on a device."""

    def __init__(self, lat, lon):
        self.lat = torch.as_tensor(lat)
        self.lon = torch.as_tensor(lon)

    def to(self, device) -> "LatLonGrid":
        return LatLonGrid(self.lat.to(device), self.lon.to(device))

    @property
    def meshgrid(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return torch.meshgrid(self.lat, self.lon, indexing="ij")


class Insolation:
    """config.py:144-175."""

    def __init__(self, config: InsolationConfig, timestep: datetime.timedelta, horizontal_coordinates):
        if horizontal_coordinates is None:
            raise ValueError("computing the insolation needs the grid's latitudes and longitudes (dataset_info.horizontal_coordinates)")
        self.config = config
        self.timestep = timestep
        self.horizontal_coordinates = horizontal_coordinates
        self.insolation_function = config.build_insolation_function()
        self._mesh: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}

    def compute(self, time, tensors: TensorMapping, device=None) -> Dict[str, torch.Tensor]:
        """-> a shallow copy of `tensors` with the insolation (time.shape + grid shape) added under ``insolation_name``.
        device: where to compute (default: the device of the tensors given, else of the coordinates)."""
        time = as_time_axis(time)
        out = dict(tensors)
        s0 = self.config.solar_constant.get(out)
        if device is None:
            device = next((v.device for v in out.values() if isinstance(v, torch.Tensor)), self.horizontal_coordinates.lat.device)
        if str(device) not in self._mesh:
            self._mesh[str(device)] = tuple(self.horizontal_coordinates.to(device).meshgrid)
        lat, lon = self._mesh[str(device)]
        out[self.config.insolation_name] = self.insolation_function(time, self.timestep, lat, lon, s0)
        return out
