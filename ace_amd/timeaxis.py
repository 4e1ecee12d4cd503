"""Times of a forcing record without cftime / xarray (neither is needed on the path, neither is in this image): a calendar name
and an integer count of microseconds since 0001-01-01T00:00 of THAT calendar, any array shape (normally (samples, time levels),
the shape of ``BatchData.time``, fme/ace/data_loading/batch_data.py).  The only consumers are the forcings derived from the time
axis (ace_amd/insolation.py), which need differences of times and the time of day - exactly what cftime's
``time - other`` / ``% timedelta(days=1)`` give the reference (fme/ace/stepper/insolation/cm4.py:296-313).

Calendars are the six the reference knows (cm4.py:196-214): ``noleap`` (365_day), ``all_leap`` (366_day), ``360_day``,
``julian``, ``proleptic_gregorian`` and ``standard`` (gregorian: Julian rules up to 1582-10-04, Gregorian from 1582-10-15).
Objects with year / month / ... attributes (``datetime.datetime``, cftime instances when a caller has them) are accepted and read
attribute by attribute; their ``calendar`` attribute names the calendar."""
import datetime
from typing import Iterable, Optional, Sequence, Union

import numpy as np

US_PER_SECOND = 1_000_000
US_PER_DAY = 86_400 * US_PER_SECOND
_ALIASES = {"gregorian": "standard", "365_day": "noleap", "366_day": "all_leap"}
CALENDARS = ("noleap", "standard", "proleptic_gregorian", "julian", "360_day", "all_leap")
_CUM = np.array([0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334], dtype=np.int64)        # days before month m, common year
_MONTH_DAYS = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31], dtype=np.int64)


def canonical_calendar(name: str) -> str:
    name = _ALIASES.get(str(name).lower(), str(name).lower())
    if name not in CALENDARS:
        raise ValueError(f"unknown calendar {name!r}; expected one of {CALENDARS}")
    return name


def _is_leap(calendar: str, y: np.ndarray, gregorian_rule: Optional[np.ndarray] = None) -> np.ndarray:
    if calendar == "noleap" or calendar == "360_day":
        return np.zeros(y.shape, dtype=bool)
    if calendar == "all_leap":
        return np.ones(y.shape, dtype=bool)
    julian = (y % 4 == 0)
    greg = julian & ((y % 100 != 0) | (y % 400 == 0))
    if calendar == "julian":
        return julian
    if calendar == "proleptic_gregorian":
        return greg
    return np.where(gregorian_rule, greg, julian)           # standard: the rule of the date's own side of the 1582 switch


def days_since_base(calendar: str, year, month, day) -> np.ndarray:
    """Whole days from 0001-01-01 of `calendar` to year-month-day (int64 array, broadcast of the arguments)."""
    calendar = canonical_calendar(calendar)
    y, m, d = np.broadcast_arrays(np.asarray(year, dtype=np.int64), np.asarray(month, dtype=np.int64), np.asarray(day, dtype=np.int64))
    if np.any(y < 1):
        raise ValueError("years before 1 are not supported")
    if np.any((m < 1) | (m > 12)):
        raise ValueError("month out of range")
    if calendar == "360_day":
        if np.any((d < 1) | (d > 30)):
            raise ValueError("day out of range for the 360_day calendar")
        return (y - 1) * 360 + (m - 1) * 30 + (d - 1)
    after = None
    if calendar == "standard":
        key = y * 10000 + m * 100 + d
        if np.any((key > 15821004) & (key < 15821015)):
            raise ValueError("1582-10-05 ... 1582-10-14 do not exist in the standard (mixed Julian / Gregorian) calendar")
        after = key >= 15821015
    leap = _is_leap(calendar, y, after)
    length = _MONTH_DAYS[m - 1] + ((m == 2) & leap)
    if np.any((d < 1) | (d > length)):
        raise ValueError("day out of range for its month")
    in_year = _CUM[m - 1] + ((m > 2) & leap) + (d - 1)
    p = y - 1
    if calendar == "noleap":
        return p * 365 + in_year
    if calendar == "all_leap":
        return p * 366 + in_year
    jul = p * 365 + p // 4 + in_year
    greg = p * 365 + p // 4 - p // 100 + p // 400 + in_year
    if calendar == "julian":
        return jul
    if calendar == "proleptic_gregorian":
        return greg
    # standard: the day count runs on through the switch - Julian 0001-01-01 is two days before the proleptic Gregorian one
    return np.where(after, greg + 2, jul)


class TimeAxis:
    """calendar + microseconds since that calendar's 0001-01-01T00:00 (int64 array)."""

    def __init__(self, calendar: str, microseconds):
        self.calendar = canonical_calendar(calendar)
        self.us = np.asarray(microseconds, dtype=np.int64)

    # -- constructors
    @classmethod
    def from_components(cls, calendar: str, components) -> "TimeAxis":
        """components: array-like (..., k) of [year, month, day, hour, minute, second, microsecond][:k], k >= 3."""
        c = np.asarray(components, dtype=np.int64)
        if c.ndim < 1 or not (3 <= c.shape[-1] <= 7):
            raise ValueError("components must be (..., 3 to 7): year, month, day[, hour, minute, second, microsecond]")
        pad = np.zeros(c.shape[:-1] + (7,), dtype=np.int64)
        pad[..., : c.shape[-1]] = c
        H, M, S, U = pad[..., 3], pad[..., 4], pad[..., 5], pad[..., 6]
        if np.any((H < 0) | (H > 23) | (M < 0) | (M > 59) | (S < 0) | (S > 59) | (U < 0) | (U >= US_PER_SECOND)):
            raise ValueError("time of day out of range")
        days = days_since_base(calendar, pad[..., 0], pad[..., 1], pad[..., 2])
        return cls(calendar, days * US_PER_DAY + ((H * 60 + M) * 60 + S) * US_PER_SECOND + U)

    @classmethod
    def from_datetimes(cls, values, calendar: Optional[str] = None) -> "TimeAxis":
        """values: (nested sequence / object array of) objects with year ... second attributes; the calendar is `calendar`, else
        the first object's ``calendar`` attribute (cftime), else proleptic_gregorian (datetime.datetime)."""
        arr = np.asarray(values, dtype=object)
        flat = arr.ravel()
        if flat.size == 0:
            raise ValueError("no times given")
        if calendar is None:
            calendar = getattr(flat[0], "calendar", None) or "proleptic_gregorian"
        comps = np.array([[v.year, v.month, v.day, v.hour, v.minute, v.second, getattr(v, "microsecond", 0)] for v in flat],
                         dtype=np.int64).reshape(arr.shape + (7,))
        return cls.from_components(calendar, comps)

    @classmethod
    def regular(cls, start: Sequence[int], timestep: datetime.timedelta, n_times: int, n_samples: int = 1,
                calendar: str = "proleptic_gregorian") -> "TimeAxis":
        """(n_samples, n_times) axis: start, start + timestep, ... in every sample (the reference's xr.date_range records)."""
        t0 = cls.from_components(calendar, np.asarray(start, dtype=np.int64)).us
        step = _timedelta_us(timestep)
        row = t0 + step * np.arange(n_times, dtype=np.int64)
        return cls(calendar, np.broadcast_to(row, (n_samples, n_times)).copy())

    # -- array behaviour
    @property
    def shape(self):
        return self.us.shape

    @property
    def ndim(self) -> int:
        return self.us.ndim

    def __getitem__(self, index) -> "TimeAxis":
        return TimeAxis(self.calendar, self.us[index])

    def __len__(self) -> int:
        return len(self.us)

    def __sub__(self, delta: datetime.timedelta) -> "TimeAxis":
        return TimeAxis(self.calendar, self.us - _timedelta_us(delta))

    def __add__(self, delta: datetime.timedelta) -> "TimeAxis":
        return TimeAxis(self.calendar, self.us + _timedelta_us(delta))

    def __eq__(self, other) -> bool:
        return isinstance(other, TimeAxis) and other.calendar == self.calendar and np.array_equal(other.us, self.us)

    def __repr__(self) -> str:
        return f"TimeAxis({self.calendar!r}, shape={self.us.shape})"

    def microseconds_since(self, components: Sequence[int]) -> np.ndarray:
        """self - (a date of the same calendar), in microseconds."""
        return self.us - TimeAxis.from_components(self.calendar, np.asarray(components, dtype=np.int64)).us

    def microseconds_of_day(self) -> np.ndarray:
        return self.us % US_PER_DAY


def _timedelta_us(delta: datetime.timedelta) -> int:
    return (delta.days * 86_400 + delta.seconds) * US_PER_SECOND + delta.microseconds


def as_time_axis(time: Union[TimeAxis, Iterable, None], calendar: Optional[str] = None) -> Optional[TimeAxis]:
    """TimeAxis as is; anything else through ``from_datetimes`` (objects with year ... second attributes)."""
    if time is None or isinstance(time, TimeAxis):
        return time
    if hasattr(time, "to_numpy"):           # an xarray.DataArray of cftime objects, when the caller has xarray
        time = time.to_numpy()
    return TimeAxis.from_datetimes(time, calendar)
