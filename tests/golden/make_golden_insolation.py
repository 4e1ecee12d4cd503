"""Emit tests/golden/gen_insolation.pt from the reference's own insolation (fme/ace/stepper/insolation/cm4.py, imported through
oracle/ref_loader.load_insolation - build container only; the stand-in for cftime's datetimes is the standard library's
datetime.datetime, see there).  First the reference is run on the inputs of ITS OWN regression test
(test_insolation.py:125-158) and must reproduce the golden tensor it holds (tests/golden/ref_insolation_*.pt, copied data) -
that pins the stand-in; then cases the reference's tests do not hold are emitted: a coarse global grid over a year of 6-hourly
times, 3-hourly and 1-hourly timesteps, other orbital parameters, a named (time-varying, fp64) solar constant, the pole rows.

Run:  python tests/golden/make_golden_insolation.py"""
import datetime
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_loader  # noqa: E402


def main():
    R = ref_loader.load_insolation()
    cm4 = R.cm4

    def times(cls, start, step, n_times, n_samples=1, sample_offset=datetime.timedelta(0)):
        rows = [[cls(*start) + s * sample_offset + k * step for k in range(n_times)] for s in range(n_samples)]
        return R.TimeArray(rows), [[(t.year, t.month, t.day, t.hour, t.minute, t.second) for t in row] for row in rows]

    # ---- pin: the reference's own regression case
    lat, lon = torch.linspace(-90.0, 90.0, 8), torch.linspace(0.0, 360.0 - 360.0 / 16, 16)
    hc = R.LatLonCoordinates(lat=lat, lon=lon)
    f = cm4.CM4Insolation(23.439, 0.0167, 102.932)
    ta, _ = times(R.standard, (2000, 1, 1), datetime.timedelta(hours=6), 2, 3)
    got = f(ta, datetime.timedelta(hours=6), hc, torch.tensor(1360.0))
    held = torch.load(os.path.join(HERE, "ref_insolation_solar-constant-as-value.pt"))
    torch.testing.assert_close(got, held, rtol=1e-4, atol=0.0)
    print("reference under the stand-in reproduces its own golden: max rel", float(((got - held).abs() / held.clamp_min(1e-30))[held > 0].max()))

    cases = []

    def case(name, cls, start, timestep_h, n_times, lat, lon, s0, orbit=(23.439, 0.0167, 102.932), n_samples=1,
             stride_h=None, sample_offset=datetime.timedelta(0)):
        fn = cm4.CM4Insolation(*orbit)
        step = datetime.timedelta(hours=timestep_h)
        stride = step if stride_h is None else datetime.timedelta(hours=stride_h)
        ta, comps = times(cls, start, stride, n_times, n_samples, sample_offset)
        out = fn(ta, step, R.LatLonCoordinates(lat=lat, lon=lon), s0)
        cases.append({"name": name, "calendar": cls.calendar, "components": comps, "timestep_seconds": int(step.total_seconds()),
                      "lat": lat.clone(), "lon": lon.clone(), "solar_constant": s0.clone(), "orbit": orbit, "out": out.clone()})
        print(name, tuple(out.shape), out.dtype, float(out.max()))

    glat = torch.linspace(-87.5, 87.5, 36)
    glon = torch.linspace(0.0, 355.0, 72)
    ylat = torch.linspace(-86.25, 86.25, 24)
    ylon = torch.linspace(0.0, 350.0, 36)
    # a year of 6-hourly steps sampled every 53 h so that all times of day and the whole orbit are visited (166 times)
    case("year_6h", R.proleptic_gregorian, (2001, 1, 1, 0), 6, 166, ylat, ylon, torch.tensor(1360.0), stride_h=53)
    case("three_hourly", R.proleptic_gregorian, (2020, 6, 19, 21), 3, 12, glat, glon, torch.tensor(1361.5))
    case("hourly_two_samples", R.standard, (1999, 12, 31, 22), 1, 6, glat, glon, torch.tensor(1360.0), n_samples=2,
         sample_offset=datetime.timedelta(days=91, hours=5))
    case("poles_and_dateline", R.proleptic_gregorian, (2010, 3, 20, 12), 6, 8, torch.linspace(-90.0, 90.0, 9),
         torch.tensor([0.0, 90.0, 179.999, 180.0, 270.0, 359.0]), torch.tensor(1360.0))
    case("other_orbit", R.proleptic_gregorian, (2005, 9, 1, 6), 6, 10, glat, glon, torch.tensor(1300.0), orbit=(10.0, 0.05, 0.0))
    case("circular_no_tilt", R.proleptic_gregorian, (2005, 1, 1, 0), 6, 4, glat, glon, torch.tensor(1360.0), orbit=(0.0, 0.0, 102.932))
    g = torch.Generator().manual_seed(0)
    s0 = (1360.0 + torch.rand(1, 5, 36, 72, generator=g, dtype=torch.float64))
    case("named_fp64_solar_constant", R.proleptic_gregorian, (2030, 2, 27, 18), 6, 5, glat, glon, s0)
    out = os.path.join(HERE, "gen_insolation.pt")
    torch.save({"cases": cases}, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
