"""CPU checks of bench.py's bookkeeping: the algorithmic work figures it prices the roofline with are the
ones SURVEY.md §8d states, and its host-core detection is sane."""
import importlib.util
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_work_per_curve_matches_survey_8d():
    b = _bench()
    # B1=1e6: A=1,980,817 adds, D=217,929 doublings (ecm.c:1849 counters)
    mul, sqr, mads15, w8 = b.work_per_curve(1980817, 217929, 15, 8)
    assert (mul, sqr) == (8577055, 4397492) and mul + sqr == 12974547
    assert w8 == 1641408616                                   # W(B1=1e6, n=8)
    assert b.work_per_curve(1980817, 217929, 30, 16)[3] == 6322861776   # W(1e6, n=16)
    assert b.work_per_curve(1980817, 217929, 23, 12)[3] == 3602129628   # W(1e6, n=12)
    assert b.work_per_curve(195448, 23269, 37, 32)[3] == 2464221376     # W(1e5, n=32)
    assert mads15 == 8577055 * 465 + 4397492 * 360
    assert abs(b.PEAK_MAD_PER_S - 39.3216e12) < 1e6


def test_host_cores_positive():
    assert 1 <= _bench().host_cores() <= 64
