"""CPU checks of bench.py's bookkeeping (no GPU): the algorithmic-bytes formula against SURVEY.md section 8d."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey_table():
    b = _bench()
    # cfg1 2.6 KB, cfg2 17.7 KB, cfg3 65.4 KB, cfg4 248.1 KB, cfg5 1.479 MB
    assert b.algorithmic_bytes(1, 10, 4, 6) == 0 + 264 + 552 + 768 + 1036
    assert b.algorithmic_bytes(64, 10, 4, 6) == 15120 + 2620
    assert b.algorithmic_bytes(256, 10, 4, 18) == 61200 + 264 + 552 + 2304 + 1036
    assert b.algorithmic_bytes(1024, 10, 4, 6) == 245520 + 2620
    assert b.algorithmic_bytes(4096, 15, 4, 18) == 1474200 + 384 + 792 + 2304 + 1516
    assert b.HBM_PEAK_GBS == 8000.0
