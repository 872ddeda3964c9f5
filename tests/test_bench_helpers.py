"""Host-only pieces of bench.py that the multi-GPU line depends on (no GPU, no library): the flag-wait percentile summary and the watchdog around the RCCL check."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_wait_percentiles_reads_log2_bins_of_10ns_ticks():
    h = np.zeros((2, 2, 32), np.uint32)
    h[0, 0, 6] = 90          # [0.64, 1.28) us
    h[1, 1, 7] = 9           # [1.28, 2.56) us
    h[0, 1, 12] = 1          # [40.96, 81.92) us
    p = bench.wait_percentiles(h)
    assert p["n"] == 100 and p["p50_us"] == 1.28 and p["p90_us"] == 1.28 and p["p99_us"] == 2.56 and p["max_us"] == 81.92
    assert bench.wait_percentiles(np.zeros((2, 2, 32), np.uint32)) == {"n": 0}


def test_guarded_returns_results_reports_hangs_and_swallows_exceptions():
    assert bench.guarded(lambda: 7, 5.0) == (7, False)
    r, alive = bench.guarded(lambda: time.sleep(3.0), 0.2)
    assert r is None and alive
    r, alive = bench.guarded(lambda: 1 / 0, 5.0)
    assert r is None and not alive
