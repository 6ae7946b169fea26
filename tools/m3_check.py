#!/usr/bin/env python
"""quick parity sweep of the 3-D MHD fused stage against the oracle (GPU box): a handful of shapes, block
decompositions, integrators and both hosts; prints one line per case.  AKMI_LIB selects the library."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402

CASES = [
    ("orszag_tang", 24, 24, {}, False),
    ("orszag_tang", 32, 16, {}, False),
    ("orszag_tang", (40, 24, 32), (40, 24, 32), {}, False),
    ("orszag_tang", (66, 34, 18), (66, 34, 18), {}, False),
    ("orszag_tang", 64, 64, {}, False),
    ("orszag_tang", 64, 64, {"integrator": "rk3"}, False),
    ("orszag_tang", 32, 32, {"integrator": "rk1"}, False),
    ("blast", 32, 16, {"recon": "plm"}, False),
    ("linear_wave_mhd", (32, 16, 16), (16, 16, 16), {}, False),
    ("orszag_tang", 48, 48, {}, True),
    ("orszag_tang", 32, 16, {"integrator": "rk3"}, True),
]
bad = 0
for prob, n, mb, kw, native in CASES:
    t0 = time.time()
    try:
        r = pu.compare_run(prob, n, 3, mb, cycles=3, fused=True, native=native, **kw)
        ok = r["bitwise_equal"] and r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1] and r["cycles"] == 3
        print("%-4s %s n=%s mb=%s %s native=%s  max_rel_l1=%.3e  (%.1fs)" % ("ok" if ok else "BAD", prob, n, mb, kw, native,
                                                                         r["max_rel_l1"], time.time() - t0), flush=True)
    except Exception as e:
        ok = False
        print("ERR  %s n=%s mb=%s %s native=%s: %r" % (prob, n, mb, kw, native, e), flush=True)
    bad += 0 if ok else 1
print("bad:", bad)
sys.exit(1 if bad else 0)
