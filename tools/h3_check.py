#!/usr/bin/env python
"""quick parity sweep of the 3-D hydro one-kernel stage against the oracle (GPU box): shapes, block decompositions, DC / PLM,
Riemann solvers, isothermal, passive scalars, integrators, both hosts; one line per case.  AKMI_LIB selects the library."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu  # noqa: E402

CASES = [
    ("sod", 32, 32, dict(cfl=0.3), False),
    ("sod", 32, 16, dict(cfl=0.3), False),
    ("sod", (40, 24, 32), (40, 24, 32), dict(cfl=0.3), False),
    ("sod", (66, 34, 18), (66, 34, 18), dict(cfl=0.3), False),
    ("linear_wave_hydro", 24, 12, {}, False),
    ("linear_wave_hydro", (32, 16, 16), (16, 16, 16), {}, False),
    ("linear_wave_hydro", 32, 32, dict(recon="dc"), False),
    ("linear_wave_hydro", 32, 16, dict(integrator="rk3"), False),
    ("sod", 32, 32, dict(cfl=0.3, integrator="rk1"), False),
    ("linear_wave_hydro", 24, 24, dict(rsolver="llf"), False),
    ("linear_wave_hydro", 24, 24, dict(rsolver="hlle"), False),
    ("linear_wave_hydro", 24, 24, dict(rsolver="roe"), False),
    ("linear_wave_hydro", 24, 12, dict(ng=4), False),
    ("linear_wave_hydro", 24, 24, dict(rsolver="hlle", extra=["hydro/eos=isothermal"]), False),
    ("linear_wave_hydro", 24, 12, dict(extra=["hydro/nscalars=2"]), False),
    ("sod", 48, 48, dict(cfl=0.3), True),
    ("linear_wave_hydro", 32, 16, dict(integrator="rk3"), True),
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
