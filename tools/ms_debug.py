#!/usr/bin/env python
"""where does the fused MHD stage differ from the oracle?  one cycle of a small deck, index ranges of the differing cells"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu
CASES = [("linear_wave_mhd", (32, 16, 16), (16, 16, 16), {}), ("blast", 32, 16, {"recon": "plm"}),
         ("linear_wave_mhd", (16, 16, 16), (16, 16, 16), {}), ("orszag_tang", 32, 16, {})]
for prob, n, mb, kw in CASES:
    sim, osim, is_mhd = pu.make_pair(prob, n, 3, mb, fused=True, **kw)
    sim.Execute(max_cycles=1); osim.step()
    P, O = pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd)
    print("==", prob, n, mb, kw, "dt", sim.pmesh.dt, osim.dt)
    for key in P:
        d = np.argwhere(P[key] != O[key])
        if len(d) == 0:
            print("  ", key, "identical"); continue
        print("  ", key, P[key].shape, "differs in", len(d), "entries; per axis min/max:", d.min(axis=0), d.max(axis=0),
              "max abs", np.abs(P[key] - O[key]).max())
        for ax in range(d.shape[1]):
            print("      axis", ax, "values:", np.unique(d[:, ax])[:40])
