import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as pu
bcs = "outflow"
extra = ["mesh/%s=%s" % (f, bcs) for f in ("ix1_bc", "ox1_bc", "ix2_bc", "ox2_bc", "ix3_bc", "ox3_bc")]
extra += ["problem/ul=-4.0", "problem/ur=4.0", "problem/dr=1.0", "problem/pl=0.4", "problem/pr=0.4"]
sim, osim, is_mhd = pu.make_pair("sod", 32, 3, 16, fused=True, native=False, cfl=0.3, extra=extra, params={"dfloor": 0.5, "pfloor": 0.39})
for c in range(8):
    sim.Execute(max_cycles=1); osim.step()
    P, O = pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd)
    d = np.argwhere(P["u0"] != O["u0"])
    w = np.argwhere(sim.phys.w0.cpu().numpy() != osim.array("w0"))
    print("cycle", c + 1, "u0 diffs", len(d), "w0 diffs", len(w), "dt", sim.pmesh.dt == osim.dt)
    if len(d):
        print("  u0 axes:", [np.unique(d[:, a])[:12] for a in range(5)])
        i = tuple(d[0]); print("  first", i, P["u0"][i], O["u0"][i])
        break
