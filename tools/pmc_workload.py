"""workload for tools/pmc.sh: calibration copies of known size + 2 RK2 cycles of the bench
configuration (3-D Orszag-Tang 256^3 MHD, fused stage path)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from athenak_amd import capi  # noqa: E402
from athenak_amd.main import Simulation, load_deck  # noqa: E402

torch.cuda.set_device(0)
L = capi.lib()
n = 64*1024*1024            # 512 MiB read + 512 MiB written: far beyond the 256 MiB MALL
a = torch.randn(n, dtype=torch.float64, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    capi.check(L.akmi_calib_copy(capi._p(b), capi._p(a), C.c_longlong(n), capi._stream()), "calib")
torch.cuda.synchronize()
nx = int(os.environ.get("AKMI_PMC_NX", "256"))
ov = ["time/cfl_number=0.3", "time/nlim=-1", "time/tlim=1.0e9"]
for q in (1, 2, 3):
    ov += ["mesh/nx%d=%d" % (q, nx), "meshblock/nx%d=%d" % (q, nx)]
deck = "orszag_tang.athinput"
if os.environ.get("AKMI_PMC_PROBLEM", "") == "sod":          # hydro PLM+HLLC (bench.py --problem sod)
    deck = "sod.athinput"
    ov += ["mesh/ix1_bc=outflow", "mesh/ox1_bc=outflow"]
if os.environ.get("AKMI_PMC_NATIVE", "") == "1":            # the C++ host instead of the Python host
    from athenak_amd.native import NativeSimulation
    sim = NativeSimulation(load_deck(deck, ov))
else:
    sim = Simulation(load_deck(deck, ov))
sim.Execute(max_cycles=2)
torch.cuda.synchronize()
print("done", sim.pmesh.ncycle)
