import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athenak_amd.main import load_deck, Simulation
from athenak_amd.native import NativeSimulation
for nx in (128, 64):
    ov=["mesh/nx1=%d"%nx,"mesh/nx2=%d"%nx,"mesh/nx3=%d"%nx,"meshblock/nx1=%d"%nx,"meshblock/nx2=%d"%nx,"meshblock/nx3=%d"%nx,"time/cfl_number=0.3","time/nlim=-1","time/tlim=1.0e9","mesh/ix1_bc=outflow","mesh/ox1_bc=outflow"]
    for kind in ("python","native"):
        pin=load_deck("sod.athinput", ov)
        sim = NativeSimulation(pin) if kind=="native" else Simulation(pin)
        sim.Execute(max_cycles=5); torch.cuda.synchronize()
        t=time.time(); n=sim.Execute(max_cycles=50); torch.cuda.synchronize(); el=time.time()-t
        print(nx, kind, "ms/cycle %.3f  Mcell/s %.1f"%(el/n*1e3, nx**3*n/el/1e6), flush=True)
