#!/usr/bin/env python
"""Developer micro-benchmark: time the pieces of one MHD stage (256^3 Orszag-Tang) with HIP
events for one or several builds of the library (AKMI_LIB=...).  Usage on the GPU box:
    python tools/kbench.py [nx] [lib1.so lib2.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(nx):
    import torch
    from athenak_amd.main import Simulation, load_deck
    torch.cuda.set_device(0)
    ov = ["time/cfl_number=0.3", "time/nlim=-1", "time/tlim=1.0e9"]
    for q in (1, 2, 3):
        ov += ["mesh/nx%d=%d" % (q, nx), "meshblock/nx%d=%d" % (q, nx)]
    sim = Simulation(load_deck("orszag_tang.athinput", ov))
    pm, drv, ph = sim.pmesh, sim.pdriver, sim.phys
    for _ in range(2):
        drv._cycle(pm)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tA = tH = tB = 0.0
    n = 5
    for _ in range(n):
        for stage in (1, 2):
            ev[0].record()
            ph.RKUpdate(drv, stage)
            ev[1].record()
            ph.SendU(drv, stage); ph.RecvU(drv, stage); ph.SendB(drv, stage); ph.RecvB(drv, stage)
            ev[2].record()
            ph.ConToPrim(drv, stage); ph.NewTimeStep(drv, stage)
            ev[3].record()
            torch.cuda.synchronize()
            tA += ev[0].elapsed_time(ev[1]); tH += ev[1].elapsed_time(ev[2]); tB += ev[2].elapsed_time(ev[3])
        pm.time += pm.dt; pm.ncycle += 1; pm.NewTimeStep(drv.tlim)
    k = 2*n
    tot = (tA + tH + tB)/k
    print("%-40s passA %.3f ms  halo %.3f ms  passB %.3f ms  stage %.3f ms  -> %.0f Mcell-updates/s"
          % (os.path.basename(os.environ.get("AKMI_LIB", "libakmi.so")), tA/k, tH/k, tB/k, tot,
             nx**3/(2*tot*1e-3)/1e6))


def native_vs_python(problem, nx, cycles):
    """whole-cycle wall time of the Python host mirror vs the C++ host driver"""
    import time
    import torch
    from athenak_amd.main import Simulation, load_deck
    from athenak_amd.native import NativeSimulation
    torch.cuda.set_device(0)
    deck = {"sod": "sod.athinput", "orszag_tang": "orszag_tang.athinput",
            "lw1d": "linear_wave_hydro.athinput"}[problem]
    ov = ["time/cfl_number=0.3", "time/nlim=-1", "time/tlim=1.0e9"]
    dims = 1 if problem == "lw1d" else 3
    for q in (1, 2, 3):
        n = nx if q <= dims else 1
        ov += ["mesh/nx%d=%d" % (q, n), "meshblock/nx%d=%d" % (q, n)]
    if problem == "lw1d":
        ov += ["problem/along_x1=true"]
    for name, cls in (("python host", Simulation), ("c++ host", NativeSimulation)):
        sim = cls(load_deck(deck, ov))
        sim.Execute(max_cycles=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.Execute(max_cycles=cycles)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("%-12s %s %d^%d: %.1f us/cycle  %.1f Mcell-updates/s" % (name, problem, nx, dims,
              el/cycles*1e6, nx**dims*cycles/el/1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hosts":
        for prob, nx, cyc in (("lw1d", 256, 300), ("sod", 64, 100), ("sod", 128, 50), ("orszag_tang", 128, 30)):
            native_vs_python(prob, nx, cyc)
    elif os.environ.get("AKMI_KBENCH_CHILD"):
        child(int(sys.argv[1]))
    else:
        nx = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
        libs = [a for a in sys.argv[1:] if a.endswith(".so")] or [None]
        for lib in libs:
            env = dict(os.environ, AKMI_KBENCH_CHILD="1")
            if lib:
                env["AKMI_LIB"] = os.path.abspath(lib)
            subprocess.call([sys.executable, os.path.abspath(__file__), str(nx)], env=env)
