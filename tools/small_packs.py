"""Cycle time of launch-bound packs (BASELINE configs 1 and 2 and neighbours) through the Python host and the
C++ host, the latter with and without the captured cycle graph (AKMI_CYCLE_GRAPH).  GPU box:
    python tools/small_packs.py > gpurun_out/small_packs.txt"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [
    # label, deck, overrides
    ("C1 lwave1d hydro 256x1x1", "linear_wave_hydro.athinput",
     ["mesh/nx1=256", "mesh/nx2=1", "mesh/nx3=1", "meshblock/nx1=256", "meshblock/nx2=1", "meshblock/nx3=1",
      "problem/along_x1=true"]),
    ("lwave1d mhd 256x1x1", "linear_wave_mhd.athinput",
     ["mesh/nx1=256", "mesh/nx2=1", "mesh/nx3=1", "meshblock/nx1=256", "meshblock/nx2=1", "meshblock/nx3=1",
      "problem/along_x1=true"]),
    ("C2 sod hydro 64^3", "sod.athinput", ["mesh/nx1=64", "mesh/nx2=64", "mesh/nx3=64", "meshblock/nx1=64",
                                          "meshblock/nx2=64", "meshblock/nx3=64", "time/cfl_number=0.3"]),
    ("sod hydro 128^3", "sod.athinput", ["mesh/nx1=128", "mesh/nx2=128", "mesh/nx3=128", "meshblock/nx1=128",
                                         "meshblock/nx2=128", "meshblock/nx3=128", "time/cfl_number=0.3"]),
    ("orszag_tang mhd 64^3", "orszag_tang.athinput", ["mesh/nx1=64", "mesh/nx2=64", "mesh/nx3=64", "meshblock/nx1=64",
                                                      "meshblock/nx2=64", "meshblock/nx3=64", "time/cfl_number=0.3"]),
    ("orszag_tang mhd 2D 256^2", "orszag_tang.athinput", ["mesh/nx1=256", "mesh/nx2=256", "mesh/nx3=1", "meshblock/nx1=256",
                                                          "meshblock/nx2=256", "meshblock/nx3=1", "time/cfl_number=0.3"]),
    ("orszag_tang mhd 64^3 in 8 blocks of 32^3", "orszag_tang.athinput",
     ["mesh/nx1=64", "mesh/nx2=64", "mesh/nx3=64", "meshblock/nx1=32", "meshblock/nx2=32", "meshblock/nx3=32",
      "time/cfl_number=0.3"]),
]


def one(idx, kind):
    import torch
    from athenak_amd.main import Simulation, load_deck
    from athenak_amd.native import NativeSimulation
    label, deck, ov = CASES[idx]
    pin = load_deck(deck, ov + ["time/nlim=-1", "time/tlim=1.0e9"])
    sim = Simulation(pin) if kind == "python" else NativeSimulation(pin)
    sim.Execute(max_cycles=20)
    torch.cuda.synchronize()
    t = time.time()
    n = sim.Execute(max_cycles=200)
    torch.cuda.synchronize()
    el = time.time() - t
    ncell = sim.pmesh.nmb_total*sim.pmesh.NumberOfMeshBlockCells()
    print("%-44s %-14s %8.1f us/cycle %9.1f Mcell-updates/s" % (label, kind, el/n*1e6, ncell*n/el/1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3:
        one(int(sys.argv[1]), sys.argv[2])
    else:
        for i in range(len(CASES)):
            for kind, env in (("python", {}), ("c++", {"AKMI_CYCLE_GRAPH": "0"}), ("c++ graph", {"AKMI_CYCLE_GRAPH": "1"})):
                e = dict(os.environ)
                e.update(env)
                subprocess.run([sys.executable, os.path.abspath(__file__), str(i), kind], env=e)
