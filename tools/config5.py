"""BASELINE config 5 (3-D MHD blast, 2-level static refinement, PPM4 + HLLD + CT, ng = 4) at the deck's own size
on one GPU: Python host, C++ host, and the CPU oracle beside them.   python tools/config5.py [cycles]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from athenak_amd.main import Simulation, load_deck  # noqa: E402
from athenak_amd.native import NativeSimulation  # noqa: E402

ncyc = int(sys.argv[1]) if len(sys.argv) > 1 else 40
extra = sys.argv[2:]
hosts = [h for h in ("python host", "c++ host") if os.environ.get("AKMI_CONFIG5_HOSTS", "python,c++").find(h.split()[0]) >= 0]
for kind in hosts:
    pin = load_deck("blast_mhd_smr.athinput", ["time/nlim=-1", "time/tlim=1.0e9"] + extra)
    sim = Simulation(pin) if kind == "python host" else NativeSimulation(pin)
    pm = sim.pmesh
    sim.Execute(max_cycles=5)
    torch.cuda.synchronize()
    t = time.time()
    n = sim.Execute(max_cycles=ncyc)
    torch.cuda.synchronize()
    el = time.time() - t
    ncell = pm.nmb_total*pm.NumberOfMeshBlockCells()
    levels = sorted(set(int(l) for l in pm.pmb_pack.pmb.mb_lev))
    print("config 5 | %-11s | %d MeshBlocks of %d^3 on levels %s | %.3f ms/cycle | %.1f Mcell-updates/s" % (
        kind, pm.nmb_total, pm.mb_indcs.nx1, levels, el/n*1e3, ncell*n/el/1e6), flush=True)

if os.environ.get("AKMI_CONFIG5_CPU", "1") != "0":
    import parity_util as pu
    from athenak_amd.mesh import Mesh
    from oracle import akref
    pin = load_deck("blast_mhd_smr.athinput", ["time/nlim=-1", "time/tlim=1.0e9"] + extra)
    okw = pu.oracle_kwargs(pin)
    okw.update(pu.smr_tables(Mesh(pin)))
    akref.lib().akref_set_threads(16)
    o = akref.Sim(**okw)
    o.initialize()
    o.step()
    t = time.time()
    k = 0
    while time.time() - t < 10.0:
        o.step()
        k += 1
    el = time.time() - t
    print("config 5 | cpu oracle (16 threads) | %.1f ms/cycle | %.2f Mcell-updates/s" % (el/k*1e3, ncell*k/el/1e6))
