"""GPU box: several processes on the one GPU, each running whole native simulations with the RUN-AHEAD Driver (C++ host one
cycle ahead of the device: pinned slots, events, Mesh::NewTimeStep on the device) to the same end state as the oracle, again
and again -- the concurrent-Execute companion of tools/stress_init.py (which exercises create / initialize / destroy).
stderr of every process is kept; AMD_LOG_LEVEL / HSA_ENABLE_DEBUG pass through for a post-mortem.
usage: python tools/stress_execute.py [nproc] [repeats]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_util as pu
n = int(sys.argv[1])
ncyc = 0
for r in range(n):
    # eligible for run-ahead: fused stage, one rank, uniform mesh (hydro one-kernel stage; MHD four-kernel stage; 1-D)
    for prob, nx, dims, mb, kw in (("sod", 48, 3, 48, dict(cfl=0.3)), ("orszag_tang", 32, 3, 32, dict(cfl=0.3)),
                                   ("linear_wave_mhd", 24, 3, 12, {}), ("linear_wave_hydro", 128, 1, 128, dict(extra=["problem/along_x1=true"]))):
        sim, osim, is_mhd = pu.make_pair(prob, nx, dims, mb, fused=True, native=True, **kw)
        done = sim.Execute(max_cycles=12)              # ONE call: the host runs ahead inside it
        for _ in range(done):
            osim.step()
        d = pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, is_mhd), is_mhd)
        assert d["bitwise_equal"] and sim.pmesh.time == osim.time and sim.pmesh.dt == osim.dt, (prob, d, sim.pmesh.time, osim.time)
        ncyc += done
        sim.close()
print("ok", n, ncyc)
""" % (ROOT, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ps = [subprocess.Popen([sys.executable, "-c", CHILD, str(rep)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for _ in range(nproc)]
    bad = 0
    for i, p in enumerate(ps):
        out, err = p.communicate()
        ok = p.returncode == 0 and "ok" in out
        bad += not ok
        print("proc %d rc %d %s" % (i, p.returncode, out.strip()[-60:]))
        if not ok:
            print(err[-3000:])
    print("FAILED" if bad else "all clean: %d processes x %d repeats" % (nproc, rep))
