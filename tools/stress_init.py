"""GPU box: several processes at once, each creating / initialising / stepping / destroying native simulations of the deck
whose Initialize aborted once under four pytest workers (isothermal MHD linear wave 24^3, 8 MeshBlocks, task chain) and of
two neighbours of it; stderr of every process is kept.  usage: python tools/stress_init.py [nproc] [repeats]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_util as pu
n = int(sys.argv[1])
for r in range(n):
    for fused in (False, True):
        res = pu.compare_run("linear_wave_mhd", 24, 3, 12, 2, native=True, fused=fused, recon="plm", rsolver="hlld",
                             extra=["problem/amp=0.1", "mhd/eos=isothermal"])
        assert res["bitwise_equal"], res
    res = pu.compare_run("sod", 32, 3, 16, 2, native=True, fused=True, cfl=0.3)
    assert res["bitwise_equal"], res
print("ok", n)
""" % (ROOT, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 40
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
    print("FAILED" if bad else "all clean")
