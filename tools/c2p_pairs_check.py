import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import parity_util as pu
for native in (True, False):
    for prob, n, mb, kw in (("orszag_tang", 32, 16, dict(cfl=0.3)), ("orszag_tang", (66, 34, 18), (66, 34, 18), {}), ("sod", 32, 16, dict(cfl=0.3)), ("blast", 24, 12, dict(recon="plm"))):
        r = pu.compare_run(prob, n, 3, mb, cycles=3, native=native, fused=True, **kw)
        print(prob, native, r["bitwise_equal"], r["dt"][0] == r["dt"][1])
        assert r["bitwise_equal"] and r["dt"][0] == r["dt"][1]
print("ok")
