"""config 5's mesh at intermediate sizes against the oracle, bit for bit: 960 MeshBlocks of 16^3 through both hosts
(the block count of the production mesh) and 120 of 32^3.  GPU box: python tools/c5_mid_check.py  (~3 min)"""
import sys, time
sys.path[:0] = ['.', 'tests']
import parity_util as pu
for n, mb, native in (((128, 128, 128), (16, 16, 16), False), ((128, 128, 128), (16, 16, 16), True), ((128, 128, 128), (32, 32, 32), True)):
    t = time.time()
    r = pu.compare_run("blast_smr", n, 3, mb, cycles=3, native=native)
    print(n, mb, "native" if native else "python", "cycles", r["cycles"], "bitwise", r["bitwise_equal"], "dt", r["dt"], "%.1f s" % (time.time() - t), flush=True)
