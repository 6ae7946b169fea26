import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import parity_util as pu
cases = [("linear_wave_hydro_smr", (64,32,32), 3, (16,8,8), {}),
         ("linear_wave_mhd_smr", (64,32,32), 3, (16,8,8), {}),
         ("blast_smr", (32,32,32), 3, (8,8,8), {}),
         ("linear_wave_mhd_smr", (64,32,32), 3, (16,8,8), dict(recon="ppm4", ng=4, rsolver="hlld"))]
for prob, n, dims, mb, kw in cases:
    try:
        r = pu.compare_run(prob, n, dims, mb, cycles=2, **kw)
        print(prob, kw, "max_rel_l1 %.3e" % r["max_rel_l1"], "bitwise", r["bitwise_equal"], r["diffs"], r["dt"])
    except Exception as e:
        import traceback; traceback.print_exc()
