"""SMR on the CPU: the product's index-table generator against the oracle's
restatement of buffs_cc.cpp / buffs_fc.cpp, and pins of the oracle's level-aware boundary values on
the invariants the reference's own SMR/AMR regressions check: conservation across fine/coarse faces,
div B at round-off (tst/test_suite/nr/test_nr_divb_amr_mpicpu.py:38-40: <= 2e-11), identical magnetic
flux through shared fine/coarse faces, second-order convergence of a linear wave through a refined
region (tst/test_suite/nr/test_nr_lwave2d_amr_mpicpu.py).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity_util as pu  # noqa: E402
from oracle import akref  # noqa: E402
from athenak_amd import bvals_smr as bs  # noqa: E402
from athenak_amd.main import load_deck  # noqa: E402
from athenak_amd.mesh import Mesh, RegionIndcs  # noqa: E402
from athenak_amd.mesh_tree import NeighborIndex  # noqa: E402


def _smr_lib():
    L = akref.lib()
    L.akref_smr_create.restype = C.c_void_p
    L.akref_smr_create.argtypes = [C.POINTER(akref.Pack), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.akref_smr_indices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.akref_smr_destroy.argtypes = [C.c_void_p]
    return L


@pytest.mark.parametrize("nx,ng", [((8, 1, 1), 2), ((8, 8, 1), 2), ((8, 8, 8), 2), ((12, 8, 16), 4),
                                   ((16, 12, 1), 4), ((8, 8, 8), 4)])
@pytest.mark.parametrize("ml", [0, 1])
def test_index_tables_match_oracle(nx, ng, ml):
    """every entry of the 2 x 6 x 56 x 3 boxes of both variable classes"""
    L = _smr_lib()
    ind = RegionIndcs(ng, *nx)
    ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
    cc, fc, ndat = bs.index_tables(ind, ndim, bool(ml))
    dx = np.ones((1, 3))
    pk = akref.Pack(1, 5, nx[0], nx[1], nx[2], ng, dx.ctypes.data, 1.4, 0, 0, 0, 0, 0, 1.0, 1)
    ngh = -np.ones((1, 56, 3), dtype=np.int32)
    lev = np.zeros(1, dtype=np.int32)
    s = L.akref_smr_create(C.byref(pk), 5, ngh.ctypes.data, lev.ctypes.data, ml)
    used_slots = {q[0] for q in bs.slot_list(ndim, bool(ml))}
    try:
        for fcq, tab in ((0, cc), (1, fc)):
            for sr in (0, 1):
                for n in range(56):
                    out = np.zeros((6, 3, 6), dtype=np.int32)
                    used = L.akref_smr_indices(s, fcq, 1 - sr, n, out.ctypes.data)
                    assert bool(used) == (n in used_slots)
                    if not used:
                        assert not tab[sr, :, n].any()
                        continue
                    for ki, kind in enumerate(bs.KINDS):
                        if (kind == "prol" and sr == 0) or (kind == "flxs" and not fcq):
                            continue
                        nv = 3 if fcq else 1
                        assert np.array_equal(tab[sr, ki, n, :nv], out[ki, :nv]), (fcq, sr, n, kind)
    finally:
        L.akref_smr_destroy(s)


def _oracle(deck, ov):
    pin = load_deck(deck, ov)
    pm = Mesh(pin)
    okw = pu.oracle_kwargs(pin)
    assert pm.multilevel
    okw.update(pu.smr_tables(pm))
    o = akref.Sim(**okw)
    o.initialize()
    return pm, okw, o


SMALL3D = ["mesh/nx1=32", "mesh/nx2=16", "mesh/nx3=16", "meshblock/nx1=8", "meshblock/nx2=4",
           "meshblock/nx3=4"]


def test_send_buffer_matches_receive_buffer():
    """the box a block packs for slot n has the extents of the box its neighbour unpacks from slot
    dest, for every neighbour of the reference's SMR test mesh (both variable classes)"""
    pin = load_deck("linear_wave_mhd_smr.athinput", SMALL3D)
    pm = Mesh(pin)
    cc, fc, _ = bs.index_tables(pm.mb_indcs, 3, True)
    pmb = pm.pmb_pack.pmb

    def ext(b):
        return (b[1] - b[0], b[3] - b[2], b[5] - b[4])
    checked = 0
    for m in range(pm.nmb_total):
        for n, nb in pmb.nghbr[m].items():
            rel = np.sign(nb.lev - pmb.mb_lev[m])
            sk = {-1: 1, 0: 0, 1: 2}[rel]           # coarser neighbour: icoar, same: isame, finer: ifine
            rk = {-1: 2, 0: 0, 1: 1}[rel]           # the receiver sees the opposite relation
            assert ext(cc[0, sk, n, 0]) == ext(cc[1, rk, nb.dest, 0]), (m, n)
            for v in range(3):
                assert ext(fc[0, sk, n, v]) == ext(fc[1, rk, nb.dest, v]), (m, n, v)
            checked += 1
    assert checked > 1000


def test_hydro_smr_conserves():
    pm, okw, o = _oracle("linear_wave_hydro_smr.athinput", SMALL3D + ["time/nlim=12"])
    t0 = np.array(o.totals())
    for _ in range(12):
        assert o.step()
    t1 = np.array(o.totals())
    assert np.abs(t1 - t0).max() < 2e-13, t1 - t0


def test_mhd_smr_conserves_and_stays_solenoidal():
    pm, okw, o = _oracle("linear_wave_mhd_smr.athinput", SMALL3D + ["time/nlim=12"])
    t0 = np.array(o.totals())
    for _ in range(12):
        assert o.step()
    t1 = np.array(o.totals())
    assert np.abs(t1 - t0).max() < 5e-13, t1 - t0
    assert o.divb()[0] <= 2.0e-11                    # test_nr_divb_amr_mpicpu.py:38-40
    # magnetic flux through every coarse face that touches finer blocks == sum over the fine faces
    ng = okw["ng"]
    nx = (okw["mb_nx1"], okw["mb_nx2"], okw["mb_nx3"])
    s = [ng]*3
    e = [ng + nx[d] - 1 for d in range(3)]
    c = [n//2 for n in nx]
    pmb = pm.pmb_pack.pmb
    B = [o.array("b0x1f"), o.array("b0x2f"), o.array("b0x3f")]
    worst, faces = 0.0, 0
    for d in range(3):
        t = [q for q in range(3) if q != d]
        for side in (-1, 1):
            for m in range(pm.nmb_total):
                for f2 in (0, 1):
                    for f1 in (0, 1):
                        off = [0, 0, 0]
                        off[d] = side
                        nb = pmb.nghbr[m].get(NeighborIndex(off[0], off[1], off[2], f1, f2))
                        if nb is None or nb.lev <= pmb.mb_lev[m]:
                            continue
                        for a2 in range(c[t[1]]):
                            for a1 in range(c[t[0]]):
                                ci, fi = [0, 0, 0], [0, 0, 0]
                                ci[d] = e[d] + 1 if side > 0 else s[d]
                                fi[d] = s[d] if side > 0 else e[d] + 1
                                ci[t[0]] = s[t[0]] + f1*c[t[0]] + a1
                                ci[t[1]] = s[t[1]] + f2*c[t[1]] + a2
                                fi[t[0]] = s[t[0]] + 2*a1
                                fi[t[1]] = s[t[1]] + 2*a2
                                acc = 0.0
                                for b2 in (0, 1):
                                    for b1 in (0, 1):
                                        g = list(fi)
                                        g[t[0]] += b1
                                        g[t[1]] += b2
                                        acc += B[d][nb.gid, g[2], g[1], g[0]]
                                worst = max(worst, abs(B[d][m, ci[2], ci[1], ci[0]] - 0.25*acc))
                                faces += 1
    assert faces > 500 and worst < 1e-13, (faces, worst)


# error-ratio thresholds of tst/test_suite/nr/test_nr_lwave2d_amr_mpicpu.py:13-16 (rk2 + plm, wave 0).
# The reference's run refines adaptively wherever the density is high, so its absolute error bound
# (1.2e-05 / 1.3e-05 at 128x64) belongs to another mesh; here a static region covers a quarter of the
# domain -- same operators (restricted fluxes / EMFs, prolongated ghost zones), same resolutions,
# solvers, amplitude and CFL number -- and the absolute bound is the measured value + 10 %.
LW2D_AMR = {"linear_wave_hydro_smr.athinput": ("hydro", "hllc", 1.65e-05, 0.29),
            "linear_wave_mhd_smr.athinput": ("mhd", "hlld", 1.8e-05, 0.28)}


@pytest.mark.parametrize("deck", sorted(LW2D_AMR))
def test_linear_wave_converges_through_refined_region_2d(deck):
    blk, rs, max_err, max_ratio = LW2D_AMR[deck]
    errs = {}
    for n in (64, 128):
        ov = ["mesh/nx1=%d" % n, "mesh/nx2=%d" % (n//2), "mesh/nx3=1", "meshblock/nx1=%d" % (n//16),
              "meshblock/nx2=%d" % (n//16), "meshblock/nx3=1", "time/tlim=1.0", "time/cfl_number=0.4",
              "problem/amp=1.0e-3", "%s/rsolver=%s" % (blk, rs),
              "refined_region1/x1min=1.0", "refined_region1/x1max=2.0", "refined_region1/x2min=0.4",
              "refined_region1/x2max=1.1"]
        pm, okw, o = _oracle(deck, ov)
        assert len({l.level for l in pm.lloc_eachmb}) == 2
        o.run()
        errs[n] = o.linear_wave_errors()[0]
    assert errs[128] <= max_err, errs
    assert errs[128]/errs[64] <= max_ratio, (errs, errs[128]/errs[64])
