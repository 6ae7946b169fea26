"""SMR (BASELINE config 5) on the GPU: the HIP path through the C ABI (akmi_smr_*)
against the CPU oracle, bit for bit -- whole runs on statically refined meshes in 1-D/2-D/3-D,
hydro and MHD, two and three levels, periodic and physical boundaries, and every boundary operator
on random data."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

S3 = ("mesh/nx1=32", "mesh/nx2=16", "mesh/nx3=16")          # root grid of the small 3-D cases
B3 = (8, 4, 4)

CASES = {
    # name: (problem, n, dims, mb, cycles, kwargs)
    "hydro3d": ("linear_wave_hydro_smr", (32, 16, 16), 3, B3, 3, {}),
    "hydro3d_hllc_rk3": ("linear_wave_hydro_smr", (32, 16, 16), 3, B3, 2, dict(rsolver="hllc", integrator="rk3")),
    "mhd3d": ("linear_wave_mhd_smr", (32, 16, 16), 3, B3, 3, dict(rsolver="hlld")),
    "mhd3d_ppm4": ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 8, 8), 2, dict(recon="ppm4", ng=4, rsolver="hlld")),
    # BASELINE config 5 at fixture size: 3-D blast, one refined region, PPM4 + HLLD, nghost = 4
    "blast3d_c5": ("blast_smr", (32, 32, 32), 3, (8, 8, 8), 3, {}),
    "mhd2d": ("linear_wave_mhd_smr", (32, 16, 1), 2, (8, 4, 1), 4, dict(rsolver="hlld")),
    "hydro2d": ("linear_wave_hydro_smr", (32, 16, 1), 2, (8, 4, 1), 4, {}),
    "hydro1d": ("linear_wave_hydro_smr", (32, 1, 1), 1, (8, 1, 1), 6, {}),
    "mhd1d": ("linear_wave_mhd_smr", (32, 1, 1), 1, (8, 1, 1), 6, dict(rsolver="hlld")),
    # <mesh_refinement>/prolong_primitives = true: ConsToPrimCoarseBndry -> ProlongateCC(w) -> PrimToConsFineBndry
    "hydro3d_pprims": ("linear_wave_hydro_smr", (32, 16, 16), 3, B3, 3, dict(extra=("mesh_refinement/prolong_primitives=true",))),
    "mhd3d_pprims": ("linear_wave_mhd_smr", (32, 16, 16), 3, B3, 3,
                     dict(rsolver="hlld", extra=("mesh_refinement/prolong_primitives=true",))),
    "blast3d_c5_pprims": ("blast_smr", (32, 32, 32), 3, (8, 8, 8), 2, dict(extra=("mesh_refinement/prolong_primitives=true",))),
    "mhd2d_pprims": ("linear_wave_mhd_smr", (32, 16, 1), 2, (8, 4, 1), 3,
                     dict(rsolver="hlld", extra=("mesh_refinement/prolong_primitives=true",))),
    # three levels: the 2:1 rule surrounds the level-2 region with level-1 blocks
    "mhd3d_3levels": ("linear_wave_mhd_smr", (32, 16, 16), 3, B3, 2,
                      dict(rsolver="hlld", extra=("refined_region1/level=2",))),
    # refined region on the mesh boundary with physical boundary conditions: coarse-buffer BCs
    "blast3d_bcs": ("blast_smr", (32, 32, 32), 3, (8, 8, 8), 2,
                    dict(recon="plm", ng=2, extra=("mesh/ix1_bc=outflow", "mesh/ox1_bc=outflow", "mesh/ix2_bc=reflect",
                                                   "mesh/ox2_bc=reflect", "refined_region1/x1min=-0.5",
                                                   "refined_region1/x1max=-0.2", "refined_region1/x2min=-0.5",
                                                   "refined_region1/x2max=-0.2"))),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_whole_run_parity_smr(name):
    import parity_util as pu
    problem, n, dims, mb, cycles, kw = CASES[name]
    r = pu.compare_run(problem, n, dims, mb, cycles=cycles, **kw)
    assert r["cycles"] == cycles
    assert r["bitwise_equal"], r
    assert r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1]


@pytest.mark.parametrize("name", ["hydro3d", "mhd3d", "blast3d_c5", "mhd2d", "hydro1d", "mhd3d_3levels", "blast3d_bcs",
                                  "hydro3d_pprims", "mhd3d_pprims"])
def test_native_cpp_host_parity_smr(name):
    """the C++ host (csrc/akmi_host.cpp + akmi_host_smr.cpp: its own MeshBlockTree, neighbour table, index
    tables and task lists) on refined meshes: bit-identical to the oracle, same dt sequence"""
    import parity_util as pu
    problem, n, dims, mb, cycles, kw = CASES[name]
    r = pu.compare_run(problem, n, dims, mb, cycles=cycles, native=True, **kw)
    assert r["cycles"] == cycles
    assert r["bitwise_equal"], r
    assert r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1]


def test_config5_product_initial_conditions_and_invariants():
    """the product's own problem generator on the SMR blast agrees with the oracle's (numpy's and
    libm's exp/log may differ in the last place inside the pressure ramp, hence not bitwise), and
    after a few cycles on the HIP path div B stays at round-off and mass/energy are conserved"""
    import parity_util as pu
    sim, osim, is_mhd = pu.make_pair("blast_smr", (32, 32, 32), 3, (8, 8, 8), inject=False)
    a, b = pu.product_arrays(sim), pu.oracle_arrays(osim, True)
    for k in a:
        assert np.allclose(a[k], b[k], rtol=4e-16, atol=0.0), k
    sim, osim, is_mhd = pu.make_pair("blast_smr", (32, 32, 32), 3, (8, 8, 8), inject=True)
    pm = sim.pmesh
    dx = pm.pmb_pack.pmb.dx
    vol = dx[:, 0]*dx[:, 1]*dx[:, 2]
    ng, nx = 4, 8
    act = (slice(None), slice(None), slice(ng, ng + nx), slice(ng, ng + nx), slice(ng, ng + nx))

    def totals():
        u = sim.phys.u0.cpu().numpy()[act]
        return (u*vol[:, None, None, None, None]).sum(axis=(0, 2, 3, 4))
    t0 = totals()
    for _ in range(4):
        assert sim.Execute(max_cycles=1) and osim.step()
    t1 = totals()
    assert abs(t1[0] - t0[0]) < 1e-13 and abs(t1[4] - t0[4]) < 1e-13, t1 - t0
    b1, b2, b3 = (x.cpu().numpy() for x in (sim.phys.b0.x1f, sim.phys.b0.x2f, sim.phys.b0.x3f))
    s, e = ng, ng + nx
    div = ((b1[:, s:e, s:e, s + 1:e + 1] - b1[:, s:e, s:e, s:e])/dx[:, 0, None, None, None] +
           (b2[:, s:e, s + 1:e + 1, s:e] - b2[:, s:e, s:e, s:e])/dx[:, 1, None, None, None] +
           (b3[:, s + 1:e + 1, s:e, s:e] - b3[:, s:e, s:e, s:e])/dx[:, 2, None, None, None])
    assert np.abs(div).max() <= 2.0e-11                # test_nr_divb_amr_mpicpu.py:38-40
    assert pu.compare_fields(pu.product_arrays(sim), pu.oracle_arrays(osim, True), True)["bitwise_equal"]


# ---- every operator on random data ---------------------------------------------------------------
def _setup(deck, ov, nvar=5):
    import torch
    import parity_util as pu
    from oracle import akref
    from athenak_amd import capi
    from athenak_amd.bvals_smr import MeshBoundaryValuesSMR
    from athenak_amd.main import load_deck
    from athenak_amd.mesh import Mesh
    pin = load_deck(deck, list(ov))
    pm = Mesh(pin)
    ind = pm.mb_indcs
    nmb = pm.nmb_total
    dxh = np.ascontiguousarray(pm.pmb_pack.pmb.dx)
    dxd = torch.from_numpy(dxh.copy()).cuda()
    pack = capi.Pack(nmb, nvar, ind.nx1, ind.nx2, ind.nx3, ind.ng, dxd.data_ptr(), 1.4, 1e-30, 1e-30, 1e-30,
                     1e-30, 1e30, 1.0, 1)
    smr = MeshBoundaryValuesSMR(pm.pmb_pack, nvar)
    smr.set_pack(pack)
    L = akref.lib()
    L.akref_smr_create.restype = C.c_void_p
    L.akref_smr_create.argtypes = [C.POINTER(akref.Pack), C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    for f in ("akref_smr_destroy",):
        getattr(L, f).argtypes = [C.c_void_p]
    opk = akref.Pack(nmb, nvar, ind.nx1, ind.nx2, ind.nx3, ind.ng, dxh.ctypes.data, 1.4, 1e-30, 1e-30, 1e-30,
                     1e-30, 1e30, 1.0, 1)
    tabs = pu.smr_tables(pm)
    lev = np.ascontiguousarray(pm.pmb_pack.pmb.mb_lev.astype(np.int32))
    h = C.c_void_p(L.akref_smr_create(C.byref(opk), nvar, tabs["smr_nghbr"].ctypes.data, lev.ctypes.data, 1))
    keep = (dxh, dxd, tabs, lev, opk)
    return pm, smr, L, h, keep


def _shapes(pm, nvar):
    ind = pm.mb_indcs
    n3, n2, n1 = ind.ncells
    ng = ind.ng
    c1 = ind.nx1//2 + 2*ng
    c2 = ind.nx2//2 + 2*ng if ind.nx2 > 1 else 1
    c3 = ind.nx3//2 + 2*ng if ind.nx3 > 1 else 1
    return (n3, n2, n1), (c3, c2, c1)


def _rand(rng, shape):
    return np.ascontiguousarray(rng.standard_normal(shape))


OPS_MESHES = {
    "3d": ("linear_wave_mhd_smr.athinput", S3 + ("meshblock/nx1=8", "meshblock/nx2=4", "meshblock/nx3=4")),
    "3d_3lev_ng4": ("linear_wave_mhd_smr.athinput", ("mesh/nx1=32", "mesh/nx2=16", "mesh/nx3=16", "meshblock/nx1=8",
                                                       "meshblock/nx2=8", "meshblock/nx3=8", "mesh/nghost=4",
                                                       "refined_region1/level=2")),
    "2d": ("linear_wave_mhd_smr.athinput", ("mesh/nx1=32", "mesh/nx2=16", "mesh/nx3=1", "meshblock/nx1=8",
                                              "meshblock/nx2=4", "meshblock/nx3=1")),
    "1d": ("linear_wave_mhd_smr.athinput", ("mesh/nx1=32", "mesh/nx2=1", "mesh/nx3=1", "meshblock/nx1=8",
                                              "meshblock/nx2=1", "meshblock/nx3=1")),
}


@pytest.mark.parametrize("mesh", sorted(OPS_MESHES))
def test_boundary_operators_on_random_data(mesh):
    import torch
    from athenak_amd.hydro import EdgeFld, FaceFld
    nvar = 5
    pm, smr, L, h, keep = _setup(*OPS_MESHES[mesh], nvar=nvar)
    nmb = pm.nmb_total
    (n3, n2, n1), (c3, c2, c1) = _shapes(pm, nvar)
    rng = np.random.default_rng(7)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    T = lambda a: torch.from_numpy(a.copy()).cuda()

    def faces(k, j, i):
        return [_rand(rng, (nmb, k, j, i + 1)), _rand(rng, (nmb, k, j + 1, i)), _rand(rng, (nmb, k + 1, j, i))]

    def dev_faces(arrs, k, j, i):
        f = FaceFld(nmb, 0, k, j, i, "cuda")
        for dst, a in zip((f.x1f, f.x2f, f.x3f), arrs):
            dst.copy_(torch.from_numpy(a))
        return f

    def same(dev, host, what):
        assert np.array_equal(dev.cpu().numpy(), host), what

    # --- cell-centred: SendU/RecvU, FillCoarseInBndryCC, ProlongateCC
    u, cu = _rand(rng, (nmb, nvar, n3, n2, n1)), _rand(rng, (nmb, nvar, c3, c2, c1))
    du, dcu = T(u), T(cu)
    smr.PackAndSendCC(du, dcu)
    smr.RecvAndUnpackCC(du, dcu)
    L.akref_smr_send_cc(h, P(u), P(cu))
    L.akref_smr_recv_cc(h, P(u), P(cu))
    same(du, u, "exchange_cc u"); same(dcu, cu, "exchange_cc cu")
    # the product fills the coarse ghost zones only of blocks that prolongate (a coarser neighbour exists:
    # akmi_smr::needs_coarse); nothing reads the others', the oracle fills them all
    needs = smr.t_needs.cpu().numpy().astype(bool)
    assert needs.any()
    smr.FillCoarseInBndryCC(du, dcu)
    L.akref_smr_fill_coarse_cc(h, P(u), P(cu))
    assert np.array_equal(dcu.cpu().numpy()[needs], cu[needs]), "fill_coarse_cc"
    smr.ProlongateCC(du, dcu)
    L.akref_smr_prolong_cc(h, P(u), P(cu))
    same(du, u, "prolong_cc")
    # --- face-centred: SendB/RecvB, FillCoarseInBndryFC, ProlongateFC
    b, cb = faces(n3, n2, n1), faces(c3, c2, c1)
    db, dcb = dev_faces(b, n3, n2, n1), dev_faces(cb, c3, c2, c1)
    smr.PackAndSendFC(db, dcb)
    smr.RecvAndUnpackFC(db, dcb)
    L.akref_smr_send_fc(h, P(b[0]), P(b[1]), P(b[2]), P(cb[0]), P(cb[1]), P(cb[2]))
    L.akref_smr_recv_fc(h, P(b[0]), P(b[1]), P(b[2]), P(cb[0]), P(cb[1]), P(cb[2]))
    for q, (x, y) in enumerate(zip((db.x1f, db.x2f, db.x3f), b)):
        same(x, y, "exchange_fc b%d" % q)
    for q, (x, y) in enumerate(zip((dcb.x1f, dcb.x2f, dcb.x3f), cb)):
        same(x, y, "exchange_fc cb%d" % q)
    smr.FillCoarseInBndryFC(db, dcb)
    L.akref_smr_fill_coarse_fc(h, P(b[0]), P(b[1]), P(b[2]), P(cb[0]), P(cb[1]), P(cb[2]))
    for q, (x, y) in enumerate(zip((dcb.x1f, dcb.x2f, dcb.x3f), cb)):
        assert np.array_equal(x.cpu().numpy()[needs], y[needs]), "fill_coarse_fc cb%d" % q
    smr.ProlongateFC(db, dcb)
    L.akref_smr_prolong_fc(h, P(b[0]), P(b[1]), P(b[2]), P(cb[0]), P(cb[1]), P(cb[2]))
    for q, (x, y) in enumerate(zip((db.x1f, db.x2f, db.x3f), b)):
        same(x, y, "prolong_fc b%d" % q)
    # --- restricted fluxes (face-shaped as MHD's, cell-shaped as hydro's)
    for fs in (1, 0):
        fl = [_rand(rng, (nmb, nvar, n3, n2, n1 + fs)), _rand(rng, (nmb, nvar, n3, n2 + fs, n1)),
              _rand(rng, (nmb, nvar, n3 + fs, n2, n1))]
        dfl = FaceFld(nmb, nvar, n3, n2, n1, "cuda", face_shaped=bool(fs))
        for dst, a in zip((dfl.x1f, dfl.x2f, dfl.x3f), fl):
            dst.copy_(torch.from_numpy(a))
        smr.PackAndSendFluxCC(dfl, bool(fs))
        smr.RecvAndUnpackFluxCC(dfl, bool(fs))
        L.akref_smr_flux_cc(h, P(fl[0]), P(fl[1]), P(fl[2]), fs)
        for q, (x, y) in enumerate(zip((dfl.x1f, dfl.x2f, dfl.x3f), fl)):
            same(x, y, "flux_cc fs=%d dir %d" % (fs, q))
    # --- edge EMFs: sum over same-level owners, zero + sum of finer, average
    e = [_rand(rng, (nmb, n3 + 1, n2 + 1, n1)), _rand(rng, (nmb, n3 + 1, n2, n1 + 1)), _rand(rng, (nmb, n3, n2 + 1, n1 + 1))]
    de = EdgeFld(nmb, n3, n2, n1, "cuda")
    for dst, a in zip((de.x1e, de.x2e, de.x3e), e):
        dst.copy_(torch.from_numpy(a))
    smr.PackAndSendFluxFC(de)
    smr.RecvAndUnpackFluxFC(de)
    L.akref_smr_flux_fc(h, P(e[0]), P(e[1]), P(e[2]))
    for q, (x, y) in enumerate(zip((de.x1e, de.x2e, de.x3e), e)):
        same(x, y, "emf_exchange e%d" % (q + 1))
    L.akref_smr_destroy(h)


@pytest.mark.parametrize("mesh", ["3d", "3d_3lev_ng4", "2d"])
def test_work_lists_hold_exactly_the_pairs_their_kernels_accept(mesh):
    """akmi_smr_build_lists (round 3): each list must contain every (block, slot) pair the kernels launched over it
    would accept -- a missing pair would silently skip work -- and nothing but pairs of valid slots.  Recomputed here
    from the neighbour table with the predicates written out again."""
    import torch
    pm, smr, L, h, keep = _setup(*OPS_MESHES[mesh], nvar=5)
    assert smr.smr_c.lists, "the host did not build work lists"
    nmb = pm.nmb_total
    ng = smr.t_nghbr.cpu().numpy().reshape(nmb, 56, 3)
    lev = smr.t_lev.cpu().numpy()
    needs = smr.t_needs.cpu().numpy()
    lists = smr.t_lists.cpu().numpy().reshape(6, nmb*56, 2)
    cnt = [int(smr.smr_c.list_cnt[q]) for q in range(6)]
    nn = smr.nnghbr
    want = [set() for _ in range(6)]
    for m in range(nmb):
        for n in range(nn):
            gid, nl = int(ng[m, n, 0]), int(ng[m, n, 1])
            ml = int(lev[m])
            if gid >= 0:
                want[0].add((m, n))
                if not (smr.direct_same and nl == ml and gid < nmb):
                    want[1].add((m, n))
                if nl < ml:
                    want[2].add((m, n))
                if nl > ml:
                    want[3].add((m, n))
                if nl == ml and needs[m]:
                    want[4].add((m, n))
            if n in (0, 4, 8, 12, 24, 28) or (16 <= n < 24 and n % 2 == 0) or (32 <= n < 48 and n % 2 == 0):
                want[5].add((m, n))
    for q in range(6):
        got = [tuple(int(x) for x in lists[q, e]) for e in range(cnt[q])]
        assert len(got) == len(set(got)), "list %d repeats a pair" % q
        assert set(got) == want[q], (q, len(got), len(want[q]))
    assert cnt[2] > 0 and cnt[3] > 0                  # the meshes do have level boundaries
    L.akref_smr_destroy(h)
