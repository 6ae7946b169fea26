"""Output formats: tab, hst, bin and -errs.dat files.

not gpu:
  * the writers, driven by the CPU oracle through the product's host logic, reproduce the
    committed fixtures tests/golden/outputs/ byte for byte;
  * (only where /root/reference exists, i.e. in the build container) the fixtures are parsed
    with the REFERENCE's own readers -- vis/python/athena_read.py tab()/hst()/error_dat() and
    bin_convert.read_binary() -- and the parsed numbers are compared with the oracle's arrays.
    This pins the formats on the reference's tools.
gpu:
  * the same cases through the HIP path (python -m athenak_amd ...) give byte-identical tab, bin
    and -errs.dat files and the same history numbers (device reduction: round-off).
"""
import filecmp
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import output_cases as oc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "outputs")
CASES = ["sod", "ot", "lwave_hydro", "lwave_mhd"]
REF_PY = "/root/reference/vis/python"


def _golden_files(case):
    out = []
    for root, _, files in os.walk(os.path.join(GOLD, case)):
        out += [os.path.relpath(os.path.join(root, f), os.path.join(GOLD, case)) for f in files]
    return sorted(out)


@pytest.fixture
def cpu_oracle_backend():
    import cpu_backend
    cpu_backend.install()
    yield
    cpu_backend.uninstall()


@pytest.mark.parametrize("case", CASES)
def test_writers_reproduce_fixtures(case, cpu_oracle_backend):
    with tempfile.TemporaryDirectory() as d:
        files = oc.run_case(case, d, fused=False)
        assert files == _golden_files(case)
        for rel in files:
            assert filecmp.cmp(os.path.join(d, rel), os.path.join(GOLD, case, rel), shallow=False), rel


def _oracle_sod():
    from oracle import akref
    s = akref.Sim(nx1=64, nx2=1, nx3=1, mb_nx1=32, mb_nx2=1, mb_nx3=1, ng=2, x1min=-0.5, x1max=0.5,
                  x2min=-0.5, x2max=0.5, x3min=-0.5, x3max=0.5,
                  bcs=["outflow", "outflow", "periodic", "periodic", "periodic", "periodic"], nstages=2,
                  cfl=0.8, tlim=0.1, nlim=-1, is_mhd=0, recon="plm", rsolver="hllc", gamma=1.4,
                  pgen="shock_tube", shock_dir=1, xshock=0.0, wl=[1.0, 0, 0, 0, 1.0, 0, 0, 0],
                  wr=[0.125, 0, 0, 0, 0.1, 0, 0, 0])
    s.initialize()
    s.run()
    return s


@pytest.mark.skipif(not os.path.isdir(REF_PY), reason="reference readers only exist in the build container")
def test_reference_readers_parse_the_files():
    sys.path.insert(0, REF_PY)
    import types
    import athena_read
    if "h5py" not in sys.modules:            # bin_convert imports h5py for its athdf writer only;
        try:                                 # read_binary() does not use it (not installed here)
            import h5py  # noqa: F401
        except ImportError:
            sys.modules["h5py"] = types.ModuleType("h5py")
    import bin_convert
    # --- tab: final Sod primitives, columns as athena_read.tab names them
    t = athena_read.tab(os.path.join(GOLD, "sod", "tab", "Sod.hydro_w.00002.tab"))
    o = _oracle_sod()
    w = o.array("w0")
    assert t["cycle"] == o.ncycle and abs(t["time"] - 0.1) < 1e-6
    r = lambda a: np.array([float("%12.5e" % v) for v in a])
    dens = np.concatenate([w[m, 0, 0, 0, 2:34] for m in range(2)])
    eint = np.concatenate([w[m, 4, 0, 0, 2:34] for m in range(2)])
    assert np.array_equal(t["dens"], r(dens)) and np.array_equal(t["eint"], r(eint))
    x = -0.5 + (np.arange(64) + 0.5)/64
    assert np.allclose(t["x1v"], x, atol=1e-6) and set(t) >= {"dens", "velx", "vely", "velz", "eint", "x1v"}
    # --- hst: names and conservation
    h = athena_read.hst(os.path.join(GOLD, "sod", "Sod.hydro.hst"))
    assert list(h)[:5] == ["time", "dt", "mass", "1-mom", "2-mom"] and "3-KE" in h
    assert np.all(h["mass"] == h["mass"][0]) and np.all(h["tot-E"] == h["tot-E"][0])
    assert h["time"][-1] == pytest.approx(0.1, rel=1e-5) and len(h["time"]) == 5
    hm = athena_read.hst(os.path.join(GOLD, "ot", "OrszagTang.mhd.hst"))
    assert "3-ME" in hm and len(hm["time"]) == 4 and np.allclose(hm["mass"], hm["mass"][0], rtol=1e-12)
    # --- errs.dat: column 4 = RMS-L1 (testutils.py:273-276), pinned numbers of the oracle tests
    e = athena_read.error_dat(os.path.join(GOLD, "lwave_hydro", "LinWave-errs.dat"))
    assert e.shape == (1, 11) and "%.6e" % e[0][4] == "7.390252e-08" and int(e[0][0]) == 32
    e = athena_read.error_dat(os.path.join(GOLD, "lwave_mhd", "LinWave-errs.dat"))
    assert e.shape == (1, 14) and "%.6e" % e[0][4] == "8.812266e-08"
    # --- bin: the reference's reader returns per-MeshBlock float32 arrays
    b = bin_convert.read_binary(os.path.join(GOLD, "ot", "bin", "OrszagTang.mhd_bcc.00000.bin"))
    assert b["var_names"] == ["bcc1", "bcc2", "bcc3"] and b["n_mbs"] == 4 and b["cycle"] == 0
    assert (b["Nx1"], b["Nx2"], b["Nx3"], b["nx1_mb"], b["nx3_mb"]) == (16, 16, 8, 8, 8)
    assert b["mb_data"]["bcc1"][0].shape == (8, 8, 8)
    from oracle import akref
    o2 = akref.Sim(nx1=16, nx2=16, nx3=8, mb_nx1=8, mb_nx2=8, mb_nx3=8, ng=2, x1min=-0.5, x1max=0.5,
                   x2min=-0.5, x2max=0.5, x3min=-0.5, x3max=0.5, bcs=["periodic"]*6, nstages=2, cfl=0.3,
                   tlim=1.0, nlim=6, is_mhd=1, recon="plm", rsolver="hlld", gamma=1.666666667,
                   pgen="orszag_tang")
    o2.initialize()
    bcc = o2.array("bcc0")
    for m in range(4):
        for q, name in enumerate(("bcc1", "bcc2", "bcc3")):
            assert np.array_equal(b["mb_data"][name][m], bcc[m, q, 2:10, 2:10, 2:10].astype(np.float32))
    assert np.array_equal(b["mb_logical"][:, :3], np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]]))
    bs = bin_convert.read_binary(os.path.join(GOLD, "sod", "bin", "Sod.hydro_u.00001.bin"))
    assert bs["var_names"] == ["dens", "mom1", "mom2", "mom3", "ener"] and bs["n_mbs"] == 2


def _bin_parts(path):
    """(pre-header, parameter dump without the fused_stage line -- the fixtures were written by
    the task-granular host path --, binary payload)"""
    raw = open(path, "rb").read()
    k = raw.index(b"  header offset=")
    e = raw.index(b"\n", k)
    n = int(raw[k:e].split(b"=")[1])
    dump = raw[e + 1:e + 1 + n]
    dump = b"\n".join(l for l in dump.split(b"\n") if not l.startswith(b"fused_stage"))
    return raw[:k], dump, raw[e + 1 + n:]


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("case", CASES)
def test_hip_path_writes_the_same_files(case, fused):
    with tempfile.TemporaryDirectory() as d:
        files = oc.run_case(case, d, fused=fused)
        assert files == _golden_files(case)
        for rel in files:
            a, b = os.path.join(d, rel), os.path.join(GOLD, case, rel)
            if rel.endswith(".hst"):
                x, y = np.loadtxt(a), np.loadtxt(b)
                assert x.shape == y.shape and np.allclose(x, y, rtol=2e-5 if case == "sod" else 1e-11, atol=1e-15)
            elif rel.endswith(".bin"):
                assert _bin_parts(a) == _bin_parts(b), rel
            else:
                assert filecmp.cmp(a, b, shallow=False), rel


@pytest.mark.gpu
@pytest.mark.parametrize("is_mhd", [0, 1])
def test_history_sums_kernel(is_mhd):
    """akmi_history_sums against its sequential oracle twin on an evolved state"""
    import ctypes as C
    import torch
    import parity_util as pu
    from athenak_amd import capi
    from athenak_amd.main import load_deck
    from oracle import akref
    deck, ov = pu.deck_overrides("orszag_tang" if is_mhd else "sod", 24, 3, 12, cfl=0.3)
    o = akref.Sim(**pu.oracle_kwargs(load_deck(deck, ov)))
    o.initialize()
    for _ in range(3):
        o.step()
    pk = o.pack()
    dxd = torch.from_numpy(o.array("dx").copy()).cuda()
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    nh = 11 if is_mhd else 8
    ref = np.zeros(nh)
    names = ["u0"] + (["b0x1f", "b0x2f", "b0x3f"] if is_mhd else [])
    h = {k: o.array(k).copy() for k in names}
    args = [akref.ptr(h[k]) for k in names] + ([] if is_mhd else [None, None, None])
    akref.lib().akref_history_sums(C.byref(pk), is_mhd, *args, akref.ptr(ref))
    dv = {k: torch.from_numpy(v).cuda() for k, v in h.items()}
    out = torch.full((nh,), 7.0, dtype=torch.float64, device="cuda")
    dargs = [capi._p(dv[k]) for k in names] + ([] if is_mhd else [None, None, None])
    capi.check(capi.lib().akmi_history_sums(C.byref(pkd), is_mhd, *dargs, capi._p(out), None), "hist")
    got = out.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.all(np.abs(got - ref) <= 1e-12*scale), (got, ref)


# ---- restart files (src/outputs/restart.cpp, read back by Mesh::BuildTreeFromRestart + the restart
# constructor of ProblemGenerator) ------------------------------------------------------------------
def _restart_roundtrip(name, fused):
    """uninterrupted run with an rst block vs a run restarted from the middle dump (-r): same tab
    files byte for byte, same bin payload, same final time/cycle"""
    import struct
    deck = (oc.SOD_DECK if name == "sod" else oc.OT_DECK).replace("FUSED", "true" if fused else "false")
    deck += "<output4>\nfile_type = rst\n" + ("dt = 0.05\n" if name == "sod" else "dcycle = 3\n")
    from athenak_amd.__main__ import main
    here = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        try:
            a, b = os.path.join(d, "full"), os.path.join(d, "resumed")
            os.makedirs(a)
            with open(os.path.join(a, "deck.athinput"), "w") as f:
                f.write(deck)
            assert main(["-i", os.path.join(a, "deck.athinput"), "-d", a]) == 0
            base = "Sod" if name == "sod" else "OrszagTang"
            rst = sorted(os.listdir(os.path.join(a, "rst")))
            # initial dump, one in the middle (t = 0.05 / cycle 3), [cycle 6,] final dump of Driver::Finalize
            assert rst == ["%s.%05d.rst" % (base, q) for q in range(3 if name == "sod" else 4)]
            # layout (restart.cpp:203-246): dump text, then nmb_total, root_level, RegionSize, 2 x RegionIndcs,
            # time, dt, ncycle, LogicalLocation[nmb], float cost[nmb], uint64 data_size, records
            blob = open(os.path.join(a, "rst", rst[1]), "rb").read()
            p = blob.index(b"<par_end>\n") + len(b"<par_end>\n")
            nmb, lev = struct.unpack_from("<ii", blob, p)
            ng, nx1, nx2, nx3 = struct.unpack_from("<4i", blob, p + 8 + 72 + 76)
            t, dt, ncyc = struct.unpack_from("<ddi", blob, p + 8 + 72 + 152)
            (dsz,) = struct.unpack_from("<Q", blob, p + 252 + 20*nmb)
            n1, n2, n3 = nx1 + 2*ng, (nx2 + 2*ng if nx2 > 1 else 1), (nx3 + 2*ng if nx3 > 1 else 1)
            if name == "sod":
                assert (nmb, lev, ng, nx1) == (2, 1, 2, 32) and dsz == 8*5*n1
            else:
                assert (nmb, lev, nx1, nx2, nx3, ncyc) == (4, 1, 8, 8, 8, 3)
                assert dsz == 8*(5*n3*n2*n1 + n3*n2*(n1 + 1) + n3*(n2 + 1)*n1 + (n3 + 1)*n2*n1)
            assert len(blob) == p + 252 + 20*nmb + 8 + dsz*nmb and t > 0.0 and dt > 0.0
            os.chdir(here)
            assert main(["-r", os.path.join(a, "rst", rst[1]), "-d", b]) == 0
        finally:
            os.chdir(here)
        # files written after the restart point exist in both runs and agree
        tabs = sorted(os.listdir(os.path.join(b, "tab")))
        assert tabs and set(tabs) <= set(os.listdir(os.path.join(a, "tab")))
        for fn in tabs:
            assert filecmp.cmp(os.path.join(a, "tab", fn), os.path.join(b, "tab", fn), shallow=False), fn
        bins = sorted(os.listdir(os.path.join(b, "bin")))
        assert bins
        for fn in bins:
            pa, pb = _bin_parts(os.path.join(a, "bin", fn)), _bin_parts(os.path.join(b, "bin", fn))
            assert pa[0] == pb[0] and pa[2] == pb[2], fn
        # the last restart dump of both runs holds the same state
        ra = open(os.path.join(a, "rst", rst[-1]), "rb").read()
        rb = open(os.path.join(b, "rst", rst[-1]), "rb").read()
        qa, qb = ra.index(b"<par_end>\n"), rb.index(b"<par_end>\n")
        assert ra[qa:] == rb[qb:]


@pytest.mark.parametrize("case", ["sod", "ot"])
def test_restart_files_roundtrip(case, cpu_oracle_backend):
    _restart_roundtrip(case, fused=False)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("case", ["sod", "ot"])
def test_restart_files_roundtrip_hip(case, fused):
    _restart_roundtrip(case, fused)


# ---- refined meshes: the block headers carry the block's OWN level (round 3) ---------------------------------
def _run_smr(workdir, extra):
    """2-D hydro linear wave on the statically refined mesh of linear_wave_hydro_smr.athinput, four cycles, a
    bin dump of the conserved variables and a restart dump every two cycles"""
    from athenak_amd.__main__ import main
    here = os.getcwd()
    text = open(os.path.join(ROOT, "athenak_amd", "inputs", "linear_wave_hydro_smr.athinput")).read()
    text += "<output1>\nfile_type = bin\nvariable = hydro_u\ndcycle = 2\n<output2>\nfile_type = rst\ndcycle = 2\n"
    deck = os.path.join(workdir, "smr.athinput")
    os.makedirs(workdir, exist_ok=True)
    with open(deck, "w") as f:
        f.write(text)
    try:
        rc = main(["-i", deck, "-d", workdir, "mesh/nx1=32", "mesh/nx2=16", "mesh/nx3=1",
                   "meshblock/nx1=8", "meshblock/nx2=4", "meshblock/nx3=1", "refined_region1/x1min=1.2",
                   "refined_region1/x1max=1.8", "refined_region1/x2min=0.6", "refined_region1/x2max=0.9",
                   "time/nlim=4", "hydro/fused_stage=false"] + list(extra))
        assert rc == 0
    finally:
        os.chdir(here)


def _bin_block_headers(path):
    """the 10 int32 + 6 float64 header of every MeshBlock record of a version-1.1 bin file"""
    import struct
    raw = open(path, "rb").read()
    k = raw.index(b"  header offset=")
    e = raw.index(b"\n", k)
    n = int(raw[k:e].split(b"=")[1])
    nvar = int(raw[:k].split(b"number of variables=")[1].split(b"\n")[0])
    pos, out = e + 1 + n, []
    while pos < len(raw):
        h = struct.unpack_from("<10i", raw, pos)
        out.append(h)
        nx = [h[1] - h[0] + 1, h[3] - h[2] + 1, h[5] - h[4] + 1]
        pos += 40 + 48 + 4*nvar*nx[0]*nx[1]*nx[2]
    assert pos == len(raw)
    return out


def test_refined_mesh_bin_and_rst_carry_block_levels(cpu_oracle_backend):
    """binary.cpp:192-193 writes loc.level - root_level per block, restart.cpp:115-124 the LogicalLocation with
    its level; a dump of a refined mesh that stores the root level for every block is placed at the wrong
    resolution by bin_convert / a restarted run"""
    from athenak_amd import outputs
    from athenak_amd.main import load_deck
    from athenak_amd.mesh import Mesh
    with tempfile.TemporaryDirectory() as d:
        _run_smr(d, [])
        pin = load_deck("linear_wave_hydro_smr.athinput", [
            "mesh/nx1=32", "mesh/nx2=16", "mesh/nx3=1", "meshblock/nx1=8", "meshblock/nx2=4", "meshblock/nx3=1",
            "refined_region1/x1min=1.2", "refined_region1/x1max=1.8", "refined_region1/x2min=0.6",
            "refined_region1/x2max=0.9"])
        pm = Mesh(pin)
        levels = [pm.level_of(g) for g in range(pm.nmb_total)]
        assert pm.multilevel and len(set(levels)) == 2, levels
        hdrs = _bin_block_headers(os.path.join(d, "bin", "LinWave.hydro_u.00001.bin"))
        assert len(hdrs) == pm.nmb_total
        for g, h in enumerate(hdrs):
            assert tuple(h[6:9]) == tuple(pm.lloc_eachmb[g][:3])
            assert h[9] == levels[g] - pm.root_level, (g, h[9], levels[g], pm.root_level)
        if os.path.isdir(REF_PY):                  # the reference's own reader (build container only)
            sys.path.insert(0, REF_PY)
            import types
            if "h5py" not in sys.modules:
                try:
                    import h5py  # noqa: F401
                except ImportError:
                    sys.modules["h5py"] = types.ModuleType("h5py")
            import bin_convert
            b = bin_convert.read_binary(os.path.join(d, "bin", "LinWave.hydro_u.00001.bin"))
            assert b["n_mbs"] == pm.nmb_total
            assert [int(x) for x in b["mb_logical"][:, 3]] == [l - pm.root_level for l in levels]
        _, hdr, _ = outputs.read_restart(os.path.join(d, "rst", "LinWave.00001.rst"))
        assert hdr["root_level"] == pm.root_level
        assert [int(x) for x in hdr["lloc"][:, 3]] == levels
        assert [tuple(int(x) for x in l[:3]) for l in hdr["lloc"]] == [tuple(l)[:3] for l in pm.lloc_eachmb]


def test_refined_mesh_restart_roundtrip(cpu_oracle_backend):
    """a run restarted from the middle dump of a refined-mesh run writes the same final restart file"""
    from athenak_amd.__main__ import main
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        _run_smr(d1, [])
        here = os.getcwd()
        try:
            rc = main(["-r", os.path.join(d1, "rst", "LinWave.00001.rst"), "-d", d2])
            assert rc == 0
        finally:
            os.chdir(here)
        a = open(os.path.join(d1, "rst", "LinWave.00002.rst"), "rb").read()
        b = open(os.path.join(d2, "rst", "LinWave.00002.rst"), "rb").read()
        assert a == b
