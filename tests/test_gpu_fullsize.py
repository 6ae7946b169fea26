"""gpu: BASELINE.json's FULL sizes (C2 hydro 128^3, C3 MHD 256^3).

Two kinds of evidence.  (1) The CPU oracle itself at full size: two cycles of C3 (Orszag-Tang 256^3, fused
stage, through the Python host and through the C++ host) and of C2 (Sod 128^3) on identical injected initial data,
the oracle on all the cores the box grants -- bit equality of the conserved variables and face fields and the same
(time, dt).  The kernels pick other chunk lengths and tile shapes at these extents (march_len, ct_tile) than at
the fixture sizes, so this closes the gap a common-mode error of both HIP paths at large extents would pass
through.  (2) Properties that do not depend on the size, over more cycles than the oracle affords: the fused
stage kernels and the task-granular kernels must agree bit for bit; the result must not depend on the MeshBlock
decomposition (one 256^3 block vs eight 128^3 blocks exchanging ghost zones); mass, momentum and total energy
of a periodic box are conserved to round-off; div B stays at round-off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import parity_util as pu  # noqa: E402


def _run(problem, n, mb, cycles, fused, **kw):
    import torch
    from athenak_amd.main import Simulation, load_deck
    deck, ov = pu.deck_overrides(problem, n, 3, mb, **kw)
    pin = load_deck(deck, ov)
    blk = "mhd" if pin.DoesBlockExist("mhd") else "hydro"
    pin.blocks[blk]["fused_stage"] = "true" if fused else "false"
    sim = Simulation(pin)
    ph = sim.phys
    ind = sim.pmesh.mb_indcs
    a = (slice(None), slice(None), slice(ind.ks, ind.ke + 1), slice(ind.js, ind.je + 1),
         slice(ind.is_, ind.ie + 1))
    tot0 = ph.u0[a].sum(dim=(0, 2, 3, 4)).cpu().numpy()
    assert sim.Execute(max_cycles=cycles) == cycles
    torch.cuda.synchronize()
    return sim, tot0, ph.u0[a].sum(dim=(0, 2, 3, 4)).cpu().numpy(), a


def _global(sim, t):
    """active cells of a cell-centred (nmb, nvar, ...) tensor assembled into the global mesh"""
    import torch
    pm = sim.pmesh
    ind = pm.mb_indcs
    nv = t.shape[1]
    out = torch.empty((nv, pm.mesh_indcs.nx3, pm.mesh_indcs.nx2, pm.mesh_indcs.nx1), dtype=t.dtype,
                      device=t.device)
    for m, l in enumerate(pm.lloc_eachmb):
        out[:, l[2]*ind.nx3:(l[2] + 1)*ind.nx3, l[1]*ind.nx2:(l[1] + 1)*ind.nx2,
            l[0]*ind.nx1:(l[0] + 1)*ind.nx1] = t[m][:, ind.ks:ind.ke + 1, ind.js:ind.je + 1, ind.is_:ind.ie + 1]
    return out


def _divb(sim):
    ph, ind = sim.phys, sim.pmesh.mb_indcs
    k, j, i = slice(ind.ks, ind.ke + 1), slice(ind.js, ind.je + 1), slice(ind.is_, ind.ie + 1)
    k1, j1, i1 = slice(ind.ks + 1, ind.ke + 2), slice(ind.js + 1, ind.je + 2), slice(ind.is_ + 1, ind.ie + 2)
    dx = sim.pmesh.pmb_pack.pmb.dx[0]
    d = ((ph.b0.x1f[:, k, j, i1] - ph.b0.x1f[:, k, j, i])/dx[0] +
         (ph.b0.x2f[:, k, j1, i] - ph.b0.x2f[:, k, j, i])/dx[1] +
         (ph.b0.x3f[:, k1, j, i] - ph.b0.x3f[:, k, j, i])/dx[2])
    return float(d.abs().max())


def _oracle_threads():
    """all the CPUs the container may use (cgroup quota), for the oracle's OpenMP loops"""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        t = open("/sys/fs/cgroup/cpu.max").read().split()
        if t[0] != "max":
            n = max(1, min(n, int(round(int(t[0])/int(t[1])))))
    except (OSError, ValueError, IndexError):
        pass
    pu.akref.lib().akref_set_threads(n)
    return n


@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
def test_c3_mhd_256_against_the_oracle(native):
    """C3 at FULL size against the CPU oracle: Orszag-Tang 256^3, PLM+HLLD+CT, RK2, cfl 0.3, one MeshBlock, fused
    stage, 2 cycles (4 stages, the first of each cycle out of place), identical injected initial data"""
    import gc
    import torch
    _oracle_threads()
    try:
        r = pu.compare_run("orszag_tang", 256, 3, 256, cycles=2, fused=True, native=native, cfl=0.3)
    finally:
        pu.akref.lib().akref_set_threads(1)
    assert r["cycles"] == 2
    assert r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1], (r["time"], r["dt"])
    assert r["bitwise_equal"], r["diffs"]
    gc.collect()
    torch.cuda.empty_cache()


def test_c2_hydro_128_against_the_oracle():
    """C2 at FULL size against the CPU oracle: Sod 128^3 single MeshBlock, PLM+HLLC, RK2, cfl 0.3 (outflow in
    x1), the one-kernel hydro stage, 2 cycles"""
    _oracle_threads()
    try:
        r = pu.compare_run("sod", 128, 3, 128, cycles=2, fused=True, cfl=0.3)
    finally:
        pu.akref.lib().akref_set_threads(1)
    assert r["cycles"] == 2
    assert r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1], (r["time"], r["dt"])
    assert r["bitwise_equal"], r["diffs"]


def test_c3_mhd_256_fused_equals_task_path_and_conserves():
    """C3: Orszag-Tang 256^3, PLM+HLLD+CT, RK2, cfl 0.3, 4 cycles"""
    import torch
    sf, t0, t1, a = _run("orszag_tang", 256, 256, 4, True, cfl=0.3)
    # periodic box: sums of the conserved variables change only by round-off of 1.7e7-term sums
    scale = np.abs(t0).max()
    assert np.all(np.abs(t1 - t0) <= 1e-12*scale*np.maximum(1.0, np.abs(t0)/scale)), (t0, t1)
    assert _divb(sf) < 1e-11
    uf = sf.phys.u0.clone()
    bf = [sf.phys.b0.x1f.clone(), sf.phys.b0.x2f.clone(), sf.phys.b0.x3f.clone()]
    tf, dtf = sf.pmesh.time, sf.pmesh.dt
    del sf
    torch.cuda.empty_cache()
    ss, _, _, _ = _run("orszag_tang", 256, 256, 4, False, cfl=0.3)
    assert (ss.pmesh.time, ss.pmesh.dt) == (tf, dtf)
    assert torch.equal(ss.phys.u0, uf)
    assert all(torch.equal(x, y) for x, y in zip((ss.phys.b0.x1f, ss.phys.b0.x2f, ss.phys.b0.x3f), bf))


def test_c3_mhd_256_is_independent_of_the_block_decomposition():
    """one 256^3 MeshBlock vs eight 128^3 and sixty-four 64^3 MeshBlocks (ghost exchange, Z-ordered
    pack, the tile shapes the kernels choose for those block sizes): same bits"""
    import torch
    s1, _, _, _ = _run("orszag_tang", 256, 256, 3, True, cfl=0.3)
    g1 = _global(s1, s1.phys.u0).clone()
    w1 = _global(s1, s1.phys.bcc0).clone()
    t1 = (s1.pmesh.time, s1.pmesh.dt)
    del s1
    torch.cuda.empty_cache()
    for mb in (128, 64):
        s8, _, _, _ = _run("orszag_tang", 256, mb, 3, True, cfl=0.3)
        assert (s8.pmesh.time, s8.pmesh.dt) == t1
        assert torch.equal(_global(s8, s8.phys.u0), g1)
        assert torch.equal(_global(s8, s8.phys.bcc0), w1)
        del s8
        torch.cuda.empty_cache()


def test_hydro_256_one_kernel_stage_equals_task_path_and_small_blocks():
    """hydro PLM+HLLC 256^3 (the bench's --problem sod): the one-kernel stage == the task-granular
    kernels == 64 MeshBlocks of 64^3"""
    import torch
    sf, _, _, _ = _run("sod", 256, 256, 3, True, cfl=0.3)
    uf, wf = sf.phys.u0.clone(), sf.phys.w0.clone()
    gf = _global(sf, sf.phys.u0).clone()
    tf = (sf.pmesh.time, sf.pmesh.dt)
    del sf
    torch.cuda.empty_cache()
    ss, _, _, _ = _run("sod", 256, 256, 3, False, cfl=0.3)
    assert (ss.pmesh.time, ss.pmesh.dt) == tf
    assert torch.equal(ss.phys.u0, uf) and torch.equal(ss.phys.w0, wf)
    del ss
    torch.cuda.empty_cache()
    s64, _, _, _ = _run("sod", 256, 64, 3, True, cfl=0.3)
    assert (s64.pmesh.time, s64.pmesh.dt) == tf
    assert torch.equal(_global(s64, s64.phys.u0), gf)


def test_c2_hydro_128_fused_equals_task_path():
    """C2: Sod 128^3, PLM+HLLC, RK2, cfl 0.3 (outflow in x1): fused == task-granular, both hosts'
    decomposition 1 x 128^3 == 8 x 64^3"""
    import torch
    sf, _, _, _ = _run("sod", 128, 128, 6, True, cfl=0.3)
    ss, _, _, _ = _run("sod", 128, 128, 6, False, cfl=0.3)
    assert (ss.pmesh.time, ss.pmesh.dt) == (sf.pmesh.time, sf.pmesh.dt)
    assert torch.equal(ss.phys.u0, sf.phys.u0) and torch.equal(ss.phys.w0, sf.phys.w0)
    s8, _, _, _ = _run("sod", 128, 64, 6, True, cfl=0.3)
    assert torch.equal(_global(s8, s8.phys.u0), _global(sf, sf.phys.u0))
    # mass leaves only through the x1 outflow faces; the transverse momenta stay exactly zero
    assert float(sf.phys.u0[:, 2:4].abs().max()) == 0.0


@pytest.mark.parametrize("native", [False, True], ids=["py", "cpp"])
def test_c5_blast_smr_at_deck_size_is_bit_identical(native):
    """BASELINE config 5 as the deck ships it -- 64^3 root grid in 16^3 MeshBlocks, the centre refined by one
    level (120 blocks on two levels), PPM4 + HLLD + CT, ng = 4 -- is small enough for the oracle: the HIP path
    through either host equals it bit for bit after 4 cycles, with the same dt sequence; div B stays at
    round-off on every block and mass / energy are conserved across the fine/coarse faces"""
    import torch
    r = pu.compare_run("blast_smr", (64, 64, 64), 3, (16, 16, 16), cycles=4, native=native, keep=True)
    assert r["cycles"] == 4 and r["bitwise_equal"], r.get("diffs")
    assert r["time"][0] == r["time"][1] and r["dt"][0] == r["dt"][1]
    sim = r["sim"]
    pm = sim.pmesh
    assert pm.nmb_total == 120 and sorted(set(int(l) for l in pm.pmb_pack.pmb.mb_lev)) == [pm.root_level, pm.root_level + 1]
    ph, ind = sim.phys, pm.mb_indcs
    k, j, i = slice(ind.ks, ind.ke + 1), slice(ind.js, ind.je + 1), slice(ind.is_, ind.ie + 1)
    k1, j1, i1 = slice(ind.ks + 1, ind.ke + 2), slice(ind.js + 1, ind.je + 2), slice(ind.is_ + 1, ind.ie + 2)
    dx = torch.as_tensor(np.asarray(pm.pmb_pack.pmb.dx), device="cuda")          # per block
    d = ((ph.b0.x1f[:, k, j, i1] - ph.b0.x1f[:, k, j, i])/dx[:, 0, None, None, None] +
         (ph.b0.x2f[:, k, j1, i] - ph.b0.x2f[:, k, j, i])/dx[:, 1, None, None, None] +
         (ph.b0.x3f[:, k1, j, i] - ph.b0.x3f[:, k, j, i])/dx[:, 2, None, None, None])
    assert float(d.abs().max()) <= 2e-11          # test_nr_divb_amr_mpicpu.py:38-40
    vol = (dx[:, 0]*dx[:, 1]*dx[:, 2])[:, None]
    tot = (ph.u0[:, :, k, j, i].sum(dim=(2, 3, 4))*vol).sum(dim=0).cpu().numpy()
    tot0 = r["totals0"]
    assert np.all(np.abs(tot - tot0) <= 1e-12*np.maximum(1.0, np.abs(tot0))), (tot, tot0)


@pytest.mark.parametrize("mhd", [False, True], ids=["hydro", "mhd"])
def test_paired_c2p_newdt_with_floors_against_the_oracle(mhd):
    """akmi_*_c2p_newdt at a size that takes the two-cells-per-thread kernel (>= 4 M cells per launch: one MeshBlock of
    160^3 + ghosts), random states with a tenth of the cells below the density or the pressure floor: u0, w0, bcc0, the
    three floor counters and the three CFL minima against the oracle's ConsToPrim + NewTimeStep"""
    import ctypes as C
    import torch
    from athenak_amd import capi
    from oracle import akref
    L, R = capi.lib(), akref.lib()
    rng = np.random.default_rng(41)
    nx, ng = 160, 2
    N = nx + 2*ng
    pk, dx = akref.make_pack(1, nx, nx, nx, ng, np.full((1, 3), 1.0/nx), 5.0/3.0)
    pk.dfloor, pk.pfloor = 0.3, 0.2
    dxd = torch.from_numpy(dx.copy()).cuda()
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    u = rng.uniform(0.5, 2.0, size=(1, 5, N, N, N))
    u[:, 1:4] = rng.normal(size=u[:, 1:4].shape)
    b = [rng.normal(size=(1, N, N, N + 1)), rng.normal(size=(1, N, N + 1, N)), rng.normal(size=(1, N + 1, N, N))] if mhd else []
    emag = 0.0
    if mhd:
        emag = 0.5*((0.5*(b[0][..., :-1] + b[0][..., 1:]))**2 + (0.5*(b[1][:, :, :-1] + b[1][:, :, 1:]))**2 +
                    (0.5*(b[2][:, :-1] + b[2][:, 1:]))**2)
    u[:, 4] = 0.5*(u[:, 1]**2 + u[:, 2]**2 + u[:, 3]**2)/u[:, 0] + emag + rng.uniform(0.5, 3.0, size=u[:, 4].shape)
    low = rng.random(size=u[:, 0].shape)
    u[:, 0][low < 0.05] = 0.1
    u[:, 4][(low > 0.05) & (low < 0.1)] = 0.01
    w, bcc = np.zeros_like(u), np.zeros((1, 3, N, N, N))
    cnt, dt3 = np.zeros(3, dtype=np.int32), np.zeros(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ud, wd, bd, bccd, cntd, dtd = t(u), t(w), [t(x) for x in b], t(bcc), t(cnt), t(dt3)
    if mhd:
        R.akref_mhd_c2p_newdt(C.byref(pk), akref.ptr(u), *[akref.ptr(x) for x in b], akref.ptr(w), akref.ptr(bcc), 1,
                              akref.ptr(cnt), akref.ptr(dt3))
        capi.check(L.akmi_mhd_c2p_newdt(C.byref(pkd), capi._p(ud), *[capi._p(x) for x in bd], capi._p(wd), capi._p(bccd), 1,
                                        capi._p(cntd), capi._p(dtd), None), "mhd_c2p_newdt")
        assert np.array_equal(bcc, bccd.cpu().numpy())
    else:
        R.akref_hydro_c2p_newdt(C.byref(pk), akref.ptr(u), akref.ptr(w), 1, akref.ptr(cnt), akref.ptr(dt3))
        capi.check(L.akmi_hydro_c2p_newdt(C.byref(pkd), capi._p(ud), capi._p(wd), 1, capi._p(cntd), capi._p(dtd), None),
                   "hydro_c2p_newdt")
    assert np.array_equal(u, ud.cpu().numpy()) and np.array_equal(w, wd.cpu().numpy())
    assert cnt.sum() > 1000 and np.array_equal(cnt, cntd.cpu().numpy()), (cnt, cntd)
    assert np.array_equal(dt3, dtd.cpu().numpy()), (dt3, dtd)
