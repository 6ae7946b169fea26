"""SMR/AMR operators between a MeshBlock and its coarse buffer: RestrictCC/FC,
ProlongCC, ProlongFCShared*, ProlongFCInternal.

not gpu: properties of the oracle's restatement that the reference relies on --
  * restriction of a prolongated block returns the coarse data (conservation of the cell averages),
  * prolongation reproduces linear data exactly where the limiter is inactive,
  * the face-field prolongation (shared faces + Toth & Roe interior) keeps div B of every fine
    cell equal to that of its coarse parent, i.e. a divergence-free coarse field stays
    divergence-free, and restricting the prolongated field returns the coarse faces.
gpu: the HIP kernels against the oracle, bit for bit, in 1-D, 2-D and 3-D."""
import ctypes as C

import numpy as np
import pytest

from oracle import akref

DIMS = {1: (16, 1, 1), 2: (16, 12, 1), 3: (12, 8, 8)}


def _setup(dims, ng=2, nmb=2, nvar=5, seed=3):
    nx1, nx2, nx3 = DIMS[dims]
    pk, dx = akref.make_pack(nmb, nx1, nx2, nx3, ng, np.ones((nmb, 3)), 1.4)
    N1, N2, N3 = nx1 + 2*ng, (nx2 + 2*ng if nx2 > 1 else 1), (nx3 + 2*ng if nx3 > 1 else 1)
    c1, c2, c3 = nx1//2 + 2*ng, (nx2//2 + 2*ng if nx2 > 1 else 1), (nx3//2 + 2*ng if nx3 > 1 else 1)
    rng = np.random.default_rng(seed)
    a = dict(pk=pk, dx=dx, ng=ng, nmb=nmb, nvar=nvar, N=(N3, N2, N1), cN=(c3, c2, c1), dims=dims,
             cis=ng, cjs=ng if dims > 1 else 0, cks=ng if dims > 2 else 0,
             cn=(nx3//2 if dims > 2 else 1, nx2//2 if dims > 1 else 1, nx1//2),
             u=rng.normal(size=(nmb, nvar, N3, N2, N1)), cu=rng.normal(size=(nmb, nvar, c3, c2, c1)),
             b=[rng.normal(size=(nmb, N3, N2, N1 + 1)), rng.normal(size=(nmb, N3, N2 + 1, N1)),
                rng.normal(size=(nmb, N3 + 1, N2, N1))],
             cb=[rng.normal(size=(nmb, c3, c2, c1 + 1)), rng.normal(size=(nmb, c3, c2 + 1, c1)),
                 rng.normal(size=(nmb, c3 + 1, c2, c1))])
    return a


def _active_box(a):
    """all active coarse cells"""
    return np.array([a["cis"], a["cis"] + a["cn"][2] - 1, a["cjs"], a["cjs"] + a["cn"][1] - 1,
                     a["cks"], a["cks"] + a["cn"][0] - 1], dtype=np.int32)


def _shared_box(a, comp):
    """coarse faces of component comp on the active coarse cells (incl. the +1 face)"""
    b = _active_box(a)
    b[2*comp + 1] += 1 if (comp == 0 or a["dims"] > comp) else 0
    return b


def _prolong_field(R, a):
    """shared faces of all components, then the interior"""
    for comp in range(3):
        R.akref_prolong_fc_shared(C.byref(a["pk"]), comp, akref.ptr(_shared_box(a, comp)),
                                  akref.ptr(a["cb"][comp]), akref.ptr(a["b"][comp]))
    R.akref_prolong_fc_internal(C.byref(a["pk"]), akref.ptr(_active_box(a)), *[akref.ptr(x) for x in a["b"]])


def _div(b, dims, sl):
    k, j, i = sl
    d = b[0][:, k, j, i.start + 1:i.stop + 1] - b[0][:, k, j, i]
    if dims > 1:
        d = d + b[1][:, k, j.start + 1:j.stop + 1, i] - b[1][:, k, j, i]
    if dims > 2:
        d = d + b[2][:, k.start + 1:k.stop + 1, j, i] - b[2][:, k, j, i]
    return d


@pytest.mark.parametrize("dims", [1, 2, 3])
def test_restriction_undoes_prolongation(dims):
    R = akref.lib()
    a = _setup(dims)
    box = _active_box(a)
    R.akref_prolong_cc(C.byref(a["pk"]), a["nvar"], akref.ptr(box), akref.ptr(a["cu"]), akref.ptr(a["u"]))
    cu2 = np.zeros_like(a["cu"])
    R.akref_restrict_cc(C.byref(a["pk"]), a["nvar"], akref.ptr(a["u"]), akref.ptr(cu2))
    act = (slice(None), slice(None), slice(box[4], box[5] + 1), slice(box[2], box[3] + 1), slice(box[0], box[1] + 1))
    assert np.abs(cu2[act] - a["cu"][act]).max() < 1e-14


@pytest.mark.parametrize("dims", [1, 2, 3])
def test_prolongation_is_exact_for_linear_data(dims):
    R = akref.lib()
    a = _setup(dims)
    c3, c2, c1 = a["cN"]
    K, J, I = np.meshgrid(np.arange(c3), np.arange(c2), np.arange(c1), indexing="ij")
    lin = 0.5 + 0.25*I + (0.125*J if dims > 1 else 0) - (0.375*K if dims > 2 else 0)
    a["cu"][:] = lin
    box = _active_box(a)
    R.akref_prolong_cc(C.byref(a["pk"]), a["nvar"], akref.ptr(box), akref.ptr(a["cu"]), akref.ptr(a["u"]))
    ng = a["ng"]
    N3, N2, N1 = a["N"]
    Kf, Jf, If = np.meshgrid(np.arange(N3), np.arange(N2), np.arange(N1), indexing="ij")
    # fine cell centres in coarse index units: coarse i covers fine (i-cis)*2+is, +1
    xi = a["cis"] + (If - ng)/2.0 - 0.25
    xj = a["cjs"] + (Jf - ng)/2.0 - 0.25 if dims > 1 else 0*Jf
    xk = a["cks"] + (Kf - ng)/2.0 - 0.25 if dims > 2 else 0*Kf
    exact = 0.5 + 0.25*xi + (0.125*xj if dims > 1 else 0) - (0.375*xk if dims > 2 else 0)
    s = (slice(None), slice(None), slice(ng, N3 - ng) if dims > 2 else slice(0, 1),
         slice(ng, N2 - ng) if dims > 1 else slice(0, 1), slice(ng, N1 - ng))
    assert np.abs(a["u"][s] - exact[s[2:]]).max() < 1e-14


@pytest.mark.parametrize("dims", [2, 3])
def test_field_prolongation_preserves_divergence(dims):
    R = akref.lib()
    a = _setup(dims)
    _prolong_field(R, a)
    ng = a["ng"]
    N3, N2, N1 = a["N"]
    fs = (slice(ng, N3 - ng) if dims > 2 else slice(0, 1), slice(ng, N2 - ng), slice(ng, N1 - ng))
    cs = (slice(a["cks"], a["cks"] + a["cn"][0]), slice(a["cjs"], a["cjs"] + a["cn"][1]),
          slice(a["cis"], a["cis"] + a["cn"][2]))
    dfine = _div(a["b"], dims, fs)            # fine faces have half the area / spacing: sum of the
    dcoarse = _div(a["cb"], dims, cs)         # 2^d children = 2^(d-1) x coarse divergence (unit spacings)
    nmb = a["nmb"]
    if dims == 2:
        child = dfine.reshape(nmb, 1, a["cn"][1], 2, a["cn"][2], 2)
        assert np.abs(child - 0.5*dcoarse[:, :, :, None, :, None]).max() < 1e-13
    else:
        child = dfine.reshape(nmb, a["cn"][0], 2, a["cn"][1], 2, a["cn"][2], 2)
        assert np.abs(child - 0.5*dcoarse[:, :, None, :, None, :, None]).max() < 1e-13
    # and restriction returns the coarse faces
    cb2 = [np.zeros_like(x) for x in a["cb"]]
    R.akref_restrict_fc(C.byref(a["pk"]), *[akref.ptr(x) for x in a["b"]], *[akref.ptr(x) for x in cb2])
    for comp in range(dims):
        bx = _shared_box(a, comp)
        s = (slice(None), slice(bx[4], bx[5] + 1), slice(bx[2], bx[3] + 1), slice(bx[0], bx[1] + 1))
        assert np.abs(cb2[comp][s] - a["cb"][comp][s]).max() < 1e-14, comp


@pytest.mark.gpu
@pytest.mark.parametrize("ng", [2, 4])
@pytest.mark.parametrize("dims", [1, 2, 3])
def test_hip_operators_match_the_oracle(dims, ng):
    import torch
    from athenak_amd import capi
    L, R = capi.lib(), akref.lib()
    a = _setup(dims, ng=ng, seed=11 + dims)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dxd = t(a["dx"])
    pkd = capi.Pack.from_buffer_copy(bytes(a["pk"]))
    pkd.dx = dxd.data_ptr()
    P, nv = C.byref(pkd), a["nvar"]
    ud, cud, bd, cbd = t(a["u"]), t(a["cu"]), [t(x) for x in a["b"]], [t(x) for x in a["cb"]]
    ibox = lambda b: t(b).cpu().numpy()           # host int32 arrays are passed by pointer
    # restriction
    cu2, cu2d = np.zeros_like(a["cu"]), torch.zeros_like(cud)
    R.akref_restrict_cc(C.byref(a["pk"]), nv, akref.ptr(a["u"]), akref.ptr(cu2))
    capi.check(L.akmi_restrict_cc(P, nv, capi._p(ud), capi._p(cu2d), None), "restrict_cc")
    assert np.array_equal(cu2, cu2d.cpu().numpy())
    cb2, cb2d = [np.zeros_like(x) for x in a["cb"]], [torch.zeros_like(x) for x in cbd]
    R.akref_restrict_fc(C.byref(a["pk"]), *[akref.ptr(x) for x in a["b"]], *[akref.ptr(x) for x in cb2])
    capi.check(L.akmi_restrict_fc(P, *[capi._p(x) for x in bd], *[capi._p(x) for x in cb2d], None), "restrict_fc")
    for x, y in zip(cb2, cb2d):
        assert np.array_equal(x, y.cpu().numpy())
    # the masked forms (round 3): blocks with mask != 0 as above, the others untouched; oracle twin with the same mask
    nmb = a["u"].shape[0]
    mask = np.array([(m % 2) for m in range(nmb)], dtype=np.uint8) if nmb > 1 else np.ones(1, dtype=np.uint8)
    maskd = t(mask)
    cu3, cu3d = np.full_like(a["cu"], -7.0), torch.full_like(cud, -7.0)
    R.akref_restrict_cc_masked(C.byref(a["pk"]), nv, akref.ptr(mask), akref.ptr(a["u"]), akref.ptr(cu3))
    capi.check(L.akmi_restrict_cc_masked(P, nv, capi._p(maskd), capi._p(ud), capi._p(cu3d), None), "restrict_cc_masked")
    assert np.array_equal(cu3, cu3d.cpu().numpy())
    sel = mask.astype(bool)
    act = (slice(None), slice(a["cks"], a["cks"] + a["cn"][0]), slice(a["cjs"], a["cjs"] + a["cn"][1]),
           slice(a["cis"], a["cis"] + a["cn"][2]))
    assert np.array_equal(cu3[sel][(slice(None),) + act], cu2[sel][(slice(None),) + act]) and np.all(cu3[~sel] == -7.0)
    cb3, cb3d = [np.full_like(x, -7.0) for x in a["cb"]], [torch.full_like(x, -7.0) for x in cbd]
    R.akref_restrict_fc_masked(C.byref(a["pk"]), akref.ptr(mask), *[akref.ptr(x) for x in a["b"]],
                               *[akref.ptr(x) for x in cb3])
    capi.check(L.akmi_restrict_fc_masked(P, capi._p(maskd), *[capi._p(x) for x in bd], *[capi._p(x) for x in cb3d],
                                         None), "restrict_fc_masked")
    for x, y in zip(cb3, cb3d):
        assert np.array_equal(x, y.cpu().numpy()) and np.all(x[~sel] == -7.0)
    # prolongation of cell-centred data: active box and a ghost-side box (what a coarser neighbour fills)
    for box in (_active_box(a), np.array([a["cis"] - ng//2, a["cis"] - 1, a["cjs"], a["cjs"] + a["cn"][1] - 1,
                                          a["cks"], a["cks"] + a["cn"][0] - 1], dtype=np.int32)):
        R.akref_prolong_cc(C.byref(a["pk"]), nv, akref.ptr(box), akref.ptr(a["cu"]), akref.ptr(a["u"]))
        capi.check(L.akmi_prolong_cc(P, nv, box.ctypes.data_as(C.c_void_p), capi._p(cud), capi._p(ud), None),
                   "prolong_cc")
        assert np.array_equal(a["u"], ud.cpu().numpy())
    # face field: shared faces of each component, then the interior
    for comp in range(3):
        box = _shared_box(a, comp)
        R.akref_prolong_fc_shared(C.byref(a["pk"]), comp, akref.ptr(box), akref.ptr(a["cb"][comp]),
                                  akref.ptr(a["b"][comp]))
        capi.check(L.akmi_prolong_fc_shared(P, comp, box.ctypes.data_as(C.c_void_p), capi._p(cbd[comp]),
                                            capi._p(bd[comp]), None), "prolong_fc_shared")
        assert np.array_equal(a["b"][comp], bd[comp].cpu().numpy()), comp
    box = _active_box(a)
    R.akref_prolong_fc_internal(C.byref(a["pk"]), akref.ptr(box), *[akref.ptr(x) for x in a["b"]])
    capi.check(L.akmi_prolong_fc_internal(P, box.ctypes.data_as(C.c_void_p), *[capi._p(x) for x in bd], None),
               "prolong_fc_internal")
    for x, y in zip(a["b"], bd):
        assert np.array_equal(x, y.cpu().numpy())
    # a box whose fine cells would lie outside the block is refused, not written
    bad = np.array([0, 1, a["cjs"], a["cjs"], a["cks"], a["cks"]], dtype=np.int32)
    assert L.akmi_prolong_cc(P, nv, bad.ctypes.data_as(C.c_void_p), capi._p(cud), capi._p(ud), None) < 0


# ---- conservation at fine/coarse faces: restricted fluxes and edge EMFs ---------------------------
def _flux_arrays(a, rng):
    N3, N2, N1 = a["N"]
    nmb, nv = a["nmb"], a["nvar"]
    flx = [rng.normal(size=(nmb, nv, N3, N2, N1 + 1)), rng.normal(size=(nmb, nv, N3, N2 + 1, N1)),
           rng.normal(size=(nmb, nv, N3 + 1, N2, N1))]
    emf = [rng.normal(size=(nmb, N3 + 1, N2 + 1, N1)), rng.normal(size=(nmb, N3 + 1, N2, N1 + 1)),
           rng.normal(size=(nmb, N3, N2 + 1, N1 + 1))]
    return flx, emf


def _face_boxes(a):
    """(dir, box) of the six coarse block faces: one face thick along dir, the active cells across"""
    out = []
    lo = [a["cis"], a["cjs"], a["cks"]]
    n = [a["cn"][2], a["cn"][1], a["cn"][0]]
    for d in range(a["dims"]):
        for side in (0, 1):
            b = [lo[0], lo[0] + n[0] - 1, lo[1], lo[1] + n[1] - 1, lo[2], lo[2] + n[2] - 1]
            f = lo[d] + (n[d] if side else 0)
            b[2*d] = b[2*d + 1] = f
            out.append((d, np.array(b, dtype=np.int32)))
    return out


def _edge_boxes(a):
    """(comp, box): the edges of component comp lying in each block face they live on"""
    out = []
    lo = [a["cis"], a["cjs"], a["cks"]]
    n = [a["cn"][2], a["cn"][1], a["cn"][0]]
    for d, fb in _face_boxes(a):
        for comp in range(3):
            if comp == d:
                continue
            b = fb.copy()
            for t in range(3):
                if t != d and t != comp and t < a["dims"]:
                    b[2*t + 1] += 1                      # edges across: one more than cells
            out.append((comp, b))
    return out


@pytest.mark.parametrize("dims", [1, 2, 3])
def test_restricted_fluxes_conserve(dims):
    """area-weighted sum of the fine fluxes through a coarse face == the restricted flux; an EMF that is
    constant along an edge restricts to itself; buffer order (t1 fastest, then t2, then variable)"""
    R = akref.lib()
    a = _setup(dims, ng=2, seed=21 + dims)
    rng = np.random.default_rng(5)
    flx, emf = _flux_arrays(a, rng)
    nv = a["nvar"]
    for d, box in _face_boxes(a):
        ext = [box[1] - box[0] + 1, box[3] - box[2] + 1, box[5] - box[4] + 1]
        out = np.zeros((a["nmb"], nv*ext[0]*ext[1]*ext[2]))
        assert R.akref_restrict_flux_cc(C.byref(a["pk"]), nv, d, akref.ptr(box), akref.ptr(flx[d]),
                                        akref.ptr(out)) == 0
        t = [q for q in range(3) if q != d]                      # (t1, t2) = the transverse directions
        o = out.reshape(a["nmb"], nv, ext[t[1]], ext[t[0]])
        # the same number from numpy: mean over the fine faces behind each coarse face
        f = flx[d]
        fidx = [None, None, None]
        cs = [a["cis"], a["cjs"], a["cks"]]
        for q in range(3):
            c = np.arange(box[2*q], box[2*q + 1] + 1)
            fidx[q] = 2*c - cs[q] if q < dims else np.zeros(1, dtype=int)
        acc = 0.0
        cnt = 0
        for dk in ((0, 1) if (dims > 2 and d != 2) else (0,)):
            for dj in ((0, 1) if (dims > 1 and d != 1) else (0,)):
                for di in ((0, 1) if d != 0 else (0,)):
                    acc = acc + f[:, :, (fidx[2] + dk)[:, None, None], (fidx[1] + dj)[None, :, None],
                                  (fidx[0] + di)[None, None, :]]
                    cnt += 1
        want = acc/cnt                                           # (nmb, nv, nk, nj, ni)
        want = np.squeeze(want, axis=4 - d) if want.shape[4 - d] == 1 else want
        assert np.allclose(o, want.reshape(o.shape), rtol=1e-15, atol=1e-15), d
    for comp, box in _edge_boxes(a):
        ext = [box[1] - box[0] + 1, box[3] - box[2] + 1, box[5] - box[4] + 1]
        e = emf[comp].copy()
        # constant along the edge direction
        e[...] = np.take(e, [0], axis=3 - comp) if comp < dims else e
        out = np.zeros((a["nmb"], ext[0]*ext[1]*ext[2]))
        assert R.akref_restrict_emf(C.byref(a["pk"]), comp, akref.ptr(box), akref.ptr(e), akref.ptr(out)) == 0
        o = out.reshape(a["nmb"], ext[2], ext[1], ext[0])
        cs = [a["cis"], a["cjs"], a["cks"]]
        fi = [2*np.arange(box[2*q], box[2*q + 1] + 1) - cs[q] if q < dims else np.zeros(1, dtype=int)
              for q in range(3)]
        want = e[:, fi[2][:, None, None], fi[1][None, :, None], fi[0][None, None, :]]
        assert np.array_equal(o, want), comp


@pytest.mark.gpu
@pytest.mark.parametrize("ng", [2, 3])
@pytest.mark.parametrize("dims", [1, 2, 3])
def test_hip_flux_restriction_matches_the_oracle(dims, ng):
    import torch
    from athenak_amd import capi
    L, R = capi.lib(), akref.lib()
    a = _setup(dims, ng=ng, seed=31 + dims)
    flx, emf = _flux_arrays(a, np.random.default_rng(7))
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dxd = t(a["dx"])
    pkd = capi.Pack.from_buffer_copy(bytes(a["pk"]))
    pkd.dx = dxd.data_ptr()
    P, nv = C.byref(pkd), a["nvar"]
    for d, box in _face_boxes(a):
        n = nv*int(np.prod([box[2*q + 1] - box[2*q] + 1 for q in range(3)]))
        out, outd = np.zeros((a["nmb"], n)), torch.zeros((a["nmb"], n), dtype=torch.float64, device="cuda")
        R.akref_restrict_flux_cc(C.byref(a["pk"]), nv, d, akref.ptr(box), akref.ptr(flx[d]), akref.ptr(out))
        fd = t(flx[d])
        capi.check(L.akmi_restrict_flux_cc(P, nv, d, box.ctypes.data_as(C.c_void_p), capi._p(fd),
                                           capi._p(outd), None), "restrict_flux_cc")
        assert np.array_equal(out, outd.cpu().numpy()), d
    for comp, box in _edge_boxes(a):
        n = int(np.prod([box[2*q + 1] - box[2*q] + 1 for q in range(3)]))
        out, outd = np.zeros((a["nmb"], n)), torch.zeros((a["nmb"], n), dtype=torch.float64, device="cuda")
        R.akref_restrict_emf(C.byref(a["pk"]), comp, akref.ptr(box), akref.ptr(emf[comp]), akref.ptr(out))
        ed = t(emf[comp])
        capi.check(L.akmi_restrict_emf(P, comp, box.ctypes.data_as(C.c_void_p), capi._p(ed),
                                       capi._p(outd), None), "restrict_emf")
        assert np.array_equal(out, outd.cpu().numpy()), comp
    # boxes outside the coarse index space are refused, not read
    bad = np.array([0, 0, 0, 0, 0, 0], dtype=np.int32)
    outd = torch.zeros(8, dtype=torch.float64, device="cuda")
    fd = t(flx[0])
    assert L.akmi_restrict_flux_cc(P, nv, 0, bad.ctypes.data_as(C.c_void_p), capi._p(fd),
                                   capi._p(outd), None) != 0


# ---- primitive -> conserved of prolongated ghost cells --------------------------------------------
def _p2c_numpy(w, bcc, ideal):
    u = np.zeros_like(w)
    d, vx, vy, vz = w[:, 0], w[:, 1], w[:, 2], w[:, 3]
    u[:, 0], u[:, 1], u[:, 2], u[:, 3] = d, d*vx, d*vy, d*vz
    nfl = 5 if ideal else 4
    if ideal:
        if bcc is None:
            u[:, 4] = w[:, 4] + 0.5*d*(vx*vx + vy*vy + vz*vz)
        else:
            u[:, 4] = w[:, 4] + 0.5*(d*(vx*vx + vy*vy + vz*vz) +
                                     (bcc[:, 0]*bcc[:, 0] + bcc[:, 1]*bcc[:, 1] + bcc[:, 2]*bcc[:, 2]))
    for n in range(nfl, w.shape[1]):
        u[:, n] = d*w[:, n]
    return u


def _p2c_case(dims, mhd, ideal, nscal, seed):
    nx1, nx2, nx3 = DIMS[dims]
    ng, nmb = 2, 2
    nv = (5 if ideal else 4) + nscal
    pk, dx = akref.make_pack(nmb, nx1, nx2, nx3, ng, np.ones((nmb, 3)), 1.4, nvar=nv)
    pk.is_ideal = int(ideal)
    N1, N2, N3 = nx1 + 2*ng, (nx2 + 2*ng if nx2 > 1 else 1), (nx3 + 2*ng if nx3 > 1 else 1)
    rng = np.random.default_rng(seed)
    w = rng.normal(size=(nmb, nv, N3, N2, N1))
    w[:, 0] = np.abs(w[:, 0]) + 0.1
    bcc = rng.normal(size=(nmb, 3, N3, N2, N1)) if mhd else None
    box = np.array([0, ng - 1, 0, N2 - 1, 0, N3 - 1], dtype=np.int32)        # the inner-x1 ghost cells
    return pk, dx, w, bcc, box


@pytest.mark.parametrize("mhd,ideal,nscal", [(False, True, 0), (True, True, 0), (False, False, 2), (True, True, 1)])
@pytest.mark.parametrize("dims", [1, 3])
def test_prim2cons_is_the_reference_formula(dims, mhd, ideal, nscal):
    R = akref.lib()
    pk, dx, w, bcc, box = _p2c_case(dims, mhd, ideal, nscal, 41)
    u = np.full_like(w, 7.0)
    assert R.akref_prim2cons(C.byref(pk), akref.ptr(box), akref.ptr(w), akref.ptr(bcc) if mhd else None,
                             akref.ptr(u)) == 0
    want = _p2c_numpy(w, bcc, ideal)
    sl = (slice(None), slice(None), slice(box[4], box[5] + 1), slice(box[2], box[3] + 1), slice(box[0], box[1] + 1))
    assert np.array_equal(u[sl], want[sl])
    mask = np.ones_like(u, dtype=bool)
    mask[sl] = False
    assert np.all(u[mask] == 7.0), "cells outside the box are not touched"


@pytest.mark.gpu
@pytest.mark.parametrize("mhd,ideal,nscal", [(False, True, 0), (True, True, 0), (False, False, 2), (True, True, 1)])
@pytest.mark.parametrize("dims", [1, 2, 3])
def test_hip_prim2cons_matches_the_oracle(dims, mhd, ideal, nscal):
    import torch
    from athenak_amd import capi
    L, R = capi.lib(), akref.lib()
    pk, dx, w, bcc, box = _p2c_case(dims, mhd, ideal, nscal, 43)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dxd = t(dx)
    pkd = capi.Pack.from_buffer_copy(bytes(pk))
    pkd.dx = dxd.data_ptr()
    u = np.full_like(w, 7.0)
    ud = t(u)
    R.akref_prim2cons(C.byref(pk), akref.ptr(box), akref.ptr(w), akref.ptr(bcc) if mhd else None, akref.ptr(u))
    wd, bd = t(w), (t(bcc) if mhd else None)         # named: the device arrays must outlive the launch
    capi.check(L.akmi_prim2cons(C.byref(pkd), box.ctypes.data_as(C.c_void_p), capi._p(wd),
                                capi._p(bd) if mhd else None, capi._p(ud), None), "prim2cons")
    assert np.array_equal(u, ud.cpu().numpy())
    bad = np.array([0, 10**6, 0, 0, 0, 0], dtype=np.int32)
    assert L.akmi_prim2cons(C.byref(pkd), bad.ctypes.data_as(C.c_void_p), capi._p(wd), None, capi._p(ud),
                            None) != 0
