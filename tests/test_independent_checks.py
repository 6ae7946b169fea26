"""Checks that do NOT share a source with the kernels: HLLC, PPM4, the corner EMF and the
conserved<->primitive conversion restated from the published papers (tests/independent.py) and
compared, to 1e-10, with the oracle's akref_* (CPU) AND with the product's akmi_* (GPU).  The
bitwise product==oracle tests cannot see a misreading both sides share; these can."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import independent as ind  # noqa: E402
from oracle import akref  # noqa: E402

TOL = 1e-10
G = 5.0/3.0
RS_HLLC, RS_ADVECT = 2, 5


class Backend:
    """calls  <prefix>_<name>(pack, *args)  with numpy arrays placed where the library wants them"""

    def __init__(self, hip):
        self.hip = hip
        if hip:
            from athenak_amd import capi
            self.capi = capi
            self.L = capi.lib()
        else:
            self.L = akref.lib()

    def __call__(self, name, pk, *args):
        if not self.hip:
            a = [akref.ptr(x) if isinstance(x, np.ndarray) else x for x in args]
            rc = getattr(self.L, "akref_" + name)(C.byref(pk), *a)
            assert rc == 0, name
            return
        import torch
        capi = self.capi
        dx = np.ctypeslib.as_array(C.cast(pk.dx, C.POINTER(C.c_double)), shape=(pk.nmb*3,)).copy()
        dxd = torch.from_numpy(dx).cuda()
        pkd = capi.Pack.from_buffer_copy(bytes(pk))
        pkd.dx = dxd.data_ptr()
        dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() if isinstance(x, np.ndarray) else x
               for x in args]
        a = [capi._p(x) if isinstance(x, torch.Tensor) else x for x in dev]
        capi.check(getattr(self.L, "akmi_" + name)(C.byref(pkd), *a, None), name)
        torch.cuda.synchronize()
        for h, d in zip(args, dev):
            if isinstance(h, np.ndarray):
                h[...] = d.cpu().numpy()


BACKENDS = [pytest.param(False, id="oracle"), pytest.param(True, id="hip", marks=pytest.mark.gpu)]


def mkpack(n1, n2, n3, ng, nvar, dx=(1.0, 1.0, 1.0)):
    pk = akref.Pack()
    pk.nmb, pk.nvar, pk.nx1, pk.nx2, pk.nx3, pk.ng = 1, nvar, n1, n2, n3, ng
    dxa = np.array([dx], dtype=np.float64)
    pk.dx = dxa.ctypes.data
    pk.gamma, pk.dfloor, pk.pfloor, pk.tfloor, pk.sfloor = G, 1e-30, 1e-30, 1e-30, 1e-30
    pk.sigma_max, pk.iso_cs, pk.is_ideal = 3.4028234663852886e38, 1.0, 1     # eos.cpp default: FLT_MAX
    pk._keep = dxa
    N1 = n1 + 2*ng
    N2 = n2 + 2*ng if n2 > 1 else 1
    N3 = n3 + 2*ng if n3 > 1 else 1
    return pk, (N3, N2, N1)


def random_prims(rng, shape, vscale=1.0):
    w = np.empty((1, 5) + shape)
    w[0, 0] = rng.uniform(0.2, 2.0, shape)
    w[0, 1:4] = vscale*rng.normal(size=(3,) + shape)
    w[0, 4] = rng.uniform(0.2, 3.0, shape)
    return w


# ---------------------------------------------------------------------------------------------
def test_hllc_single_state_vs_toro():
    """akref_hllc (hllc_hyd.hpp:20-115 restated) against Toro's star-state form"""
    L = akref.lib()
    rng = np.random.default_rng(3)
    worst, n = 0.0, 0
    for t in range(6000):
        wl = np.array([rng.uniform(.1, 2), rng.normal()*1.5, rng.normal(), rng.normal(), rng.uniform(.05, 3)])
        wr = np.array([rng.uniform(.1, 2), rng.normal()*1.5, rng.normal(), rng.normal(), rng.uniform(.05, 3)])
        if t % 13 == 0:
            wr = wl.copy()
        if t % 17 == 0:
            wl[1] += 4.0
            wr[1] += 4.0          # supersonic to the right: F = F_L
        if t % 19 == 0:
            wl[1] -= 4.0
            wr[1] -= 4.0
        f2, ok = ind.hllc_toro(G, wl, wr)
        if not ok:
            continue              # outside the solver's validity, see hllc_toro
        f = np.zeros(5)
        L.akref_hllc(C.c_double(G), akref.ptr(wl), akref.ptr(wr), akref.ptr(f))
        worst = max(worst, np.max(np.abs(f - f2))/np.max(np.abs(f2)))
        n += 1
    assert n > 4500 and worst < TOL, (n, worst)


@pytest.mark.parametrize("hip", BACKENDS)
def test_hllc_flux_kernel_vs_toro(hip):
    """*_hydro_fluxes with donor-cell states + HLLC on a 1-D row: flux(i) = HLLC(w[i-1], w[i])"""
    be = Backend(hip)
    n1, ng = 256, 2
    pk, (N3, N2, N1) = mkpack(n1, 1, 1, ng, 5)
    rng = np.random.default_rng(5)
    w = random_prims(rng, (N3, N2, N1), vscale=0.7)
    f = [np.zeros((1, 5, N3, N2, N1)) for _ in range(3)]
    be("hydro_fluxes", pk, akref.RECON["dc"], RS_HLLC, w, f[0], f[1], f[2], 0)
    worst, n = 0.0, 0
    for i in range(ng, ng + n1 + 1):
        f2, ok = ind.hllc_toro(G, w[0, :, 0, 0, i-1], w[0, :, 0, 0, i])
        if not ok:
            continue
        worst = max(worst, np.max(np.abs(f[0][0, :, 0, 0, i] - f2))/np.max(np.abs(f2)))
        n += 1
    assert n > 200 and worst < TOL, (n, worst)


# ---------------------------------------------------------------------------------------------
def test_ppm4_single_cell_vs_colella_woodward():
    L = akref.lib()
    rng = np.random.default_rng(7)
    worst = 0.0
    for t in range(300):
        kind = t % 4
        x = np.arange(40.0)
        if kind == 0:
            a = rng.normal(size=40)                        # rough: limiter everywhere
        elif kind == 1:
            a = np.sin(0.21*x + rng.uniform(0, 6)) + 2.0   # smooth: parabola mostly untouched
        elif kind == 2:
            a = np.where(x < 20, 1.0, 0.125) + 0.01*rng.normal(size=40)      # a step
        else:
            a = np.cumsum(rng.uniform(0.0, 1.0, 40))**2*0.01                # monotone, steepening
        aL, aR = ind.ppm_cw(a)
        for j in range(2, 38):
            ql, qr = C.c_double(), C.c_double()
            L.akref_ppm4(*[C.c_double(a[j + o]) for o in (-2, -1, 0, 1, 2)], C.byref(ql), C.byref(qr))
            s = max(1.0, np.max(np.abs(a[j-2:j+3])))
            worst = max(worst, abs(ql.value - aR[j])/s, abs(qr.value - aL[j])/s)
    assert worst < TOL, worst


@pytest.mark.parametrize("hip", BACKENDS)
def test_ppm4_flux_kernel_vs_colella_woodward(hip):
    """*_hydro_fluxes with PPM4 states and the ADVECT solver (advect_hyd.hpp:19-55: every flux
    is the upwind state times its own normal velocity, by the sign of the LEFT velocity), so the
    flux exposes the reconstructed L/R states directly; all three sweeps on a 3-D block"""
    be = Backend(hip)
    n, ng = 12, 3
    pk, (N3, N2, N1) = mkpack(n, n, n, ng, 5)
    rng = np.random.default_rng(11)
    w = random_prims(rng, (N3, N2, N1))
    # smooth it a little so that some cells keep an unlimited parabola
    for ax in (2, 3, 4):
        w = 0.5*w + 0.25*(np.roll(w, 1, ax) + np.roll(w, -1, ax))
    f = [np.zeros((1, 5, N3, N2, N1)) for _ in range(3)]
    be("hydro_fluxes", pk, akref.RECON["ppm4"], RS_ADVECT, w, f[0], f[1], f[2], 0)
    worst = 0.0
    lo, hi = ng, ng + n
    for d, ax in ((0, 2), (1, 1), (2, 0)):
        # velocity components rotate with the sweep: (normal, t1, t2)
        iv = [(1, 2, 3), (2, 3, 1), (3, 1, 2)][d]
        wd = np.moveaxis(w[0], 1 + ax, -1)            # sweep axis last
        fd = np.moveaxis(f[d][0], 1 + ax, -1)
        t_rng = range(lo, hi)
        for p in t_rng:
            for q in t_rng:
                rows = wd[:, p, q, :]
                LR = [ind.ppm_cw(rows[v]) for v in range(5)]
                for i in range(lo, hi + 1):
                    wl = np.array([LR[v][1][i-1] for v in range(5)])     # right edge of cell i-1
                    wr = np.array([LR[v][0][i] for v in range(5)])       # left edge of cell i
                    s = wl if wl[iv[0]] >= 0.0 else wr
                    vn = s[iv[0]]
                    ex = np.zeros(5)
                    ex[0] = s[0]*vn
                    ex[iv[0]] = s[0]*vn*vn
                    ex[iv[1]] = s[iv[1]]*vn
                    ex[iv[2]] = s[iv[2]]*vn
                    ex[4] = s[4]*vn
                    got = fd[:, p, q, i]
                    worst = max(worst, np.max(np.abs(got - ex))/max(1.0, np.max(np.abs(ex))))
    assert worst < TOL, worst


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hip", BACKENDS)
def test_corner_emf_vs_gardiner_stone(hip):
    be = Backend(hip)
    n, ng = 10, 2
    dx = (0.37, 0.61, 1.13)
    pk, (N3, N2, N1) = mkpack(n, n, n, ng, 5, dx)
    rng = np.random.default_rng(13)
    sh = (N3, N2, N1)
    w = random_prims(rng, sh)
    bcc = rng.normal(size=(1, 3) + sh)
    face = {k: rng.normal(size=(1,) + sh) for k in ("e3x1", "e2x1", "e1x2", "e3x2", "e2x3", "e1x3")}
    flx = [rng.normal(size=(1, 5, N3, N2, N1 + 1)), rng.normal(size=(1, 5, N3, N2 + 1, N1)),
           rng.normal(size=(1, 5, N3 + 1, N2, N1))]
    e1 = np.zeros((1, N3 + 1, N2 + 1, N1))
    e2 = np.zeros((1, N3 + 1, N2, N1 + 1))
    e3 = np.zeros((1, N3, N2 + 1, N1 + 1))
    be("mhd_corner_e", pk, w, bcc, face["e3x1"], face["e2x1"], face["e1x2"], face["e3x2"], face["e2x3"],
       face["e1x3"], flx[0], flx[1], flx[2], e1, e2, e3)
    v = w[0, 1:4]
    Ecc = -np.cross(v, bcc[0], axis=0)                           # E = -v x B
    M = [flx[0][0, 0, :, :, :N1], flx[1][0, 0, :, :N2, :], flx[2][0, 0, :N3, :, :]]
    AX = {1: 2, 2: 1, 3: 0}                                       # x1 is the last array axis
    s = slice(ng, ng + n + 1)
    worst = 0.0
    for c, out in ((1, e1), (2, e2), (3, e3)):
        a, b = c % 3 + 1, (c + 1) % 3 + 1
        ex = ind.corner_emf_gs(Ecc[c-1], face["e%dx%d" % (c, a)][0], face["e%dx%d" % (c, b)][0],
                               M[a-1], M[b-1], AX[a], AX[b], dx[a-1], dx[b-1])
        got = out[0][:N3, :N2, :N1]
        worst = max(worst, np.max(np.abs(got[s, s, s] - ex[s, s, s])))
        assert np.max(np.abs(got[s, s, s])) > 0.5
    assert worst < TOL, worst


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hip", BACKENDS)
def test_cons_to_prim_inverts_textbook_prim_to_cons(hip):
    be = Backend(hip)
    n, ng = 8, 2
    pk, (N3, N2, N1) = mkpack(n, n, n, ng, 5)
    sh = (N3, N2, N1)
    rng = np.random.default_rng(17)
    w = random_prims(rng, sh)
    cnt = np.zeros(3, dtype=np.int32)
    # hydro
    u = ind.cons_from_prim(G, w[0])[None].copy()
    u_in = u.copy()
    w2 = np.zeros_like(w)
    be("hydro_c2p", pk, u, w2, 0, N1-1, 0, N2-1, 0, N3-1, cnt)
    assert np.max(np.abs(w2 - w)/np.maximum(1.0, np.abs(w))) < TOL and not cnt.any()
    assert np.array_equal(u, u_in)                       # no floor touched the conserved state
    # and the library's own prim->cons agrees with the textbook one
    u3 = np.zeros_like(u)
    box = np.array([0, N1-1, 0, N2-1, 0, N3-1], dtype=np.int32)
    be("prim2cons", pk, box, w, None, u3)
    assert np.max(np.abs(u3 - u)/np.maximum(1.0, np.abs(u))) < TOL
    # MHD: face fields -> cell-centred averages, total energy includes B^2/2
    b1 = rng.normal(size=(1, N3, N2, N1 + 1))
    b2 = rng.normal(size=(1, N3, N2 + 1, N1))
    b3 = rng.normal(size=(1, N3 + 1, N2, N1))
    bc = np.stack([0.5*(b1[0, :, :, :-1] + b1[0, :, :, 1:]), 0.5*(b2[0, :, :-1, :] + b2[0, :, 1:, :]),
                   0.5*(b3[0, :-1] + b3[0, 1:])])
    u = ind.cons_from_prim(G, w[0], bc)[None].copy()
    u_in = u.copy()
    w2 = np.zeros_like(w)
    bcc = np.zeros((1, 3) + sh)
    cnt[:] = 0
    be("mhd_c2p", pk, u, b1, b2, b3, w2, bcc, 0, N1-1, 0, N2-1, 0, N3-1, cnt)
    assert np.max(np.abs(bcc[0] - bc)) < 1e-15
    assert np.max(np.abs(w2 - w)/np.maximum(1.0, np.abs(w))) < TOL and not cnt.any()
    assert np.array_equal(u, u_in)
    u3 = np.zeros_like(u)
    be("prim2cons", pk, box, w, bcc, u3)
    assert np.max(np.abs(u3 - u)/np.maximum(1.0, np.abs(u))) < TOL


# ---------------------------------------------------------------------------------------------
# round 3: the pieces that had no check outside the shared source -- PLM (the reconstruction of the headline
# run), the signs of CT, the face-field prolongation and restriction of the refined-mesh path
@pytest.mark.parametrize("hip", BACKENDS)
def test_plm_flux_kernel_vs_van_leer(hip):
    """*_hydro_fluxes with PLM states and the ADVECT solver exposes the reconstructed L/R states (as for
    PPM4 above); src/reconstruct/plm.hpp:20-37 against van Leer's limiter in flux-limiter form"""
    be = Backend(hip)
    n, ng = 12, 2
    pk, (N3, N2, N1) = mkpack(n, n, n, ng, 5)
    rng = np.random.default_rng(23)
    w = random_prims(rng, (N3, N2, N1))
    w[0, 0, :, :, 5] = w[0, 0, :, :, 4]              # flat spots and extrema: limiter branches
    w[0, 4, 6] = w[0, 4, 5]
    f = [np.zeros((1, 5, N3, N2, N1)) for _ in range(3)]
    be("hydro_fluxes", pk, akref.RECON["plm"], RS_ADVECT, w, f[0], f[1], f[2], 0)
    worst, nlim = 0.0, 0
    lo, hi = ng, ng + n
    for d, ax in ((0, 2), (1, 1), (2, 0)):
        iv = [(1, 2, 3), (2, 3, 1), (3, 1, 2)][d]
        wd = np.moveaxis(w[0], 1 + ax, -1)
        fd = np.moveaxis(f[d][0], 1 + ax, -1)
        for p in range(lo, hi):
            for q in range(lo, hi):
                rows = wd[:, p, q, :]
                LR = [ind.plm_van_leer(rows[v]) for v in range(5)]
                nlim += sum(int(np.sum(LR[v][0][lo:hi] == rows[v][lo:hi])) for v in range(5))
                for i in range(lo, hi + 1):
                    wl = np.array([LR[v][1][i-1] for v in range(5)])
                    wr = np.array([LR[v][0][i] for v in range(5)])
                    s = wl if wl[iv[0]] >= 0.0 else wr
                    vn = s[iv[0]]
                    ex = np.zeros(5)
                    ex[0] = s[0]*vn
                    ex[iv[0]] = s[0]*vn*vn
                    ex[iv[1]] = s[iv[1]]*vn
                    ex[iv[2]] = s[iv[2]]*vn
                    ex[4] = s[4]*vn
                    worst = max(worst, np.max(np.abs(fd[:, p, q, i] - ex))/max(1.0, np.max(np.abs(ex))))
    assert worst < TOL, worst
    assert nlim > 100                               # the limiter did switch slopes off in places


@pytest.mark.parametrize("hip", BACKENDS)
def test_ct_is_the_discrete_stokes_theorem(hip):
    """MHD::CT (src/mhd/mhd_ct.cpp:45-77): the change of every face field equals minus dt times the circulation
    of a RANDOM edge field around that face divided by its area, with the right-hand orientation, for
    dx1 != dx2 != dx3 -- signs and spacings checked against Stokes' theorem, not against the source; and the
    divergence of the change vanishes"""
    be = Backend(hip)
    n, ng = 8, 2
    dx = (0.37, 0.61, 1.13)
    pk, (N3, N2, N1) = mkpack(n, n, n, ng, 5, dx)
    rng = np.random.default_rng(29)
    e1 = rng.normal(size=(1, N3 + 1, N2 + 1, N1))
    e2 = rng.normal(size=(1, N3 + 1, N2, N1 + 1))
    e3 = rng.normal(size=(1, N3, N2 + 1, N1 + 1))
    b0 = [rng.normal(size=(1, N3, N2, N1 + 1)), rng.normal(size=(1, N3, N2 + 1, N1)), rng.normal(size=(1, N3 + 1, N2, N1))]
    b1 = [x.copy() for x in b0]
    old = [x.copy() for x in b0]
    dt = 0.0137
    be("mhd_ct", pk, C.c_double(1.0), C.c_double(0.0), C.c_double(dt), e1, e2, e3, b0[0], b0[1], b0[2], b1[0], b1[1], b1[2])
    # a common (N3+1, N2+1, N1+1) layout: pad the short direction of each edge array
    full = (N3 + 1, N2 + 1, N1 + 1)
    E = []
    for c, e in enumerate((e1, e2, e3)):
        A = np.zeros(full)
        A[:e.shape[1], :e.shape[2], :e.shape[3]] = e[0]
        E.append(A)
    s = slice(ng, ng + n)
    worst = 0.0
    rate = []
    for axis in range(3):
        db = ind.faraday_circulation(E, axis, dx)
        got = (b0[axis][0] - old[axis][0])/dt
        sl = [s, s, s]
        sl[2 - axis] = slice(ng, ng + n + 1)          # faces: one more along their own direction
        sl = tuple(sl)
        worst = max(worst, np.max(np.abs(got[sl] - db[sl])))
        rate.append(got)
        assert np.max(np.abs(got[sl])) > 1.0
    assert worst < 1e-10*np.max(np.abs(E[0]))/min(dx), worst
    div = ((rate[0][s, s, ng + 1:ng + n + 1] - rate[0][s, s, s])/dx[0] + (rate[1][s, ng + 1:ng + n + 1, s] - rate[1][s, s, s])/dx[1]
           + (rate[2][ng + 1:ng + n + 1, s, s] - rate[2][s, s, s])/dx[2])
    assert np.max(np.abs(div)) < 1e-10*np.max(np.abs(rate[0]))/min(dx)


def _curl_of_random_potential(rng, shape3, d=(1.0, 1.0, 1.0)):
    """a solenoidal face field on a block of `shape3` = (n3, n2, n1) cells: B = curl A with A random on the edges"""
    n3, n2, n1 = shape3
    a1 = rng.normal(size=(n3 + 1, n2 + 1, n1))
    a2 = rng.normal(size=(n3 + 1, n2, n1 + 1))
    a3 = rng.normal(size=(n3, n2 + 1, n1 + 1))
    b1 = (a3[:, 1:, :] - a3[:, :-1, :])/d[1] - (a2[1:, :, :] - a2[:-1, :, :])/d[2]
    b2 = (a1[1:, :, :] - a1[:-1, :, :])/d[2] - (a3[:, :, 1:] - a3[:, :, :-1])/d[0]
    b3 = (a2[:, :, 1:] - a2[:, :, :-1])/d[0] - (a1[:, 1:, :] - a1[:, :-1, :])/d[1]
    return b1, b2, b3


@pytest.mark.parametrize("hip", BACKENDS)
def test_field_prolongation_keeps_a_solenoidal_field_solenoidal(hip):
    """ProlongFCShared* + ProlongFCInternal (src/mesh/prolongation.hpp:69-230, Toth & Roe 2002): a coarse face
    field that is the discrete curl of a random vector potential comes out divergence-free on every fine cell,
    the four fine faces on a coarse face carry its flux, and RestrictFC (src/mesh/mesh_refinement.cpp:1283-1382)
    -- by definition the area average of the fine faces -- returns the coarse field"""
    be = Backend(hip)
    n, ng = 8, 2
    pk, (N3, N2, N1) = mkpack(n, n, n, ng, 5)
    c = n//2 + 2*ng
    rng = np.random.default_rng(31)
    cb = [x[None].copy() for x in _curl_of_random_potential(rng, (c, c, c))]
    fb = [np.zeros((1, N3, N2, N1 + 1)), np.zeros((1, N3, N2 + 1, N1)), np.zeros((1, N3 + 1, N2, N1))]
    lo, hi = ng, ng + n//2 - 1
    for comp in range(3):
        box = np.array([lo, hi, lo, hi, lo, hi], dtype=np.int32)
        box[2*comp + 1] += 1
        be("prolong_fc_shared", pk, comp, box, cb[comp], fb[comp])
    box = np.array([lo, hi, lo, hi, lo, hi], dtype=np.int32)
    be("prolong_fc_internal", pk, box, fb[0], fb[1], fb[2])
    s = slice(ng, ng + n)
    s1 = slice(ng + 1, ng + n + 1)
    div = (fb[0][0][s, s, s1] - fb[0][0][s, s, s]) + (fb[1][0][s, s1, s] - fb[1][0][s, s, s]) + (fb[2][0][s1, s, s] - fb[2][0][s, s, s])
    scale = max(np.max(np.abs(x)) for x in cb)
    assert np.max(np.abs(div)) < 1e-10*scale, np.max(np.abs(div))
    assert max(np.max(np.abs(x[0][s, s, s])) for x in fb) > 0.5
    # flux through every coarse face = sum of the fluxes through its four fine faces (areas 1 : 1/4)
    cs = slice(lo, hi + 1)
    cs1 = slice(lo, hi + 2)
    f0 = fb[0][0][s, s, ng:ng + n + 1:2]
    avg0 = 0.25*(f0[0::2, 0::2] + f0[1::2, 0::2] + f0[0::2, 1::2] + f0[1::2, 1::2])
    assert np.max(np.abs(avg0 - cb[0][0][cs, cs, cs1])) < 1e-10*scale
    f1 = fb[1][0][s, ng:ng + n + 1:2, s]
    avg1 = 0.25*(f1[0::2, :, 0::2] + f1[1::2, :, 0::2] + f1[0::2, :, 1::2] + f1[1::2, :, 1::2])
    assert np.max(np.abs(avg1 - cb[1][0][cs, cs1, cs])) < 1e-10*scale
    f2 = fb[2][0][ng:ng + n + 1:2, s, s]
    avg2 = 0.25*(f2[:, 0::2, 0::2] + f2[:, 1::2, 0::2] + f2[:, 0::2, 1::2] + f2[:, 1::2, 1::2])
    assert np.max(np.abs(avg2 - cb[2][0][cs1, cs, cs])) < 1e-10*scale
    # RestrictFC of an arbitrary solenoidal FINE field: area averages, and the coarse field is solenoidal too
    fine = [x[None].copy() for x in _curl_of_random_potential(rng, (N3, N2, N1))]
    cr = [np.zeros((1, c, c, c + 1)), np.zeros((1, c, c + 1, c)), np.zeros((1, c + 1, c, c))]
    be("restrict_fc", pk, fine[0], fine[1], fine[2], cr[0], cr[1], cr[2])
    g0 = fine[0][0][s, s, ng:ng + n + 1:2]
    assert np.max(np.abs(0.25*(g0[0::2, 0::2] + g0[1::2, 0::2] + g0[0::2, 1::2] + g0[1::2, 1::2]) - cr[0][0][cs, cs, cs1])) < 1e-10*scale
    g1 = fine[1][0][s, ng:ng + n + 1:2, s]
    assert np.max(np.abs(0.25*(g1[0::2, :, 0::2] + g1[1::2, :, 0::2] + g1[0::2, :, 1::2] + g1[1::2, :, 1::2]) - cr[1][0][cs, cs1, cs])) < 1e-10*scale
    g2 = fine[2][0][ng:ng + n + 1:2, s, s]
    assert np.max(np.abs(0.25*(g2[:, 0::2, 0::2] + g2[:, 1::2, 0::2] + g2[:, 0::2, 1::2] + g2[:, 1::2, 1::2]) - cr[2][0][cs1, cs, cs])) < 1e-10*scale
    cdiv = ((cr[0][0][cs, cs, lo + 1:hi + 2] - cr[0][0][cs, cs, cs]) + (cr[1][0][cs, lo + 1:hi + 2, cs] - cr[1][0][cs, cs, cs])
            + (cr[2][0][lo + 1:hi + 2, cs, cs] - cr[2][0][cs, cs, cs]))
    assert np.max(np.abs(cdiv)) < 1e-10*scale
