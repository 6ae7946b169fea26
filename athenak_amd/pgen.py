"""ProblemGenerator: initial conditions of the four decks on the hot path + L1 error norms.

Mirrors src/pgen/pgen.cpp:680-978 (CallProblemGenerator, OutputErrors),
src/pgen/tests/linear_wave.cpp:244-1010,1430-1437, tests/shock_tube.cpp:40-330,
tests/orszag_tang.cpp:42-123 and fluids/blast.cpp:134-392.  ICs are evaluated on the host in
fp64 (numpy) with the reference's formulas and uploaded; they are not part of the timed path.
"""
import math

import numpy as np
import torch

from . import capi
from .mesh import CellCenterX, FLT_MAX, LeftEdgeX

IDN, IVX, IVY, IVZ, IEN = 0, 1, 2, 3, 4


def HydroEigensystemPrim(d, v1, p, gamma):
    """linear_wave.cpp:793-867 (ideal gas): eigenvalues[5], right eigenvectors as columns"""
    a = math.sqrt(gamma*p/d)
    ev = [v1 - a, v1, v1, v1, v1 + a]
    rem = np.zeros((5, 5))
    rem[0][0], rem[1][0], rem[4][0] = 1.0, -a/d, a*a
    rem[0][1] = 1.0
    rem[2][2] = 1.0
    rem[3][3] = 1.0
    rem[0][4], rem[1][4], rem[4][4] = 1.0, a/d, a*a
    return ev, rem


def MHDEigensystemPrim(d, v1, p, b1, b2, b3, x, y, gamma):
    """linear_wave.cpp:876-1010 (ideal gas)"""
    btsq = b2*b2 + b3*b3
    bt = math.sqrt(btsq)
    asq = gamma*p/d
    if bt == 0.0:
        bet2, bet3 = 1.0, 0.0
    else:
        bet2, bet3 = b2/bt, b3/bt
    gm1 = gamma - 1.0
    bt_starsq = (gm1 - (gm1 - 1.0)*y)*btsq
    vaxsq = b1*b1/d
    ct2 = bt_starsq/d
    tsum = vaxsq + ct2 + asq
    tdif = vaxsq + ct2 - asq
    cf2_cs2 = math.sqrt(tdif*tdif + 4.0*asq*ct2)
    cfsq = 0.5*(tsum + cf2_cs2)
    cf = math.sqrt(cfsq)
    cssq = asq*vaxsq/cfsq
    cs = math.sqrt(cssq)
    if (cfsq - cssq) == 0.0:
        alpha_f, alpha_s = 1.0, 0.0
    elif (asq - cssq) <= 0.0:
        alpha_f, alpha_s = 0.0, 1.0
    elif (cfsq - asq) <= 0.0:
        alpha_f, alpha_s = 1.0, 0.0
    else:
        alpha_f = math.sqrt((asq - cssq)/(cfsq - cssq))
        alpha_s = math.sqrt((cfsq - asq)/(cfsq - cssq))
    sqrtd = math.sqrt(d)
    s = -1.0 if b1 < 0.0 else 1.0
    a = math.sqrt(asq)
    qf, qs = cf*alpha_f*s, cs*alpha_s*s
    af, as_ = a*alpha_f*sqrtd, a*alpha_s*sqrtd
    vax = math.sqrt(vaxsq)
    ev = [v1 - cf, v1 - vax, v1 - cs, v1, v1 + cs, v1 + vax, v1 + cf]
    r = np.zeros((7, 7))
    r[0] = [d*alpha_f, 0.0, d*alpha_s, 1.0, d*alpha_s, 0.0, d*alpha_f]
    r[1] = [-cf*alpha_f, 0.0, -cs*alpha_s, 0.0, cs*alpha_s, 0.0, cf*alpha_f]
    r[2] = [qs*bet2, -bet3, -qf*bet2, 0.0, qf*bet2, bet3, -qs*bet2]
    r[3] = [qs*bet3, bet2, -qf*bet3, 0.0, qf*bet3, -bet2, -qs*bet3]
    r[4] = [d*asq*alpha_f, 0.0, d*asq*alpha_s, 0.0, d*asq*alpha_s, 0.0, d*asq*alpha_f]
    r[5] = [as_*bet2, -bet3*s*sqrtd, -af*bet2, 0.0, -af*bet2, -bet3*s*sqrtd, as_*bet2]
    r[6] = [as_*bet3, bet2*s*sqrtd, -af*bet3, 0.0, -af*bet3, bet2*s*sqrtd, as_*bet3]
    return ev, r


def HydroEigensystemPrimIso(d, v1, iso_cs):
    """linear_wave.cpp:838-866 (isothermal): 4 waves, rows d,vx,vy,vz"""
    ev = [v1 - iso_cs, v1, v1, v1 + iso_cs]
    rem = np.zeros((4, 4))
    rem[0][0], rem[1][0] = 1.0, -iso_cs/d
    rem[2][1] = 1.0
    rem[3][2] = 1.0
    rem[0][3], rem[1][3] = 1.0, iso_cs/d
    return ev, rem


def MHDEigensystemPrimIso(d, v1, b1, b2, b3, y, iso_cs):
    """linear_wave.cpp:1005-1098 (isothermal): 6 waves, rows d,vx,vy,vz,by,bz"""
    btsq = b2*b2 + b3*b3
    bt = math.sqrt(btsq)
    if bt == 0.0:
        bet2, bet3 = 1.0, 0.0
    else:
        bet2, bet3 = b2/bt, b3/bt
    bt_starsq = btsq*y
    vaxsq = b1*b1/d
    iso_cs2 = iso_cs*iso_cs
    ct2 = bt_starsq/d
    tsum = vaxsq + ct2 + iso_cs2
    tdif = vaxsq + ct2 - iso_cs2
    cf2_cs2 = math.sqrt(tdif*tdif + 4.0*iso_cs2*ct2)
    cfsq = 0.5*(tsum + cf2_cs2)
    cf = math.sqrt(cfsq)
    cssq = iso_cs2*vaxsq/cfsq
    cs = math.sqrt(cssq)
    if (cfsq - cssq) == 0.0:
        alpha_f, alpha_s = 1.0, 0.0
    elif (iso_cs2 - cssq) <= 0.0:
        alpha_f, alpha_s = 0.0, 1.0
    elif (cfsq - iso_cs2) <= 0.0:
        alpha_f, alpha_s = 1.0, 0.0
    else:
        alpha_f = math.sqrt((iso_cs2 - cssq)/(cfsq - cssq))
        alpha_s = math.sqrt((cfsq - iso_cs2)/(cfsq - cssq))
    sqrtd = math.sqrt(d)
    s = -1.0 if b1 < 0.0 else 1.0
    qf, qs = cf*alpha_f*s, cs*alpha_s*s
    af, as_ = iso_cs*alpha_f*sqrtd, iso_cs*alpha_s*sqrtd
    vax = math.sqrt(vaxsq)
    ev = [v1 - cf, v1 - vax, v1 - cs, v1 + cs, v1 + vax, v1 + cf]
    r = np.zeros((6, 6))
    r[0] = [d*alpha_f, 0.0, d*alpha_s, d*alpha_s, 0.0, d*alpha_f]
    r[1] = [-cf*alpha_f, 0.0, -cs*alpha_s, cs*alpha_s, 0.0, cf*alpha_f]
    r[2] = [qs*bet2, -bet3, -qf*bet2, qf*bet2, bet3, -qs*bet2]
    r[3] = [qs*bet3, bet2, -qf*bet3, qf*bet3, -bet2, -qs*bet3]
    r[4] = [as_*bet2, -bet3*s*sqrtd, -af*bet2, -af*bet2, -bet3*s*sqrtd, as_*bet2]
    r[5] = [as_*bet3, bet2*s*sqrtd, -af*bet3, -af*bet3, bet2*s*sqrtd, as_*bet3]
    return ev, r


class ProblemGenerator:
    def __init__(self, pin, pmesh, restart=False):
        """restart=True: the restart constructor (src/pgen/pgen.cpp:97-330) -- the dependent
        variables were read from the rst file, the problem function only re-registers its
        final-work function"""
        self.pmy_mesh_ = pmesh
        self.pin = pin
        self.pgen_final_func = None
        self.set_initial_conditions = True
        name = pin.GetOrAddString("problem", "pgen_name", "none")
        self.pgen_name = name
        table = {"linear_wave": self.LinearWave, "shock_tube": self.ShockTube,
                 "orszag_tang": self.OrszagTang, "blast": self.UserProblem, "diffusion": self.Diffusion,
                 "cpaw": self.AlfvenWave}
        # user-defined boundary conditions (pgen.cpp:57-62): enrolled by the problem function
        self.user_bcs = any(b == capi.BC["user"] for b in pmesh.mesh_bcs)
        self.user_bcs_func = None
        if name not in table:
            raise RuntimeError("### FATAL ERROR problem/pgen_name = '%s' is not one of the "
                               "decks on this build's path %s" % (name, sorted(table)))
        table[name](pin, restart)

    # ---- helpers -------------------------------------------------------------------
    def _phys(self):
        pk = self.pmy_mesh_.pmb_pack
        return pk.phydro if pk.phydro is not None else pk.pmhd

    def _coords(self, m):
        """cell centres and left edges (incl. the +1 face) of block m over active cells"""
        pm = self.pmy_mesh_
        ind = pm.mb_indcs
        sz = pm.pmb_pack.pmb.mb_size[m]
        i = np.arange(ind.nx1 + 1)
        j = np.arange(ind.nx2 + 1)
        k = np.arange(ind.nx3 + 1)
        x1v = CellCenterX(i[:-1], ind.nx1, sz.x1min, sz.x1max)
        x2v = CellCenterX(j[:-1], ind.nx2, sz.x2min, sz.x2max)
        x3v = CellCenterX(k[:-1], ind.nx3, sz.x3min, sz.x3max)
        x1f = LeftEdgeX(i, ind.nx1, sz.x1min, sz.x1max)
        x2f = LeftEdgeX(j, ind.nx2, sz.x2min, sz.x2max)
        x3f = LeftEdgeX(k, ind.nx3, sz.x3min, sz.x3max)
        return x1v, x2v, x3v, x1f, x2f, x3f, sz

    def _finer_edge_masks(self, m):
        """Boolean masks over the edge-centred potentials a1, a2, a3 of block m ([nx3+1, nx2+1, nx1+1]
        each): True where the edge lies on a face or edge shared with a FINER neighbour.  There the
        reference evaluates the potential as the mean of the two fine-edge values, which makes the
        flux through shared fine/coarse faces identical (linear_wave.cpp:569-667, cpaw.cpp:227-325).
        None on a single-level mesh."""
        pm = self.pmy_mesh_
        if not pm.multilevel:
            return None
        ind = pm.mb_indcs
        nb = pm.pmb_pack.pmb.nghbr[m]
        mylev = int(pm.pmb_pack.pmb.mb_lev[m])

        def finer(*slots):
            return any(n in nb and nb[n].lev > mylev for n in slots)
        n3, n2, n1 = ind.nx3 + 1, ind.nx2 + 1, ind.nx1 + 1
        K, J, I = np.meshgrid(np.arange(n3), np.arange(n2), np.arange(n1), indexing="ij")
        ilo, ihi, jlo, jhi, klo, khi = I == 0, I == ind.nx1, J == 0, J == ind.nx2, K == 0, K == ind.nx3
        f = np.zeros((n3, n2, n1), dtype=bool)
        x1 = (ilo & finer(0, 1, 2, 3)) | (ihi & finer(4, 5, 6, 7))
        x2 = f.copy() if ind.nx2 == 1 else (jlo & finer(8, 9, 10, 11)) | (jhi & finer(12, 13, 14, 15))
        x3 = f.copy() if ind.nx3 == 1 else (klo & finer(24, 25, 26, 27)) | (khi & finer(28, 29, 30, 31))
        e12 = f.copy() if ind.nx2 == 1 else ((ilo & jlo & finer(16, 17)) | (ihi & jlo & finer(18, 19)) |
                                             (ilo & jhi & finer(20, 21)) | (ihi & jhi & finer(22, 23)))
        e31 = f.copy() if ind.nx3 == 1 else ((ilo & klo & finer(32, 33)) | (ihi & klo & finer(34, 35)) |
                                             (ilo & khi & finer(36, 37)) | (ihi & khi & finer(38, 39)))
        e23 = f.copy() if ind.nx3 == 1 else ((jlo & klo & finer(40, 41)) | (jhi & klo & finer(42, 43)) |
                                             (jlo & khi & finer(44, 45)) | (jhi & khi & finer(46, 47)))
        return (x2 | x3 | e23), (x1 | x3 | e31), (x1 | x2 | e12)

    def _active(self):
        ind = self.pmy_mesh_.mb_indcs
        return (slice(ind.ks, ind.ke + 1), slice(ind.js, ind.je + 1), slice(ind.is_, ind.ie + 1))

    def _upload_cc(self, dst, host):
        dst.copy_(torch.from_numpy(host).to(dst.device))

    def _prim_to_cons(self, w, bcc=None):
        """SingleP2C_IdealHyd / SingleP2C_IdealMHD (ideal_c2p_hyd.hpp:76-83, _mhd.hpp:75-84)"""
        u = np.zeros_like(w)
        d, vx, vy, vz = w[:, 0], w[:, 1], w[:, 2], w[:, 3]
        u[:, 0] = d
        u[:, 1] = d*vx
        u[:, 2] = d*vy
        u[:, 3] = d*vz
        ideal = self._phys().peos.eos_data.is_ideal
        for n in range(5 if ideal else 4, w.shape[1]):       # scalars: d*s
            u[:, n] = d*w[:, n]
        if not ideal:                    # SingleP2C_Isothermal*: no energy
            return u
        e = w[:, 4]
        if bcc is None:
            u[:, 4] = e + 0.5*d*(vx*vx + vy*vy + vz*vz)
        else:
            bx, by, bz = bcc[:, 0], bcc[:, 1], bcc[:, 2]
            u[:, 4] = e + 0.5*(d*(vx*vx + vy*vy + vz*vz) + (bx*bx + by*by + bz*bz))
        return u

    def _store(self, w, bfaces, to_u1=False):
        """write host prims (+faces) of all blocks into the physics object's registers"""
        phys = self._phys()
        is_mhd = self.pmy_mesh_.pmb_pack.pmhd is not None
        ks, js, is_ = self._active()
        bcc = None
        if is_mhd:
            b1, b2, b3 = bfaces
            bcc = np.zeros((w.shape[0], 3) + w.shape[2:])
            bcc[:, 0][:, ks, js, is_] = 0.5*(b1[:, ks, js, is_] + b1[:, ks, js, is_.start + 1:is_.stop + 1])
            bcc[:, 1][:, ks, js, is_] = 0.5*(b2[:, ks, js, is_] + b2[:, ks, js.start + 1:js.stop + 1, is_])
            bcc[:, 2][:, ks, js, is_] = 0.5*(b3[:, ks, js, is_] + b3[:, ks.start + 1:ks.stop + 1, js, is_])
        u = np.zeros_like(w)
        a = (slice(None), slice(None), ks, js, is_)
        u[a] = self._prim_to_cons(w[a], None if bcc is None else bcc[a])
        self._upload_cc(phys.u1 if to_u1 else phys.u0, u)
        if not to_u1:
            self._upload_cc(phys.w0, w)
        if is_mhd:
            dst = phys.b1 if to_u1 else phys.b0
            self._upload_cc(dst.x1f, b1)
            self._upload_cc(dst.x2f, b2)
            self._upload_cc(dst.x3f, b3)
            if not to_u1:
                self._upload_cc(phys.bcc0, bcc)

    def _alloc_host(self):
        pm = self.pmy_mesh_
        n3, n2, n1 = pm.mb_indcs.ncells
        nmb = pm.pmb_pack.nmb_thispack
        w = np.zeros((nmb, self._phys().nvars, n3, n2, n1))      # scalars (if any) start at zero
        b = (np.zeros((nmb, n3, n2, n1 + 1)), np.zeros((nmb, n3, n2 + 1, n1)),
             np.zeros((nmb, n3 + 1, n2, n1)))
        return w, b

    # ---- linear wave ---------------------------------------------------------------
    def LinearWave(self, pin, restart):
        pm = self.pmy_mesh_
        self.pgen_final_func = self.LinearWaveErrors
        if restart:                                      # linear_wave.cpp:247
            return
        along_x1 = pin.GetOrAddBoolean("problem", "along_x1", False)
        along_x2 = pin.GetOrAddBoolean("problem", "along_x2", False)
        along_x3 = pin.GetOrAddBoolean("problem", "along_x3", False)
        if (along_x1 and (along_x2 or along_x3)) or (along_x2 and along_x3):
            raise RuntimeError("### FATAL ERROR Can only specify one of along_x1/2/3 to be true")
        if (along_x2 or along_x3) and pm.one_d:
            raise RuntimeError("### FATAL ERROR Cannot specify waves along x2 or x3 axis in 1D")
        if along_x3 and pm.two_d:
            raise RuntimeError("### FATAL ERROR Cannot specify waves along x3 axis in 2D")
        ms = pm.mesh_size
        x1size, x2size, x3size = ms.x1max - ms.x1min, ms.x2max - ms.x2min, ms.x3max - ms.x3min
        cos_a3, sin_a3, cos_a2, sin_a2 = 1.0, 0.0, 1.0, 0.0
        if pm.multi_d and not along_x1:
            ang_3 = math.atan(x1size/x2size)
            sin_a3, cos_a3 = math.sin(ang_3), math.cos(ang_3)
        if pm.three_d and not along_x1:
            ang_2 = math.atan(0.5*(x1size*cos_a3 + x2size*sin_a3)/x3size)
            sin_a2, cos_a2 = math.sin(ang_2), math.cos(ang_2)
        if along_x2:
            cos_a3, sin_a3, cos_a2, sin_a2 = 0.0, 1.0, 1.0, 0.0
        if along_x3:
            cos_a3, sin_a3, cos_a2, sin_a2 = 0.0, 1.0, 0.0, 1.0
        lx = FLT_MAX
        if cos_a2*cos_a3 > 0.0:
            lx = min(lx, x1size*cos_a2*cos_a3)
        if cos_a2*sin_a3 > 0.0:
            lx = min(lx, x2size*cos_a2*sin_a3)
        if sin_a2 > 0.0:
            lx = min(lx, x3size*sin_a2)
        k_par = 2.0*math.pi/lx
        wave_flag = pin.GetInteger("problem", "wave_flag")
        amp = pin.GetReal("problem", "amp")
        d0, p0 = pin.GetReal("problem", "dens"), pin.GetReal("problem", "pgas")
        vx_0 = pin.GetOrAddReal("problem", "vx0", 0.0)
        vy_0 = pin.GetOrAddReal("problem", "vy0", 0.0)
        vz_0 = pin.GetOrAddReal("problem", "vz0", 0.0)
        bx_0 = pin.GetOrAddReal("problem", "bx0", 0.0)
        by_0 = pin.GetOrAddReal("problem", "by0", 0.0)
        bz_0 = pin.GetOrAddReal("problem", "bz0", 0.0)
        phys = self._phys()
        is_mhd = pm.pmb_pack.pmhd is not None
        eos = phys.peos.eos_data
        gamma = eos.gamma
        gm1 = gamma - 1.0
        dby = dbz = 0.0
        if not is_mhd and eos.is_ideal:
            ev, rem = HydroEigensystemPrim(d0, vx_0, p0, gamma)
        elif not is_mhd:
            ev, rem = HydroEigensystemPrimIso(d0, vx_0, eos.iso_cs)
        elif eos.is_ideal:
            ev, rem = MHDEigensystemPrim(d0, vx_0, p0, bx_0, by_0, bz_0, 0.0, 1.0, gamma)
            dby, dbz = amp*rem[5][wave_flag], amp*rem[6][wave_flag]
        else:
            ev, rem = MHDEigensystemPrimIso(d0, vx_0, bx_0, by_0, bz_0, 1.0, eos.iso_cs)
            dby, dbz = amp*rem[4][wave_flag], amp*rem[5][wave_flag]       # rem[nmhd_][wave]
        if self.set_initial_conditions:
            tlim = pin.GetReal("time", "tlim")
            pin.SetReal("time", "tlim", tlim*abs(lx/ev[wave_flag]))
        r = [rem[q][wave_flag] for q in range(4)] + [rem[4][wave_flag] if eos.is_ideal else 0.0]

        def A1(x1, x2, x3):
            x = x1*cos_a2*cos_a3 + x2*cos_a2*sin_a3 + x3*sin_a2
            y = -x1*sin_a3 + x2*cos_a3
            Ay = bz_0*x - (dbz/k_par)*np.cos(k_par*(x))
            Az = -by_0*x + (dby/k_par)*np.cos(k_par*(x)) + bx_0*y
            return -Ay*sin_a3 - Az*sin_a2*cos_a3

        def A2(x1, x2, x3):
            x = x1*cos_a2*cos_a3 + x2*cos_a2*sin_a3 + x3*sin_a2
            y = -x1*sin_a3 + x2*cos_a3
            Ay = bz_0*x - (dbz/k_par)*np.cos(k_par*(x))
            Az = -by_0*x + (dby/k_par)*np.cos(k_par*(x)) + bx_0*y
            return Ay*cos_a3 - Az*sin_a2*sin_a3

        def A3(x1, x2, x3):
            x = x1*cos_a2*cos_a3 + x2*cos_a2*sin_a3 + x3*sin_a2
            y = -x1*sin_a3 + x2*cos_a3
            Az = -by_0*x + (dby/k_par)*np.cos(k_par*(x)) + bx_0*y
            return Az*cos_a2

        w, bf = self._alloc_host()
        ks, js, is_ = self._active()
        for m in range(w.shape[0]):
            x1v, x2v, x3v, x1f, x2f, x3f, sz = self._coords(m)
            X3, X2, X1 = np.meshgrid(x3v, x2v, x1v, indexing="ij")
            x = cos_a2*(X1*cos_a3 + X2*sin_a3) + X3*sin_a2
            sn = np.sin(k_par*x)
            rho = d0 + amp*sn*r[0]
            vx = vx_0 + amp*sn*r[1]
            vy = vy_0 + amp*sn*r[2]
            vz = vz_0 + amp*sn*r[3]
            w[m, IDN][ks, js, is_] = rho
            w[m, IVX][ks, js, is_] = vx*cos_a2*cos_a3 - vy*sin_a3 - vz*sin_a2*cos_a3
            w[m, IVY][ks, js, is_] = vx*cos_a2*sin_a3 + vy*cos_a3 - vz*sin_a2*sin_a3
            w[m, IVZ][ks, js, is_] = vx*sin_a2 + vz*cos_a2
            if eos.is_ideal:
                w[m, IEN][ks, js, is_] = (p0 + amp*sn*r[4])/gm1
            if is_mhd:
                # vector potential at [ks:ke+1, js:je+1, is:ie+1] (linear_wave.cpp:545-567)
                F3, F2, F1 = np.meshgrid(x3f, x2f, x1f, indexing="ij")
                nx3, nx2, nx1 = len(x3v), len(x2v), len(x1v)
                # cell-centre coordinate at index n (one past the last cell) continues the formula
                ind = pm.mb_indcs
                x1vx = CellCenterX(np.arange(nx1 + 1), ind.nx1, sz.x1min, sz.x1max)
                x2vx = CellCenterX(np.arange(nx2 + 1), ind.nx2, sz.x2min, sz.x2max)
                x3vx = CellCenterX(np.arange(nx3 + 1), ind.nx3, sz.x3min, sz.x3max)
                V3, V2, V1 = np.meshgrid(x3vx, x2vx, x1vx, indexing="ij")
                a1 = A1(V1, F2, F3)
                a2 = A2(F1, V2, F3)
                a3 = A3(F1, F2, V3)
                dx1, dx2, dx3 = sz.dx1, sz.dx2, sz.dx3
                masks = self._finer_edge_masks(m)
                if masks is not None:                    # linear_wave.cpp:569-667
                    a1 = np.where(masks[0], 0.5*(A1(V1 + 0.25*dx1, F2, F3) + A1(V1 - 0.25*dx1, F2, F3)), a1)
                    a2 = np.where(masks[1], 0.5*(A2(F1, V2 + 0.25*dx2, F3) + A2(F1, V2 - 0.25*dx2, F3)), a2)
                    a3 = np.where(masks[2], 0.5*(A3(F1, F2, V3 + 0.25*dx3) + A3(F1, F2, V3 - 0.25*dx3)), a3)
                b1 = (a3[:-1, 1:, :] - a3[:-1, :-1, :])/dx2 - (a2[1:, :-1, :] - a2[:-1, :-1, :])/dx3
                b2 = (a1[1:, :, :-1] - a1[:-1, :, :-1])/dx3 - (a3[:-1, :, 1:] - a3[:-1, :, :-1])/dx1
                b3 = (a2[:, :-1, 1:] - a2[:, :-1, :-1])/dx1 - (a1[:, 1:, :-1] - a1[:, :-1, :-1])/dx2
                bf[0][m][ks, js, is_.start:is_.stop + 1] = b1
                bf[1][m][ks, js.start:js.stop + 1, is_] = b2
                bf[2][m][ks.start:ks.stop + 1, js, is_] = b3
        self._store(w, bf, to_u1=not self.set_initial_conditions)

    # ---- circularly polarized Alfven wave (the reference's static-refinement regression) ----------
    def AlfvenWave(self, pin, restart):
        """ProblemGenerator::AlfvenWave, src/pgen/tests/cpaw.cpp:79-435 (MHD; conserved variables and
        face fields from a vector potential, with the two-point potential on edges shared with finer
        blocks, :227-325)"""
        pm = self.pmy_mesh_
        self.pgen_final_func = self.AlfvenWaveErrors
        if restart:
            return
        if pm.pmb_pack.pmhd is None:
            return
        b_par, b_perp = pin.GetReal("problem", "b_par"), pin.GetReal("problem", "b_perp")
        v_par, pres = pin.GetReal("problem", "v_par"), pin.GetReal("problem", "pres")
        den = 1.0
        v_perp = b_perp/math.sqrt(den)
        along_x1 = pin.GetOrAddBoolean("problem", "along_x1", False)
        along_x2 = pin.GetOrAddBoolean("problem", "along_x2", False)
        along_x3 = pin.GetOrAddBoolean("problem", "along_x3", False)
        if (along_x1 and (along_x2 or along_x3)) or (along_x2 and along_x3):
            raise RuntimeError("### FATAL ERROR Can only specify one of along_x1/2/3 to be true")
        if (along_x2 or along_x3) and pm.one_d:
            raise RuntimeError("### FATAL ERROR Cannot specify waves along x2 or x3 axis in 1D")
        if along_x3 and pm.two_d:
            raise RuntimeError("### FATAL ERROR Cannot specify waves along x3 axis in 2D")
        ms = pm.mesh_size
        x1size, x2size, x3size = ms.x1max - ms.x1min, ms.x2max - ms.x2min, ms.x3max - ms.x3min
        cos_a3, sin_a3, cos_a2, sin_a2 = 1.0, 0.0, 1.0, 0.0
        if pm.multi_d and not along_x1:
            ang_3 = math.atan(x1size/x2size)
            sin_a3, cos_a3 = math.sin(ang_3), math.cos(ang_3)
        if pm.three_d and not along_x1:
            ang_2 = math.atan(0.5*(x1size*cos_a3 + x2size*sin_a3)/x3size)
            sin_a2, cos_a2 = math.sin(ang_2), math.cos(ang_2)
        if along_x2:
            cos_a3, sin_a3, cos_a2, sin_a2 = 0.0, 1.0, 1.0, 0.0
        if along_x3:
            cos_a3, sin_a3, cos_a2, sin_a2 = 0.0, 1.0, 0.0, 1.0
        lam = FLT_MAX
        if cos_a2*cos_a3 > 0.0:
            lam = min(lam, x1size*cos_a2*cos_a3)
        if cos_a2*sin_a3 > 0.0:
            lam = min(lam, x2size*cos_a2*sin_a3)
        if sin_a2 > 0.0:
            lam = min(lam, x3size*sin_a2)
        k_par = 2.0*math.pi/lam
        pol = 1.0 if pin.GetOrAddBoolean("problem", "right_polar", True) else -1.0
        phys = pm.pmb_pack.pmhd
        eos = phys.peos.eos_data
        gm1 = eos.gamma - 1.0
        if self.set_initial_conditions:
            tlim = pin.GetReal("time", "tlim")
            pin.SetReal("time", "tlim", tlim*abs(lam/(b_par/math.sqrt(den))))

        def _xy(x1, x2, x3):
            return (x1*cos_a2*cos_a3 + x2*cos_a2*sin_a3 + x3*sin_a2, -x1*sin_a3 + x2*cos_a3)

        def A1(x1, x2, x3):
            x, y = _xy(x1, x2, x3)
            ay = pol*(b_perp/k_par)*np.sin(k_par*(x))
            az = (b_perp/k_par)*np.cos(k_par*(x)) + b_par*y
            return -ay*sin_a3 - az*sin_a2*cos_a3

        def A2(x1, x2, x3):
            x, y = _xy(x1, x2, x3)
            ay = pol*(b_perp/k_par)*np.sin(k_par*(x))
            az = (b_perp/k_par)*np.cos(k_par*(x)) + b_par*y
            return ay*cos_a3 - az*sin_a2*sin_a3

        def A3(x1, x2, x3):
            x, y = _xy(x1, x2, x3)
            az = (b_perp/k_par)*np.cos(k_par*(x)) + b_par*y
            return az*cos_a2

        w, bf = self._alloc_host()
        u = np.zeros_like(w)
        ks, js, is_ = self._active()
        ind = pm.mb_indcs
        for m in range(w.shape[0]):
            x1v, x2v, x3v, x1f, x2f, x3f, sz = self._coords(m)
            F3, F2, F1 = np.meshgrid(x3f, x2f, x1f, indexing="ij")
            x1vx = CellCenterX(np.arange(ind.nx1 + 1), ind.nx1, sz.x1min, sz.x1max)
            x2vx = CellCenterX(np.arange(ind.nx2 + 1), ind.nx2, sz.x2min, sz.x2max)
            x3vx = CellCenterX(np.arange(ind.nx3 + 1), ind.nx3, sz.x3min, sz.x3max)
            V3, V2, V1 = np.meshgrid(x3vx, x2vx, x1vx, indexing="ij")
            a1, a2, a3 = A1(V1, F2, F3), A2(F1, V2, F3), A3(F1, F2, V3)
            dx1, dx2, dx3 = sz.dx1, sz.dx2, sz.dx3
            masks = self._finer_edge_masks(m)
            if masks is not None:
                a1 = np.where(masks[0], 0.5*(A1(V1 + 0.25*dx1, F2, F3) + A1(V1 - 0.25*dx1, F2, F3)), a1)
                a2 = np.where(masks[1], 0.5*(A2(F1, V2 + 0.25*dx2, F3) + A2(F1, V2 - 0.25*dx2, F3)), a2)
                a3 = np.where(masks[2], 0.5*(A3(F1, F2, V3 + 0.25*dx3) + A3(F1, F2, V3 - 0.25*dx3)), a3)
            X3, X2, X1 = np.meshgrid(x3v, x2v, x1v, indexing="ij")
            x = cos_a2*(X1*cos_a3 + X2*sin_a3) + X3*sin_a2
            sn = np.sin(k_par*x)
            cs = pol*np.cos(k_par*x)
            mx = den*v_par
            my = -pol*den*v_perp*sn
            mz = -pol*den*v_perp*cs
            u[m, IDN][ks, js, is_] = den
            u[m, 1][ks, js, is_] = mx*cos_a2*cos_a3 - my*sin_a3 - mz*sin_a2*cos_a3
            u[m, 2][ks, js, is_] = mx*cos_a2*sin_a3 + my*cos_a3 - mz*sin_a2*sin_a3
            u[m, 3][ks, js, is_] = mx*sin_a2 + mz*cos_a2
            b1 = (a3[:-1, 1:, :] - a3[:-1, :-1, :])/dx2 - (a2[1:, :-1, :] - a2[:-1, :-1, :])/dx3
            b2 = (a1[1:, :, :-1] - a1[:-1, :, :-1])/dx3 - (a3[:-1, :, 1:] - a3[:-1, :, :-1])/dx1
            b3 = (a2[:, :-1, 1:] - a2[:, :-1, :-1])/dx1 - (a1[:, 1:, :-1] - a1[:, :-1, :-1])/dx2
            bf[0][m][ks, js, is_.start:is_.stop + 1] = b1
            bf[1][m][ks, js.start:js.stop + 1, is_] = b2
            bf[2][m][ks.start:ks.stop + 1, js, is_] = b3
            if eos.is_ideal:
                sq = lambda q: q*q
                u[m, IEN][ks, js, is_] = pres/gm1 + \
                    0.5*(sq(0.5*(b1[:, :, :-1] + b1[:, :, 1:])) + sq(0.5*(b2[:, :-1, :] + b2[:, 1:, :])) +
                         sq(0.5*(b3[:-1, :, :] + b3[1:, :, :]))) + \
                    (0.5/den)*(sq(u[m, 1][ks, js, is_]) + sq(u[m, 2][ks, js, is_]) + sq(u[m, 3][ks, js, is_]))
        to_u1 = not self.set_initial_conditions
        self._upload_cc(phys.u1 if to_u1 else phys.u0, u)
        dst = phys.b1 if to_u1 else phys.b0
        for q, name in enumerate(("x1f", "x2f", "x3f")):
            self._upload_cc(getattr(dst, name), bf[q])

    def AlfvenWaveErrors(self):
        """AlfvenWaveErrors, cpaw.cpp:441-598: the error file of LinearWaveErrors"""
        self.set_initial_conditions = False
        self.AlfvenWave(self.pin, False)
        self.set_initial_conditions = True
        return self.OutputErrors()

    def LinearWaveErrors(self):
        """linear_wave.cpp:1430-1437 + pgen.cpp:680-900: returns [RMS-L1, L-infty, L1...]"""
        self.set_initial_conditions = False
        self.LinearWave(self.pin, False)
        self.set_initial_conditions = True
        return self.OutputErrors()

    def OutputErrors(self):
        pm = self.pmy_mesh_
        phys = self._phys()
        is_mhd = pm.pmb_pack.pmhd is not None
        ks, js, is_ = self._active()
        a = (slice(None), slice(None), ks, js, is_)
        u0 = phys.u0.cpu().numpy()[a]
        u1 = phys.u1.cpu().numpy()[a]
        dx = pm.pmb_pack.pmb.dx
        vol = (dx[:, 0]*dx[:, 1]*dx[:, 2])[:, None, None, None, None]
        ev = vol*np.abs(u0 - u1)
        l1 = list(ev.sum(axis=(0, 2, 3, 4)))
        linf = float(ev.max())
        if is_mhd:
            def bcc(f):
                b1, b2, b3 = f.x1f.cpu().numpy(), f.x2f.cpu().numpy(), f.x3f.cpu().numpy()
                return (0.5*(b1[:, ks, js, is_] + b1[:, ks, js, is_.start + 1:is_.stop + 1]),
                        0.5*(b2[:, ks, js, is_] + b2[:, ks, js.start + 1:js.stop + 1, is_]),
                        0.5*(b3[:, ks, js, is_] + b3[:, ks.start + 1:ks.stop + 1, js, is_]))
            c0, c1 = bcc(phys.b0), bcc(phys.b1)
            v3 = vol[:, 0]
            ideal = phys.peos.eos_data.is_ideal
            for q in range(3):
                e = v3*np.abs(c0[q] - c1[q])
                l1.append(e.sum())
                # pgen.cpp:793-805 reads the maxima from slots IEN+1..IEN+3 whatever bindx is:
                # with the isothermal EOS the B1 error (slot 4) does not enter L-infty
                if ideal or q > 0:
                    linf = max(linf, float(e.max()))
        l1 = np.array(l1, dtype=np.float64)
        linf_arr = np.array([linf])
        if pm.nranks > 1:
            import torch.distributed as dist
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.from_numpy(l1).to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            l1 = t.cpu().numpy()
            t2 = torch.from_numpy(linf_arr).to(dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            linf_arr = t2.cpu().numpy()
        ms = pm.mesh_size
        volm = (ms.x1max - ms.x1min)*(ms.x2max - ms.x2min)*(ms.x3max - ms.x3min)
        l1 = l1/volm
        rms = math.sqrt(float((l1*l1).sum()))
        errs = np.concatenate([[rms, linf_arr[0]/volm], l1])
        if self.write_errors_file and pm.my_rank == 0:
            self.WriteErrorsFile(errs)
        return errs

    write_errors_file = False       # set by Driver.Finalize: <basename>-errs.dat (pgen.cpp:850-898)

    def WriteErrorsFile(self, errs):
        """pgen.cpp:850-898: header on first use, then one appended line per run; column 4 is
        the RMS-L1 error the reference's regression tests read (testutils.py:273-276)"""
        import os
        pm = self.pmy_mesh_
        is_mhd = pm.pmb_pack.pmhd is not None
        fname = self.pin.GetString("job", "basename") + "-errs.dat"
        new = not os.path.exists(fname)
        with open(fname, "a") as f:
            if new:
                f.write("# Nx1  Nx2  Nx3   Ncycle   RMS-L1       L-infty       ")
                f.write("d_L1          M1_L1         M2_L1         M3_L1         ")
                if self._phys().peos.eos_data.is_ideal:
                    f.write("E_L1          ")
                if is_mhd:
                    f.write("B1_L1         B2_L1         B3_L1")
                f.write("\n")
            mi = pm.mesh_indcs
            f.write("%04d  %04d  %04d  %05d  %e %e" % (mi.nx1, mi.nx2, mi.nx3, pm.ncycle, errs[0], errs[1]))
            for v in errs[2:]:
                f.write("  %e" % v)
            f.write("\n")

    # ---- shock tube ----------------------------------------------------------------
    def ShockTube(self, pin, restart):
        if restart:                                      # shock_tube.cpp:41
            return
        pm = self.pmy_mesh_
        shk_dir = pin.GetInteger("problem", "shock_dir")
        if shk_dir < 1 or shk_dir > 3:
            raise RuntimeError("### FATAL ERROR shock_dir=%d must be either 1,2, or 3" % shk_dir)
        ivx = shk_dir
        ivy = IVX + ((ivx - IVX) + 1) % 3
        ivz = IVX + ((ivx - IVX) + 2) % 3
        xshock = pin.GetReal("problem", "xshock")
        ms = pm.mesh_size
        lo = (ms.x1min, ms.x2min, ms.x3min)[shk_dir - 1]
        hi = (ms.x1max, ms.x2max, ms.x3max)[shk_dir - 1]
        if xshock < lo or xshock > hi:
            raise RuntimeError("### FATAL ERROR xshock=%g lies outside x%d domain" % (xshock, shk_dir))
        phys = self._phys()
        is_mhd = pm.pmb_pack.pmhd is not None
        gm1 = phys.peos.eos_data.gamma - 1.0
        g = pin.GetReal
        wl = [g("problem", "dl"), g("problem", "ul"), g("problem", "vl"), g("problem", "wl"),
              g("problem", "pl")/gm1]
        wr = [g("problem", "dr"), g("problem", "ur"), g("problem", "vr"), g("problem", "wr"),
              g("problem", "pr")/gm1]
        if is_mhd:
            bL = [g("problem", "bxl"), g("problem", "byl"), g("problem", "bzl")]
            bR = [g("problem", "bxr"), g("problem", "byr"), g("problem", "bzr")]
            # rotate (bx,by,bz) into the sweep frame, shock_tube.cpp:236-256
            rot = {1: (0, 1, 2), 2: (2, 0, 1), 3: (1, 2, 0)}[shk_dir]
            bL = [bL[q] for q in rot]
            bR = [bR[q] for q in rot]
        w, bf = self._alloc_host()
        ks, js, is_ = self._active()
        for m in range(w.shape[0]):
            x1v, x2v, x3v, x1f, x2f, x3f, sz = self._coords(m)
            X3, X2, X1 = np.meshgrid(x3v, x2v, x1v, indexing="ij")
            x = (X1, X2, X3)[shk_dir - 1]
            left = x < xshock
            sel = lambda a, b: np.where(left, a, b)
            w[m, IDN][ks, js, is_] = sel(wl[0], wr[0])
            w[m, ivx][ks, js, is_] = sel(wl[1]*1.0, wr[1]*1.0)
            w[m, ivy][ks, js, is_] = sel(wl[2]*1.0, wr[2]*1.0)
            w[m, ivz][ks, js, is_] = sel(wl[3]*1.0, wr[3]*1.0)
            if self._phys().peos.eos_data.is_ideal:
                w[m, IEN][ks, js, is_] = sel(wl[4], wr[4])
            if is_mhd:
                v1, v2, v3 = sel(bL[0], bR[0]), sel(bL[1], bR[1]), sel(bL[2], bR[2])
                bf[0][m][ks, js, is_] = v1
                bf[0][m][ks, js, is_.stop] = v1[:, :, -1]
                bf[1][m][ks, js, is_] = v2
                bf[1][m][ks, js.stop, is_] = v2[:, -1, :]
                bf[2][m][ks, js, is_] = v3
                bf[2][m][ks.stop, js, is_] = v3[-1, :, :]
        self._store(w, bf)

    # ---- Gaussian-pulse diffusion tests (src/pgen/tests/diffusion.cpp) --------------------------
    def _diff_gaussian(self, coef, time, x1, x2, x3):
        """DiffusionGaussian, diffusion.cpp:63-76"""
        dv = self.diffvars
        ndim = float(dv["spread_x1"]) + float(dv["spread_x2"]) + float(dv["spread_x3"])
        spread = 1.0 + 4.0*coef*time
        r2 = 0.0
        if dv["spread_x1"]:
            r2 = r2 + (x1 - dv["x10"])**2
        if dv["spread_x2"]:
            r2 = r2 + (x2 - dv["x20"])**2
        if dv["spread_x3"]:
            r2 = r2 + (x3 - dv["x30"])**2
        return (dv["amp"]/spread**(0.5*ndim))*np.exp(-r2/spread)

    def _diff_cons_state(self, coef, gamma, time, x1, x2, x3):
        """DiffusionConsState, diffusion.cpp:85-116: (5, ...) conserved state of the analytic solution"""
        dv = self.diffvars
        g = self._diff_gaussian(coef, time, x1, x2, x3)
        gm1 = gamma - 1.0
        rho = np.ones_like(g)
        m = [np.zeros_like(g), np.zeros_like(g), np.zeros_like(g)]
        p0 = np.full_like(g, 1.0/gamma)
        if dv["conduction_test"]:
            p0 = g
        if dv["viscosity_test"]:
            m[dv["vel_comp"] - 1] = rho*g
        return np.stack([rho, m[0], m[1], m[2],
                         p0/gm1 + 0.5*(m[0]**2 + m[1]**2 + m[2]**2)/rho])

    def _diff_coef(self):
        ph = self.pmy_mesh_.pmb_pack.phydro
        gamma = ph.peos.eos_data.gamma
        coef = 0.0
        if self.diffvars["conduction_test"] and ph.pcond is not None:
            coef = (gamma - 1.0)*ph.pcond.alpha_iso
        if self.diffvars["viscosity_test"] and ph.pvisc is not None:
            coef = ph.pvisc.nu_iso
        return coef, gamma

    def Diffusion(self, pin, restart):
        pm = self.pmy_mesh_
        if pin.GetString("time", "evolution") != "kinematic":
            raise RuntimeError("### FATAL ERROR Diffusion tests must be run in kinematic mode")
        self.pgen_final_func = self.DiffusionErrors
        self.user_bcs_func = self.GaussianProfileBCs
        if restart:
            return
        g, gb = pin.GetOrAddReal, pin.GetOrAddBoolean
        self.diffvars = dv = dict(
            amp=g("problem", "amp", 1.0e-6), x10=g("problem", "x10", 0.0), x20=g("problem", "x20", 0.0),
            x30=g("problem", "x30", 0.0), conduction_test=pin.GetBoolean("problem", "conduction_test"),
            viscosity_test=pin.GetBoolean("problem", "viscosity_test"),
            resistivity_test=pin.GetBoolean("problem", "resistivity_test"),
            spread_x1=gb("problem", "spread_x1", True), spread_x2=gb("problem", "spread_x2", False),
            spread_x3=gb("problem", "spread_x3", False), vel_comp=pin.GetOrAddInteger("problem", "vel_comp", 2))
        ntests = int(dv["conduction_test"]) + int(dv["viscosity_test"]) + int(dv["resistivity_test"])
        if ntests != 1:
            raise RuntimeError("### FATAL ERROR Exactly one of conduction_test/viscosity_test/"
                               "resistivity_test must be set true (got %d)" % ntests)
        if pm.pmb_pack.pmhd is not None:
            self._diffusion_mhd()
            return
        ph = pm.pmb_pack.phydro
        if dv["conduction_test"] and ph.pcond is None:
            raise RuntimeError("### FATAL ERROR Conduction not defined in Hydro input block")
        if dv["viscosity_test"] and ph.pvisc is None:
            raise RuntimeError("### FATAL ERROR Viscosity not defined in Hydro input block")
        if not ph.peos.eos_data.is_ideal:
            raise RuntimeError("### FATAL ERROR Diffusion test requires ideal EOS in Hydro block")
        coef, gamma = self._diff_coef()
        n3, n2, n1 = pm.mb_indcs.ncells
        u = np.zeros((pm.pmb_pack.nmb_thispack, 5, n3, n2, n1))
        ks, js, is_ = self._active()
        for m in range(u.shape[0]):
            x1v, x2v, x3v, _, _, _, _ = self._coords(m)
            X3, X2, X1 = np.meshgrid(x3v, x2v, x1v, indexing="ij")
            u[m][:, ks, js, is_] = self._diff_cons_state(coef, gamma, pm.time, X1, X2, X3)
        # solution in u1 when computing errors, in u0 as initial condition (diffusion.cpp:201)
        self._upload_cc(ph.u0 if self.set_initial_conditions else ph.u1, u)

    def _diffusion_mhd(self):
        """diffusion.cpp:237-311: Gaussian pulse in one field component (uniform along its own
        axis, so the staggered faces carry the cell-centred value and div B = 0), rho = 1, v = 0,
        p = 1/gamma; diffuses with eta_ohm"""
        pm = self.pmy_mesh_
        ph = pm.pmb_pack.pmhd
        dv = self.diffvars
        if not dv["resistivity_test"]:
            raise RuntimeError("### FATAL ERROR MHD diffusion test only supports the resistivity test")
        if ph.presist is None:
            raise RuntimeError("### FATAL ERROR Resistivity (mhd/eta_ohm) not defined in MHD input block")
        if not ph.peos.eos_data.is_ideal:
            raise RuntimeError("### FATAL ERROR Diffusion test requires ideal EOS in MHD block")
        gamma = ph.peos.eos_data.gamma
        coef = ph.presist.eta_ohm
        bcomp = dv["vel_comp"]
        w, bf = self._alloc_host()
        ks, js, is_ = self._active()
        for m in range(w.shape[0]):
            x1v, x2v, x3v, _, _, _, _ = self._coords(m)
            X3, X2, X1 = np.meshgrid(x3v, x2v, x1v, indexing="ij")
            g = self._diff_gaussian(coef, pm.time, X1, X2, X3)
            w[m, IDN][ks, js, is_] = 1.0
            w[m, IEN][ks, js, is_] = (1.0/gamma)/(gamma - 1.0)
            if bcomp == 1:
                bf[0][m][ks, js, is_] = g
                bf[0][m][ks, js, is_.stop] = g[:, :, -1]
            elif bcomp == 2:
                bf[1][m][ks, js, is_] = g
                bf[1][m][ks, js.stop, is_] = g[:, -1, :]
            else:
                bf[2][m][ks, js, is_] = g
                bf[2][m][ks.stop, js, is_] = g[-1, :, :]
        self._store(w, bf, to_u1=not self.set_initial_conditions)

    def DiffusionErrors(self):
        """diffusion.cpp:330-337"""
        self.set_initial_conditions = False
        self.Diffusion(self.pin, False)
        self.set_initial_conditions = True
        return self.OutputErrors()

    def GaussianProfileBCs(self):
        """diffusion.cpp:344-461: ghost zones of `user` boundaries hold the analytic solution at the
        current time (all transverse cells incl. ghosts, x1 then x2 then x3)"""
        pm = self.pmy_mesh_
        ph = pm.pmb_pack.phydro
        if ph is None:
            return
        coef, gamma = self._diff_coef()
        ind = pm.mb_indcs
        ng = ind.ng
        n3, n2, n1 = ind.ncells
        user = capi.BC["user"]
        bcs = pm.pmb_pack.pmb.mb_bcs
        for m in range(pm.pmb_pack.nmb_thispack):
            sz = pm.pmb_pack.pmb.mb_size[m]
            xv = [CellCenterX(np.arange(n) - s, nx, lo, hi) for n, s, nx, lo, hi in (
                (n1, ind.is_, ind.nx1, sz.x1min, sz.x1max), (n2, ind.js, ind.nx2, sz.x2min, sz.x2max),
                (n3, ind.ks, ind.nx3, sz.x3min, sz.x3max))]
            slabs = [(0, slice(0, ng), 2), (1, slice(ind.ie + 1, ind.ie + 1 + ng), 2)]
            if pm.multi_d:
                slabs += [(2, slice(0, ng), 1), (3, slice(ind.je + 1, ind.je + 1 + ng), 1)]
            if pm.three_d:
                slabs += [(4, slice(0, ng), 0), (5, slice(ind.ke + 1, ind.ke + 1 + ng), 0)]
            for face, sl, axis in slabs:
                if int(bcs[m][face]) != user:
                    continue
                idx = [slice(None), slice(None), slice(None)]
                idx[axis] = sl
                X3, X2, X1 = np.meshgrid(xv[2][idx[0]], xv[1][idx[1]], xv[0][idx[2]], indexing="ij")
                cons = self._diff_cons_state(coef, gamma, pm.time, X1, X2, X3)
                ph.u0[m][(slice(None),) + tuple(idx)] = torch.from_numpy(cons).to(ph.u0.device)

    # ---- Orszag-Tang ---------------------------------------------------------------
    def OrszagTang(self, pin, restart):
        if restart:                                      # orszag_tang.cpp:43
            return
        pm = self.pmy_mesh_
        if pm.pmb_pack.pmhd is None:
            raise RuntimeError("### FATAL ERROR Orszag-Tang test can only be run in MHD, but no "
                               "<mhd> block in input file")
        phys = pm.pmb_pack.pmhd
        B0 = 1.0/math.sqrt(4.0*math.pi)
        d0 = 25.0/(36.0*math.pi)
        v0 = 1.0
        p0 = 5.0/(12.0*math.pi)
        gm1 = phys.peos.eos_data.gamma - 1.0

        def A3(x1, x2):
            return (B0/(4.0*math.pi))*(np.cos(4.0*math.pi*x1) - 2.0*np.cos(2.0*math.pi*x2))

        n3, n2, n1 = pm.mb_indcs.ncells
        nmb = pm.pmb_pack.nmb_thispack
        u = np.zeros((nmb, phys.nvars, n3, n2, n1))
        _, bf = self._alloc_host()
        ks, js, is_ = self._active()
        for m in range(nmb):
            x1v, x2v, x3v, x1f, x2f, x3f, sz = self._coords(m)
            nk = len(x3v)
            X2, X1 = np.meshgrid(x2v, x1v, indexing="ij")
            ones = np.ones((nk, 1, 1))
            u[m, IDN][ks, js, is_] = d0
            u[m, IVX][ks, js, is_] = ones*(d0*v0*np.sin(2.0*math.pi*X2))
            u[m, IVY][ks, js, is_] = ones*(-d0*v0*np.sin(2.0*math.pi*X1))
            u[m, IVZ][ks, js, is_] = 0.0
            # faces from curl(A3): b1 on (j, i face), b2 on (j face, i)
            F2, F1 = np.meshgrid(x2f, x1f, indexing="ij")
            a3 = A3(F1, F2)                                  # [nx2+1, nx1+1]
            b1 = (a3[1:, :] - a3[:-1, :])/sz.dx2             # [nx2, nx1+1]
            b2 = -(a3[:, 1:] - a3[:, :-1])/sz.dx1            # [nx2+1, nx1]
            bf[0][m][ks, js, is_.start:is_.stop + 1] = ones*b1
            bf[1][m][ks, js.start:js.stop + 1, is_] = ones*b2
            bx = 0.5*(b1[:, :-1] + b1[:, 1:])
            by = 0.5*(b2[:-1, :] + b2[1:, :])
            bz = 0.5*(0.0 + 0.0)
            e = p0/gm1 + (0.5/u[m, IDN][ks, js, is_])*(
                u[m, IVX][ks, js, is_]**2 + u[m, IVY][ks, js, is_]**2 + u[m, IVZ][ks, js, is_]**2) \
                + 0.5*(ones*(bx*bx) + ones*(by*by) + bz*bz)
            u[m, IEN][ks, js, is_] = e
        self._upload_cc(phys.u0, u)
        self._upload_cc(phys.b0.x1f, bf[0])
        self._upload_cc(phys.b0.x2f, bf[1])
        self._upload_cc(phys.b0.x3f, bf[2])

    # ---- blast (user problem in the reference: -D PROBLEM=fluids/blast) --------------
    def UserProblem(self, pin, restart):
        if restart:
            return
        pm = self.pmy_mesh_
        phys = self._phys()
        is_mhd = pm.pmb_pack.pmhd is not None
        rout = pin.GetReal("problem", "outer_radius")
        rin = rout - pin.GetReal("problem", "inner_radius")
        if is_mhd:
            p_amb = pin.GetOrAddReal("problem", "pi_amb", 1.0)
            d_amb = pin.GetOrAddReal("problem", "di_amb", 1.0)
        else:
            p_amb = pin.GetOrAddReal("problem", "pn_amb", 1.0)
            d_amb = pin.GetOrAddReal("problem", "dn_amb", 1.0)
        prat = pin.GetReal("problem", "prat")
        drat = pin.GetOrAddReal("problem", "drat", 1.0)
        b_amb = pin.GetOrAddReal("problem", "b_amb", 0.1)
        gm1 = phys.peos.eos_data.gamma - 1.0
        w, bf = self._alloc_host()
        ks, js, is_ = self._active()
        for m in range(w.shape[0]):
            x1v, x2v, x3v, x1f, x2f, x3f, sz = self._coords(m)
            X3, X2, X1 = np.meshgrid(x3v, x2v, x1v, indexing="ij")
            rad = np.sqrt(X1*X1 + X2*X2 + X3*X3)
            den = np.full_like(rad, d_amb)
            pres = np.full_like(rad, p_amb)
            inner = rad < rin
            den[inner] = d_amb*drat
            pres[inner] = p_amb*prat
            ramp = (rad < rout) & ~inner
            if ramp.any():
                f = (rad[ramp] - rin)/(rout - rin)
                den[ramp] = np.exp((1.0 - f)*math.log(drat*d_amb) + f*math.log(d_amb))
                pres[ramp] = np.exp((1.0 - f)*math.log(prat*p_amb) + f*math.log(p_amb))
            w[m, IDN][ks, js, is_] = den
            w[m, IEN][ks, js, is_] = pres/gm1
            if is_mhd:
                a3 = b_amb*x2f                                    # blast.cpp:355
                b1 = ((a3[1:] - a3[:-1])/sz.dx2)[None, :, None]    # [1, nx2, 1]
                nk, nj, ni = len(x3v), len(x2v), len(x1v)
                bf[0][m][ks, js, is_.start:is_.stop + 1] = np.broadcast_to(b1, (nk, nj, ni + 1))
                bf[1][m][ks, js.start:js.stop + 1, is_] = -(0.0)/sz.dx1
        self._store(w, bf)
