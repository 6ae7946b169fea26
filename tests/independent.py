"""Independent restatements FROM THE PUBLISHED PAPERS of four pieces of the hot path, written in a
different algebraic form from both the product kernels and the oracle so that a mistake shared by
those two (they were written by the same hand from the same source) shows up:

  hllc_toro      Toro, "Riemann Solvers and Numerical Methods for Fluid Dynamics" (3rd ed.), ch. 10:
                 star states U*_K (10.39), F*_K = F_K + S_K (U*_K - U_K) (10.38), S* (10.37),
                 pressure-based wave-speed estimates (10.59)-(10.61) with the PVRS pressure (10.67)
  ppm_cw         Colella & Woodward, JCP 54, 174 (1984): interface value (1.6) with unlimited
                 average slopes on a uniform grid, monotonisation (1.10) in its published product
                 form; the bound of the interface value by its two neighbours is Colella & Sekora,
                 JCP 227, 7069 (2008) eq. 13
  corner_emf_gs  Gardiner & Stone, JCP 205, 509 (2005): eq. 41 with the derivative estimates of
                 eq. 45 upwinded by the sign of the mass flux (eq. 50), for any of the 3 components
  cons_from_prim textbook definitions of the conserved variables (ideal gas)

Used only by tests/test_independent_checks.py."""
import numpy as np


# ---------------------------------------------------------------------------------------------
def hllc_toro(g, wl, wr):
    """wl, wr = (rho, u, v, w, e_int): primitive L/R states with the internal energy DENSITY as the
    fifth entry (the reference's primitive set).  Returns F = (rho, mx, my, mz, E) and ok = False
    for inputs outside the solver's validity, which callers skip: a negative star pressure (the
    reference clips it at zero, Toro does not) or wave-speed estimates that are not ordered
    S_L < S* < S_R (very strong collisions, where the PVRS estimate can put S_L above S*: Toro's
    branch order then returns F_L, the reference's averaged form a star flux; ~0.3% of the random
    states drawn by the tests)."""
    def state(w):
        r, u, v, ww, ei = w
        p = (g - 1.0)*ei
        E = ei + 0.5*r*(u*u + v*v + ww*ww)
        U = np.array([r, r*u, r*v, r*ww, E])
        F = np.array([r*u, r*u*u + p, r*u*v, r*u*ww, u*(E + p)])
        return r, u, v, ww, p, E, U, F, np.sqrt(g*p/r)
    rl, ul, vl, wwl, pl, El, UL, FL, al = state(wl)
    rr, ur, vr, wwr, pr, Er, UR, FR, ar = state(wr)
    ppvrs = 0.5*(pl + pr) - 0.5*(ur - ul)*0.5*(rl + rr)*0.5*(al + ar)          # (10.67)

    def q(pk):                                                                   # (10.60)
        return 1.0 if ppvrs <= pk else np.sqrt(1.0 + (g + 1.0)/(2.0*g)*(ppvrs/pk - 1.0))
    SL = ul - al*q(pl)                                                           # (10.59)
    SR = ur + ar*q(pr)
    Ss = (pr - pl + rl*ul*(SL - ul) - rr*ur*(SR - ur))/(rl*(SL - ul) - rr*(SR - ur))   # (10.37)
    pstar = pl + rl*(SL - ul)*(Ss - ul)                                          # (10.36)

    def ustar(r, u, v, ww, p, E, S):                                             # (10.39)
        f = r*(S - u)/(S - Ss)
        return f*np.array([1.0, Ss, v, ww, E/r + (Ss - u)*(Ss + p/(r*(S - u)))])
    if SL >= 0.0:
        F = FL
    elif Ss >= 0.0:
        F = FL + SL*(ustar(rl, ul, vl, wwl, pl, El, SL) - UL)                    # (10.38)
    elif SR > 0.0:
        F = FR + SR*(ustar(rr, ur, vr, wwr, pr, Er, SR) - UR)
    else:
        F = FR
    return F, bool(pstar > 0.0 and SL < Ss < SR)


# ---------------------------------------------------------------------------------------------
def ppm_cw(a):
    """a = 1-D array of cell averages on a uniform grid.  Returns (aL, aR): the limited parabola's
    left and right edge values of every cell j in [2, n-3] (entries outside are NaN)."""
    n = len(a)
    da = np.full(n, np.nan)
    da[1:-1] = 0.5*(a[2:] - a[:-2])                       # average slope, CW (1.7), not limited
    ah = np.full(n, np.nan)                               # ah[j] = a_{j+1/2}
    for j in range(1, n - 2):
        ah[j] = a[j] + 0.5*(a[j+1] - a[j]) - (da[j+1] - da[j])/6.0        # CW (1.6)
        ah[j] = min(max(ah[j], min(a[j], a[j+1])), max(a[j], a[j+1]))     # CS08 eq. 13
    aL = np.full(n, np.nan)
    aR = np.full(n, np.nan)
    for j in range(2, n - 2):
        L, R, c = ah[j-1], ah[j], a[j]
        if (R - c)*(c - L) <= 0.0:                        # CW (1.10), first line
            L = R = c
        else:
            d = R - L
            m = c - 0.5*(L + R)
            L0, R0 = L, R
            if d*m > d*d/6.0:                             # second line
                L = 3.0*c - 2.0*R0
            if -d*d/6.0 > d*m:                            # third line
                R = 3.0*c - 2.0*L0
        aL[j], aR[j] = L, R
    return aL, aR


# ---------------------------------------------------------------------------------------------
def _sh(A, ax, n):
    """A shifted so that result[idx] = A[idx + n] along axis ax (wraps; callers use the interior)"""
    return np.roll(A, -n, axis=ax)


def corner_emf_gs(Ecc, Ea, Eb, Ma, Mb, axa, axb, da, db):
    """One component E_c of the corner (cell-edge) EMF; (a, b, c) is a cyclic permutation of the
    axes.  All arrays are cell-shaped 3-D: Ecc = cell-centred E_c; Ea = E_c on a-faces (entry idx =
    the face on the LOW a side of cell idx); Eb likewise on b-faces; Ma, Mb = mass flux on the same
    faces.  Result[idx] = E_c on the edge at the low-a, low-b corner of cell idx."""
    # derivative estimates on the two halves of a cell, eq. 45
    dEdb_lo = (Ecc - Eb)/(0.5*db)                # between the low b-face of the cell and its centre
    dEdb_hi = (_sh(Eb, axb, 1) - Ecc)/(0.5*db)   # between the centre and the high b-face
    dEda_lo = (Ecc - Ea)/(0.5*da)
    dEda_hi = (_sh(Ea, axa, 1) - Ecc)/(0.5*da)

    def upwind(D, M, ax):
        """D lives in cells, wanted on the low face along ax of cell idx: eq. 50"""
        return np.where(M >= 0.0, _sh(D, ax, -1), D)
    # on the a-face through the corner: dE/db a quarter cell below and above the corner.  Below the
    # corner means the high half of the cells of row b-1; above = the low half of row b.
    dEdb_below = _sh(upwind(dEdb_hi, Ma, axa), axb, -1)
    dEdb_above = upwind(dEdb_lo, Ma, axa)
    dEda_below = _sh(upwind(dEda_hi, Mb, axb), axa, -1)
    dEda_above = upwind(dEda_lo, Mb, axb)
    return (0.25*(Ea + _sh(Ea, axb, -1) + Eb + _sh(Eb, axa, -1))
            + db/8.0*(dEdb_below - dEdb_above) + da/8.0*(dEda_below - dEda_above))      # eq. 41


# ---------------------------------------------------------------------------------------------
def cons_from_prim(g, w, bcc=None):
    """w = (rho, vx, vy, vz, e_int) stacked on axis 0 -> (rho, Mx, My, Mz, E_total)"""
    r, vx, vy, vz, ei = w
    E = ei + 0.5*r*(vx**2 + vy**2 + vz**2)
    if bcc is not None:
        E = E + 0.5*(bcc[0]**2 + bcc[1]**2 + bcc[2]**2)
    return np.stack([r, r*vx, r*vy, r*vz, E])


# ---------------------------------------------------------------------------------------------
def plm_van_leer(q):
    """Piecewise-linear face values of a row of cell averages with van Leer's harmonic limiter in its
    flux-limiter form (van Leer 1974; LeVeque 2002 eq. 6.39b): slope_i = phi(r_i) * (q_{i+1} - q_i),
    r_i = (q_i - q_{i-1}) / (q_{i+1} - q_i), phi(r) = (r + |r|) / (1 + |r|).  Returns (left, right) edge
    values of every cell (ends unused)."""
    q = np.asarray(q, dtype=float)
    dL = np.zeros_like(q)
    dR = np.zeros_like(q)
    dL[1:] = q[1:] - q[:-1]
    dR[:-1] = q[1:] - q[:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = dL/dR
        phi = (r + np.abs(r))/(1.0 + np.abs(r))
        slope = np.where(dR != 0.0, phi*dR, 0.0)
    slope = np.where(np.isfinite(slope), slope, 0.0)
    slope = np.where(dL*dR > 0.0, slope, 0.0)
    return q - 0.5*slope, q + 0.5*slope


def faraday_circulation(E, axis, d):
    """-(1/area) * closed line integral of E around every face normal to `axis`, counter-clockwise seen from
    +axis (Stokes: dB_n/dt = -(curl E)_n).  E[c] = component c on the edges parallel to c, as arrays whose
    index (k, j, i) is the edge at the LOW corner of cell (k, j, i) in the two directions transverse to c;
    all three are cropped to a common cell-shaped (K, J, I) + 1 layout by the caller.  Array axes: x1 is the
    last.  Returns dB_n/dt on the low face of every cell."""
    b, c = (axis + 1) % 3, (axis + 2) % 3            # (n, b, c) right-handed
    ax = {0: 2, 1: 1, 2: 0}
    Eb, Ec = E[b], E[c]
    hi = lambda A, comp: np.roll(A, -1, axis=ax[comp])
    # walk: along +b at low c, along +c at high b, along -b at high c, along -c at low b
    circ = Eb*d[b] + hi(Ec, b)*d[c] - hi(Eb, c)*d[b] - Ec*d[c]
    return -circ/(d[b]*d[c])
