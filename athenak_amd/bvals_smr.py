"""Boundary values of a statically refined MeshBlockPack.

Host mirror of MeshBoundaryValuesCC / MeshBoundaryValuesFC for multilevel meshes
(src/bvals/bvals.hpp:134-267): the 56 MeshBoundaryBuffers with their index ranges
(InitializeBuffers src/bvals/bvals.cpp:322-439, InitSendIndices/InitRecvIndices
src/bvals/buffs_cc.cpp, src/bvals/buffs_fc.cpp) become flat tables on the device and every task body
(PackAndSend*/RecvAndUnpack*, FillCoarseInBndry*, Prolongate*, *FluxCC, *FluxFC) is one call through
the C ABI (include/akmi.h, akmi_smr_*).  The index ranges are generated from ONE rule per direction:
with `o` the offset of the slot along the direction, `f` the sub-block flag that applies to it (the
tangential directions of a slot take f1, f2 in order) and `a` = 1 where a face-field component has
its extra face, every range of the reference's tables is an interval of

      lo = base_lo(o) + shift,   hi = base_hi(o) + shift

whose pieces are listed in `_interval` below; the test suite holds a case-by-case restatement of the
reference's formulas and compares the two tables entry by entry.

Neighbours on another rank are not supported on this path yet (one pack per process holds the whole
mesh); Mesh refuses `refinement = static` with more than one rank.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

from . import capi
from .mesh_tree import NeighborIndex
from .tasklist import TaskStatus

KINDS = ("same", "coar", "fine", "prol", "flxs", "flxc")      # order of the device tables


def slot_list(ndim, multilevel):
    """(slot, ox1, ox2, ox3, f1, f2) of every buffer InitializeBuffers sets up."""
    nfx = 2 if multilevel else 1
    nfy = 2 if multilevel and ndim > 1 else 1
    nfz = 2 if multilevel and ndim > 2 else 1
    out = []
    for n in (-1, 1):
        for fz in range(nfz):
            for fy in range(nfy):
                out.append((NeighborIndex(n, 0, 0, fy, fz), n, 0, 0, fy, fz))
    if ndim > 1:
        for m in (-1, 1):
            for fz in range(nfz):
                for fx in range(nfx):
                    out.append((NeighborIndex(0, m, 0, fx, fz), 0, m, 0, fx, fz))
        for m in (-1, 1):
            for n in (-1, 1):
                for fz in range(nfz):
                    out.append((NeighborIndex(n, m, 0, fz, 0), n, m, 0, fz, 0))
    if ndim > 2:
        for l in (-1, 1):
            for fy in range(nfy):
                for fx in range(nfx):
                    out.append((NeighborIndex(0, 0, l, fx, fy), 0, 0, l, fx, fy))
        for l in (-1, 1):
            for n in (-1, 1):
                for fy in range(nfy):
                    out.append((NeighborIndex(n, 0, l, fy, 0), n, 0, l, fy, 0))
        for l in (-1, 1):
            for m in (-1, 1):
                for fx in range(nfx):
                    out.append((NeighborIndex(0, m, l, fx, 0), 0, m, l, fx, 0))
        for l in (-1, 1):
            for m in (-1, 1):
                for n in (-1, 1):
                    out.append((NeighborIndex(n, m, l, 0, 0), n, m, l, 0, 0))
    return out


def _interval(kind, send, s, e, cs, ce, cnx, ng, act, o, f, a, st, ml_oth):
    """[lo, hi] of one direction of one index table.
    s,e / cs,ce: fine / coarse active range; act: the direction exists; o: slot offset along it;
    f: sub-block flag; a: 1 for the direction a face-field component is normal to (0 for
    cell-centred data); st: 1 where an edge-field component is staggered (EMF tables);
    ml_oth: multilevel mesh and the slot is offset in another direction (face fields only)."""
    if kind in ("flxs", "flxc"):
        # one layer on the block surface; flxc of a sender is in coarse indices
        if send:
            lo, hi = (cs, ce) if kind == "flxc" else (s, e)
        else:
            lo, hi = s, e
        if o == 0:
            hi += st
            if not send and kind == "flxc" and act:
                lo, hi = (lo + cnx, hi) if f == 1 else (lo, hi - cnx)
            return lo, hi
        edge = (hi + 1) if o > 0 else lo
        return edge, edge
    if send:
        lo, hi = (cs, ce) if kind == "coar" else (s, e)
        if o == 0:
            hi += a
            if kind == "fine" and act:
                lo, hi = (lo + cnx - ng, hi) if f == 1 else (lo, hi - (cnx - ng))
            return lo, hi
        if o > 0:
            lo, hi = hi - ng + 1, hi
            if kind == "fine":
                hi += a
            elif a and ml_oth:
                hi += 1
        else:
            lo, hi = lo, lo + ng - 1
            if kind == "fine":
                hi += a
            else:
                lo, hi = lo + a, hi + a
                if a and ml_oth:
                    lo -= 1
        return lo, hi
    # receive side
    n = ng//2 if kind == "prol" else ng
    lo, hi = (cs, ce) if kind in ("coar", "prol") else (s, e)
    if o == 0:
        hi += a
        if act:
            if kind in ("coar", "prol"):
                lo, hi = (lo, hi + n) if f == 0 else (lo - n, hi)
            elif kind == "fine":
                lo, hi = (lo + cnx, hi) if f == 1 else (lo, hi - cnx)
        return lo, hi
    if o > 0:
        lo, hi = hi + 1 + a, hi + n + a
        if kind == "coar":
            lo -= a                      # the shared coarse face travels with the data from a coarser block
        elif kind in ("same", "fine") and a and ml_oth:
            lo -= 1
    else:
        lo, hi = lo - n, lo - 1
        if kind == "coar":
            hi += a
        elif kind in ("same", "fine") and a and ml_oth:
            hi += 1
    return lo, hi


def index_tables(indcs, ndim, multilevel):
    """cc_tab, fc_tab [2][6][56][3][6] and ndat [2][56][2][5] as int32 arrays."""
    ng = indcs.ng
    S = (indcs.is_, indcs.js, indcs.ks)
    E = (indcs.ie, indcs.je, indcs.ke)
    nx = (indcs.nx1, indcs.nx2, indcs.nx3)
    cnx = (indcs.nx1//2, indcs.nx2//2 if nx[1] > 1 else 1, indcs.nx3//2 if nx[2] > 1 else 1)
    CS = (ng, ng if nx[1] > 1 else 0, ng if nx[2] > 1 else 0)
    CE = tuple(CS[d] + cnx[d] - 1 if (d == 0 or nx[d] > 1) else 0 for d in range(3))
    act = (True, nx[1] > 1, nx[2] > 1)
    tabs = [np.zeros((2, 6, 56, 3, 6), dtype=np.int32) for _ in range(2)]
    ndat = np.zeros((2, 56, 2, 5), dtype=np.int32)
    for (n, ox1, ox2, ox3, f1, f2) in slot_list(ndim, multilevel):
        o = (ox1, ox2, ox3)
        fl = (f1, f1 if ox1 != 0 else f2, f1 if (ox1 != 0 and ox2 != 0) else f2)
        for fc in (0, 1):
            for sr in (0, 1):
                for ki, kind in enumerate(KINDS):
                    if kind == "same" and (f1 or f2):
                        continue
                    if kind == "prol" and sr == 0:
                        continue
                    if kind == "flxs" and not fc:
                        continue
                    for v in range(3 if fc else 1):
                        for d in range(3):
                            a = 1 if (fc and v == d) else 0
                            st = 1 if (fc and v != d) else 0
                            oth = bool(fc and multilevel and any(o[q] != 0 for q in range(3) if q != d))
                            lo, hi = _interval(kind, sr == 0, S[d], E[d], CS[d], CE[d], cnx[d], ng, act[d],
                                               o[d], fl[d], a, st, oth)
                            tabs[fc][sr, ki, n, v, 2*d] = lo
                            tabs[fc][sr, ki, n, v, 2*d + 1] = hi
                # data counts: max over the components
                for qi, kind in enumerate(("same", "coar", "fine", "flxs", "flxc")):
                    ki = KINDS.index(kind)
                    if (kind == "same" and (f1 or f2)) or (kind == "flxs" and not fc):
                        continue
                    b = tabs[fc][sr, ki, n]
                    cnt = [(b[v, 1] - b[v, 0] + 1)*(b[v, 3] - b[v, 2] + 1)*(b[v, 5] - b[v, 4] + 1)
                           for v in range(3 if fc else 1)]
                    ndat[fc, n, sr, qi] = max(cnt)
    return tabs[0], tabs[1], ndat


def edge_counts(nghbr, mblev, nnghbr):
    """nflx[nmb][48]: owners of each block edge after SumBoundaryFluxes(same level),
    ZeroFluxesAtBoundaryWithFiner, SumBoundaryFluxes(finer) (src/bvals/flux_correct_fc.cpp:445-790);
    the counting touches the neighbour table only."""
    face_edges = {0: (16, 20, 32, 36), 4: (18, 22, 34, 38), 8: (16, 18, 40, 44), 12: (20, 22, 42, 46),
                  24: (32, 34, 40, 42), 28: (36, 38, 44, 46)}
    nmb = len(mblev)
    nflx = np.ones((nmb, 48), dtype=np.int32)

    def add(m, want_finer):
        for n in range(min(nnghbr, 48)):
            g, l = nghbr[m, n, 0], nghbr[m, n, 1]
            if g < 0 or (l > mblev[m]) != want_finer or l < mblev[m]:
                continue
            if n in face_edges:
                for q in face_edges[n]:
                    nflx[m, q] += 1
            elif 16 <= n < 24 or 32 <= n < 48:
                nflx[m, n] += 1
    for m in range(nmb):
        add(m, False)
        for n in range(min(nnghbr, 48)):
            if nghbr[m, n, 0] >= 0 and nghbr[m, n, 1] > mblev[m]:
                if n in face_edges:
                    for q in face_edges[n]:
                        nflx[m, q] = 0
                elif 16 <= n < 24 or 32 <= n < 48:
                    nflx[m, n] = 0
        add(m, True)
    return nflx


class HipSmrKernels:
    """the akmi_smr_* entry points (device tensors only)"""

    def __init__(self):
        self.L = capi.lib()

    def _call(self, name, *args):
        capi.check(getattr(self.L, name)(*args), name)

    def exchange_cc(self, pack, smr, nvar, u, cu, buf):
        self._call("akmi_smr_exchange_cc", C.byref(pack), C.byref(smr), nvar, capi._p(u), capi._p(cu),
                   capi._p(buf), capi._stream())

    def exchange_fc(self, pack, smr, b, cb, buf):
        self._call("akmi_smr_exchange_fc", C.byref(pack), C.byref(smr), capi._p(b.x1f), capi._p(b.x2f),
                   capi._p(b.x3f), capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f), capi._p(buf),
                   capi._stream())

    def fill_coarse_cc(self, pack, smr, nvar, u, cu):
        self._call("akmi_smr_fill_coarse_cc", C.byref(pack), C.byref(smr), nvar, capi._p(u), capi._p(cu),
                   capi._stream())

    def fill_coarse_fc(self, pack, smr, b, cb):
        self._call("akmi_smr_fill_coarse_fc", C.byref(pack), C.byref(smr), capi._p(b.x1f), capi._p(b.x2f),
                   capi._p(b.x3f), capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f), capi._stream())

    def prolong_cc(self, pack, smr, nvar, cu, u):
        self._call("akmi_smr_prolong_cc", C.byref(pack), C.byref(smr), nvar, capi._p(cu), capi._p(u),
                   capi._stream())

    def c2p_coarse(self, pack, smr, nvar, cu, cb, cw):
        f = (capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f)) if cb is not None else (None, None, None)
        self._call("akmi_smr_c2p_coarse", C.byref(pack), C.byref(smr), nvar, capi._p(cu), *f, capi._p(cw),
                   capi._stream())

    def p2c_fine(self, pack, smr, nvar, w, b, u):
        f = (capi._p(b.x1f), capi._p(b.x2f), capi._p(b.x3f)) if b is not None else (None, None, None)
        self._call("akmi_smr_p2c_fine", C.byref(pack), C.byref(smr), nvar, capi._p(w), *f, capi._p(u),
                   capi._stream())

    def gather_same(self, pack, nvar, tab27, u):
        self._call("akmi_bvals_cc_local", C.byref(pack), nvar, capi._p(tab27), capi._p(u), capi._stream())

    def prolong_fc(self, pack, smr, cb, b):
        self._call("akmi_smr_prolong_fc", C.byref(pack), C.byref(smr), capi._p(cb.x1f), capi._p(cb.x2f),
                   capi._p(cb.x3f), capi._p(b.x1f), capi._p(b.x2f), capi._p(b.x3f), capi._stream())

    def flux_cc(self, pack, smr, nvar, face_shaped, flx, buf):
        self._call("akmi_smr_flux_cc", C.byref(pack), C.byref(smr), nvar, int(face_shaped), capi._p(flx.x1f),
                   capi._p(flx.x2f), capi._p(flx.x3f), capi._p(buf), capi._stream())

    def emf_exchange(self, pack, smr, nflx, efld, buf):
        self._call("akmi_smr_emf_exchange", C.byref(pack), C.byref(smr), capi._p(nflx), capi._p(efld.x1e),
                   capi._p(efld.x2e), capi._p(efld.x3e), capi._p(buf), capi._stream())

    # the two halves of each exchange (PackAndSend* | RecvAndUnpack*): messages travel in between
    def pack_cc(self, pack, smr, nvar, u, cu, buf):
        self._call("akmi_smr_pack_cc", C.byref(pack), C.byref(smr), nvar, capi._p(u), capi._p(cu), capi._p(buf),
                   capi._stream())

    def unpack_cc(self, pack, smr, nvar, buf, u, cu):
        self._call("akmi_smr_unpack_cc", C.byref(pack), C.byref(smr), nvar, capi._p(buf), capi._p(u), capi._p(cu),
                   capi._stream())

    def pack_fc(self, pack, smr, b, cb, buf):
        self._call("akmi_smr_pack_fc", C.byref(pack), C.byref(smr), capi._p(b.x1f), capi._p(b.x2f), capi._p(b.x3f),
                   capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f), capi._p(buf), capi._stream())

    def unpack_fc(self, pack, smr, buf, b, cb):
        self._call("akmi_smr_unpack_fc", C.byref(pack), C.byref(smr), capi._p(buf), capi._p(b.x1f), capi._p(b.x2f),
                   capi._p(b.x3f), capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f), capi._stream())

    def pack_flux_cc(self, pack, smr, nvar, face_shaped, flx, buf):
        self._call("akmi_smr_pack_flux_cc", C.byref(pack), C.byref(smr), nvar, 1 if face_shaped else 0,
                   capi._p(flx.x1f), capi._p(flx.x2f), capi._p(flx.x3f), capi._p(buf), capi._stream())

    def unpack_flux_cc(self, pack, smr, nvar, face_shaped, buf, flx):
        self._call("akmi_smr_unpack_flux_cc", C.byref(pack), C.byref(smr), nvar, 1 if face_shaped else 0,
                   capi._p(buf), capi._p(flx.x1f), capi._p(flx.x2f), capi._p(flx.x3f), capi._stream())

    def pack_emf(self, pack, smr, efld, buf):
        self._call("akmi_smr_pack_emf", C.byref(pack), C.byref(smr), capi._p(efld.x1e), capi._p(efld.x2e),
                   capi._p(efld.x3e), capi._p(buf), capi._stream())

    def unpack_emf(self, pack, smr, nflx, buf, efld):
        self._call("akmi_smr_unpack_emf", C.byref(pack), C.byref(smr), capi._p(nflx), capi._p(buf),
                   capi._p(efld.x1e), capi._p(efld.x2e), capi._p(efld.x3e), capi._stream())

    def restrict_cc(self, pack, nvar, u, cu, mask=None):
        self._call("akmi_restrict_cc_masked", C.byref(pack), nvar, capi._p(mask) if mask is not None else None,
                   capi._p(u), capi._p(cu), capi._stream())

    def restrict_fc(self, pack, b, cb, mask=None):
        self._call("akmi_restrict_fc_masked", C.byref(pack), capi._p(mask) if mask is not None else None,
                   capi._p(b.x1f), capi._p(b.x2f), capi._p(b.x3f),
                   capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f), capi._stream())



def _map_or_fallback(build, name):
    """A copy list is an optimisation of the pack / unpack kernels, which stay in the library: when the library cannot
    build it (index space of 2^31 elements, no memory for the set-up scratch, an unclassifiable pair) warn once and use
    the kernels.  AKMI_SMR_<name>_MAP=1 makes the list mandatory (the tests of the lists themselves)."""
    try:
        return build()
    except capi.AkmiError as e:
        if os.environ.get("AKMI_SMR_%s_MAP" % name) == "1":
            raise
        sys.stderr.write("### WARNING %s -- falling back to the pack / unpack kernels\n" % e)
        return False


class MeshBoundaryValuesSMR:
    """level-aware boundary values of one pack: tables on the device + the task bodies"""

    def __init__(self, ppack, nvar, kernels=None, device=None):
        device = device or capi.DEVICE
        self.pmy_pack = ppack
        self.device = device
        self.k = kernels if kernels is not None else HipSmrKernels()
        pm, pmb = ppack.pmesh, ppack.pmb
        indcs = pm.mb_indcs
        ndim = 3 if pm.three_d else (2 if pm.multi_d else 1)
        self.nnghbr = 56 if ndim == 3 else (24 if ndim == 2 else 8)
        self.nvar = nvar
        nmb = ppack.nmb_thispack
        cc, fc, ndat = index_tables(indcs, ndim, pm.multilevel)
        self.cc_tab_host, self.fc_tab_host, self.ndat_host = cc, fc, ndat
        # neighbour table: NeighborBlock {gid, lev, rank, dest} -> {local index, level, dest}; a
        # neighbour on another rank is marked by the index nmb (tested for existence only, see soff/roff)
        ng = -np.ones((nmb, 56, 3), dtype=np.int32)
        remote = []                     # (m, n, NeighborBlock) on other ranks
        for m in range(nmb):
            for n, nb in pmb.nghbr[m].items():
                if nb.rank != pm.my_rank:
                    ng[m, n] = (nmb, nb.lev, nb.dest)
                    remote.append((m, n, nb))
                else:
                    ng[m, n] = (nb.gid - ppack.gids, nb.lev, nb.dest)
        self.nghbr_host = ng
        self.nflx_host = edge_counts(ng, pmb.mb_lev, self.nnghbr)
        slot_ox = np.zeros((56, 3), dtype=np.int32)
        for (n, ox1, ox2, ox3, f1, f2) in slot_list(ndim, pm.multilevel):
            slot_ox[n] = (ox1, ox2, ox3)
        # receive buffers: per slot nvar*max(ndat) doubles per block, slot after slot
        layout = np.zeros((4, 56, 2), dtype=np.int64)
        sizes = []
        for cls, (fcq, nv, cols) in enumerate(((0, nvar, (0, 1, 2)), (0, nvar, (4,)), (1, 3, (0, 1, 2)),
                                               (1, 3, (3, 4)))):
            off = 0
            for n in range(56):
                stride = nv*int(ndat[fcq, n][:, list(cols)].max())
                layout[cls, n] = (off, stride)
                off += stride*nmb
            sizes.append(off)
        self.layout_host = layout
        self._plan_ranks(remote, layout, sizes)

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.t_nghbr, self.t_lev = dev(ng), dev(pmb.mb_lev.astype(np.int32))
        self.t_cc, self.t_fc, self.t_ndat = dev(cc), dev(fc), dev(ndat)
        self.t_ox, self.t_layout, self.t_nflx = dev(slot_ox), dev(layout), dev(self.nflx_host)
        self.buf = [torch.zeros(max(sz, 1), dtype=torch.float64, device=device) for sz in self.buf_sizes]
        self.t_soff = dev(self.soff_host) if self.peers else None
        self.t_roff = dev(self.roff_host) if self.peers else None
        # same-level neighbours in this pack: one direct gather (akmi_bvals_cc_local with the 27-direction table of
        # the uniform-mesh path) instead of pack -> buffer -> unpack; the SMR kernels skip those slots
        same = -np.ones((nmb, 27), dtype=np.int32)
        for m in range(nmb):
            for o3 in (-1, 0, 1):
                for o2 in (-1, 0, 1):
                    for o1 in (-1, 0, 1):
                        if (o1, o2, o3) == (0, 0, 0) or (ndim < 3 and o3) or (ndim < 2 and o2):
                            continue
                        n = NeighborIndex(o1, o2, o3, 0, 0)
                        nb = pmb.nghbr[m].get(n)
                        if nb is not None and nb.lev == int(pmb.mb_lev[m]) and nb.rank == pm.my_rank:
                            same[m, (o3 + 1)*9 + (o2 + 1)*3 + (o1 + 1)] = nb.gid - ppack.gids
        self.t_same = dev(same)
        # blocks with a coarser neighbour: the only ones whose coarse ghost zones are ever read (by their prolongation)
        needs = np.array([1 if any(nb.lev < int(pmb.mb_lev[m]) for nb in pmb.nghbr[m].values()) else 0
                          for m in range(nmb)], dtype=np.uint8)
        self.t_needs = dev(needs)
        self.direct_same = 1 if os.environ.get("AKMI_SMR_DIRECT", "1") != "0" else 0
        self.smr_c = capi.Smr(self.nnghbr, 1 if pm.multilevel else 0, self.t_nghbr.data_ptr(),
                              self.t_lev.data_ptr(), self.t_cc.data_ptr(), self.t_fc.data_ptr(),
                              self.t_ndat.data_ptr(), self.t_ox.data_ptr(), self.t_layout.data_ptr(),
                              self.t_soff.data_ptr() if self.peers else None,
                              self.t_roff.data_ptr() if self.peers else None, self.direct_same,
                              self.t_needs.data_ptr())
        self.pack_c = None
        self._fcmap = None
        self._ccmap = None
        self._works = [[], [], [], []]
        self._hsend = [None]*4
        self._hrecv = [None]*4

    # ---- ranks: where the segments of off-rank neighbours live and which slices travel -------------
    @staticmethod
    def carries(cls, lev_s, lev_r, dn):
        """does the sender (level lev_s) fill the segment of the receiver's slot dn (level lev_r)?
        Variables always; restricted fluxes go from fine to coarse across faces
        (flux_correct_cc.cpp:60-76); edge EMFs to same-level and coarser neighbours across faces and
        edges (flux_correct_fc.cpp:60-76)."""
        if cls in (0, 2):
            return True
        if cls == 1:
            return lev_s > lev_r and (dn < 16 or 24 <= dn < 32)
        return lev_s >= lev_r and dn < 48

    def _plan_ranks(self, remote, layout, local_sizes):
        """Per class one buffer: [segments of neighbours in this pack, by (slot, block)] [segments
        received from other ranks, rank after rank] [segments sent to other ranks, rank after rank].
        Inside a message the segments are ordered by (receiver gid, receiver slot) -- both sides
        derive that order from the tree alone (no header exchange).  A segment has the size of the
        receiver's slot (layout stride), whatever the level relation."""
        ppack = self.pmy_pack
        pm, pmb = ppack.pmesh, ppack.pmb
        nmb, gids = ppack.nmb_thispack, ppack.gids
        self.peers = sorted({nb.rank for (_, _, nb) in remote})
        soff = np.zeros((4, nmb, 56), dtype=np.int64)
        roff = np.zeros((4, nmb, 56), dtype=np.int64)
        for cls in range(4):
            for m in range(nmb):
                for n, nb in pmb.nghbr[m].items():
                    if nb.rank == pm.my_rank:
                        soff[cls, m, n] = layout[cls, nb.dest, 0] + (nb.gid - gids)*layout[cls, nb.dest, 1]
                    roff[cls, m, n] = layout[cls, n, 0] + m*layout[cls, n, 1]
        self.buf_sizes = list(local_sizes)
        self.send_slices = [dict() for _ in range(4)]
        self.recv_slices = [dict() for _ in range(4)]
        for cls in range(4):
            off = local_sizes[cls]
            for r in self.peers:                       # what I receive from rank r
                start = off
                items = sorted((gids + m, n) for (m, n, nb) in remote if nb.rank == r and layout[cls, n, 1] > 0 and
                               self.carries(cls, nb.lev, int(pmb.mb_lev[m]), n))
                for (g, n) in items:
                    roff[cls, g - gids, n] = off
                    off += int(layout[cls, n, 1])
                self.recv_slices[cls][r] = (start, off)
            for r in self.peers:                       # what I send to rank r
                start = off
                items = sorted((nb.gid, nb.dest, m, n) for (m, n, nb) in remote if nb.rank == r and
                               layout[cls, nb.dest, 1] > 0 and self.carries(cls, int(pmb.mb_lev[m]), nb.lev, nb.dest))
                for (g, dn, m, n) in items:
                    soff[cls, m, n] = off
                    off += int(layout[cls, dn, 1])
                self.send_slices[cls][r] = (start, off)
            self.buf_sizes[cls] = off
        self.soff_host, self.roff_host = soff, roff

    def _staged(self):
        import torch.distributed as dist
        return torch.device(self.device).type == "cuda" and dist.get_backend() != "nccl"

    def _post(self, cls):
        """the messages of one class (bvals.cpp:134-310: one per peer rank)"""
        if not self.peers:
            return
        import torch.distributed as dist
        buf = self.buf[cls]
        sbuf = rbuf = buf
        if self._staged():
            if self._hsend[cls] is None:
                self._hsend[cls] = torch.empty(buf.shape, dtype=torch.float64, pin_memory=True)
                self._hrecv[cls] = torch.empty(buf.shape, dtype=torch.float64, pin_memory=True)
            for r in self.peers:
                a, b = self.send_slices[cls][r]
                self._hsend[cls][a:b].copy_(buf[a:b], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            sbuf, rbuf = self._hsend[cls], self._hrecv[cls]
        ops = []
        for r in self.peers:
            a, b = self.recv_slices[cls][r]
            if b > a:
                ops.append(dist.P2POp(dist.irecv, rbuf[a:b], r))
        for r in self.peers:
            a, b = self.send_slices[cls][r]
            if b > a:
                ops.append(dist.P2POp(dist.isend, sbuf[a:b], r))
        self._works[cls] = dist.batch_isend_irecv(ops) if ops else []

    def _wait(self, cls):
        if not self.peers:
            return
        for w in self._works[cls]:
            w.wait()
        self._works[cls] = []
        if self._hrecv[cls] is not None:
            for r in self.peers:
                a, b = self.recv_slices[cls][r]
                self.buf[cls][a:b].copy_(self._hrecv[cls][a:b], non_blocking=True)

    def set_pack(self, pack_c):
        self.pack_c = pack_c
        # work lists of (block, slot) pairs (akmi_smr::lists, include/akmi.h): the SMR kernels are launched over the
        # pairs that can have work instead of all nmb*56.  Built once by the library from the tables above.
        # AKMI_SMR_LISTS=0: A/B switch.  (The CPU test backend has no use for them.)
        if capi.DEVICE != "cpu" and os.environ.get("AKMI_SMR_LISTS", "1") != "0" and self.smr_c.lists is None:
            import torch
            nmb = int(pack_c.nmb)
            self.t_lists = torch.zeros(2*nmb*56*6, dtype=torch.int32, device=self.device)
            cnt = (C.c_int*6)()
            capi.check(capi.lib().akmi_smr_build_lists(C.byref(pack_c), C.byref(self.smr_c), capi._p(self.t_lists), cnt,
                                                       capi._stream()), "smr_build_lists")
            self.smr_c.lists = self.t_lists.data_ptr()
            for q in range(6):
                self.smr_c.list_cnt[q] = cnt[q]

    # ---- task bodies ----------------------------------------------------------------------------
    def RestrictCC(self, u, cu):
        self.k.restrict_cc(self.pack_c, self.nvar, u, cu, self.t_needs)     # only blocks with a coarser neighbour
        return TaskStatus.complete

    def RestrictFC(self, b, cb):
        self.k.restrict_fc(self.pack_c, b, cb, self.t_needs)
        return TaskStatus.complete

    def _cc_maps(self):
        """one rank: the cell-centred exchange (pack + unpack across levels, direct same-level gather) as one list of
        element copies applied to every variable (include/akmi.h, akmi_smr_cc_map); AKMI_SMR_CC_MAP=0: the three
        kernels (A/B switch).  HIP library only; with ranks the messages go through the buffers."""
        if self._ccmap is None:
            self._ccmap = False
            if capi.DEVICE != "cpu" and not self.peers and os.environ.get("AKMI_SMR_CC_MAP", "1") != "0":
                def build():
                    import torch
                    L, buf = capi.lib(), self.buf[0]
                    args = (C.byref(self.pack_c), C.byref(self.smr_c), self.nvar, capi._p(self.t_same), capi._p(buf),
                            C.c_longlong(int(buf.numel())))
                    tail = C.c_longlong(0)
                    n = int(L.akmi_smr_cc_map(*args, None, C.c_longlong(0), C.byref(tail), capi._stream()))
                    capi.check(n, "smr_cc_map")
                    m = torch.zeros(max(2*n, 2), dtype=torch.int32, device=self.device)
                    capi.check(int(L.akmi_smr_cc_map(*args, capi._p(m), C.c_longlong(n), C.byref(tail), capi._stream())),
                               "smr_cc_map")
                    return (m, n, int(tail.value))
                self._ccmap = _map_or_fallback(build, "CC")
        return self._ccmap

    def PackAndSendCC(self, u, cu):
        if not self._cc_maps():
            self.k.pack_cc(self.pack_c, self.smr_c, self.nvar, u, cu, self.buf[0])
        self._post(0)
        return TaskStatus.complete

    def RecvAndUnpackCC(self, u, cu):
        self._wait(0)
        if self._cc_maps():
            m, n, tail = self._ccmap
            capi.check(capi.lib().akmi_smr_cc_copy(C.byref(self.pack_c), self.nvar, capi._p(m), C.c_longlong(n),
                                                   C.c_longlong(tail), capi._p(u), capi._p(cu), capi._stream()),
                       "smr_cc_copy")
            return TaskStatus.complete
        self.k.unpack_cc(self.pack_c, self.smr_c, self.nvar, self.buf[0], u, cu)
        if self.direct_same:
            self.k.gather_same(self.pack_c, self.nvar, self.t_same, u)
        return TaskStatus.complete

    def _fc_maps(self):
        """the face-field exchange as lists of element copies (include/akmi.h, akmi_smr_fc_map): [1] buffer <- arrays for
        the messages to other ranks, [0] arrays <- arrays of this pack / received part of the buffer.  Built at the
        first exchange; AKMI_SMR_FC_MAP=0: pack + slot-by-slot unpack (A/B switch).  HIP library only."""
        if self._fcmap is None:
            self._fcmap = False
            if capi.DEVICE != "cpu" and os.environ.get("AKMI_SMR_FC_MAP", "1") != "0":
                def build():
                    import torch
                    L, buf = capi.lib(), self.buf[2]
                    nb = int(buf.numel())
                    lo = min([a for (a, _) in self.send_slices[2].values()], default=nb) if self.peers else nb
                    maps = []
                    for which in (0, 1):
                        args = (C.byref(self.pack_c), C.byref(self.smr_c), capi._p(buf), C.c_longlong(nb), C.c_longlong(lo),
                                C.c_longlong(nb), which)
                        tail = C.c_longlong(0)
                        n = int(L.akmi_smr_fc_map(*args, None, C.c_longlong(0), C.byref(tail), capi._stream()))
                        capi.check(n, "smr_fc_map")
                        m = torch.zeros(max(2*n, 2), dtype=torch.int32, device=self.device)
                        capi.check(int(L.akmi_smr_fc_map(*args, capi._p(m), C.c_longlong(n), C.byref(tail), capi._stream())),
                                   "smr_fc_map")
                        maps.append((m, n, int(tail.value)))
                    return maps
                self._fcmap = _map_or_fallback(build, "FC")
        return self._fcmap

    def _fc_copy(self, which, b, cb):
        m, n, tail = self._fcmap[which]
        capi.check(capi.lib().akmi_smr_fc_copy(
            C.byref(self.pack_c), capi._p(m), C.c_longlong(n), C.c_longlong(tail), C.c_longlong(int(self.buf[2].numel())),
            capi._p(b.x1f),
            capi._p(b.x2f), capi._p(b.x3f), capi._p(cb.x1f), capi._p(cb.x2f), capi._p(cb.x3f), capi._p(self.buf[2]),
            capi._stream()), "smr_fc_copy")

    def PackAndSendFC(self, b, cb):
        if self._fc_maps():
            self._fc_copy(1, b, cb)
        else:
            self.k.pack_fc(self.pack_c, self.smr_c, b, cb, self.buf[2])
        self._post(2)
        return TaskStatus.complete

    def RecvAndUnpackFC(self, b, cb):
        self._wait(2)
        if self._fc_maps():
            self._fc_copy(0, b, cb)
        else:
            self.k.unpack_fc(self.pack_c, self.smr_c, self.buf[2], b, cb)
        return TaskStatus.complete

    def FillCoarseInBndryCC(self, u, cu):
        self.k.fill_coarse_cc(self.pack_c, self.smr_c, self.nvar, u, cu)

    def FillCoarseInBndryFC(self, b, cb):
        self.k.fill_coarse_fc(self.pack_c, self.smr_c, b, cb)

    def ProlongateCC(self, u, cu):
        self.k.prolong_cc(self.pack_c, self.smr_c, self.nvar, cu, u)

    def ProlongateFC(self, b, cb):
        self.k.prolong_fc(self.pack_c, self.smr_c, cb, b)

    def ConsToPrimCoarseBndry(self, cu, cb, cw):
        """prolong_prims.cpp:35-186 (cb None: hydro), 303-461"""
        self.k.c2p_coarse(self.pack_c, self.smr_c, self.nvar, cu, cb, cw)

    def PrimToConsFineBndry(self, w, b, u):
        """prolong_prims.cpp:190-296 (b None: hydro), 465-575"""
        self.k.p2c_fine(self.pack_c, self.smr_c, self.nvar, w, b, u)

    def PackAndSendFluxCC(self, flx, face_shaped):
        self.k.pack_flux_cc(self.pack_c, self.smr_c, self.nvar, face_shaped, flx, self.buf[1])
        self._post(1)
        return TaskStatus.complete

    def RecvAndUnpackFluxCC(self, flx, face_shaped):
        self._wait(1)
        self.k.unpack_flux_cc(self.pack_c, self.smr_c, self.nvar, face_shaped, self.buf[1], flx)
        return TaskStatus.complete

    def PackAndSendFluxFC(self, efld):
        self.k.pack_emf(self.pack_c, self.smr_c, efld, self.buf[3])
        self._post(3)
        return TaskStatus.complete

    def RecvAndUnpackFluxFC(self, efld):
        self._wait(3)
        self.k.unpack_emf(self.pack_c, self.smr_c, self.t_nflx, self.buf[3], efld)
        return TaskStatus.complete
