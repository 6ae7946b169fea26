"""Mesh / MeshBlock / MeshBlockPack: the data model the hot path operates on.

Mirrors src/mesh/mesh.hpp:25-185, mesh.cpp:46-330,573-643, meshblock.cpp:25-131,
meshblock_pack.hpp:44-97, build_tree.cpp:243-268, load_balance.cpp:38-88 for uniform
(single-level) meshes: a root grid of MeshBlocks in Z-order, contiguous chunks of the
Z-ordered list per rank (one MeshBlockPack per rank = one GPU), same-level neighbours with
periodic wrap.  SMR/AMR trees are a later row of SURVEY.md section 8(f).
"""
import math

import numpy as np

from . import capi

FLT_MAX = float(np.finfo(np.float32).max)
FLT_MIN = float(np.finfo(np.float32).tiny)

_BCNAMES = {k: capi.BC[k] for k in ("periodic", "outflow", "reflect", "user", "inflow", "diode", "vacuum")}


def LeftEdgeX(ith, n, xmin, xmax):
    """src/coordinates/cell_locations.hpp:23-28 (round-off symmetric form)"""
    x = np.asarray(ith, dtype=np.float64)/np.float64(n)
    return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax)


def CellCenterX(ith, n, xmin, xmax):
    """src/coordinates/cell_locations.hpp:35-39"""
    x = (np.asarray(ith, dtype=np.float64) + 0.5)/np.float64(n)
    return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax)


class RegionSize:
    """src/mesh/mesh.hpp:25-29"""

    def __init__(self, x1min, x1max, x2min, x2max, x3min, x3max):
        self.x1min, self.x1max = float(x1min), float(x1max)
        self.x2min, self.x2max = float(x2min), float(x2max)
        self.x3min, self.x3max = float(x3min), float(x3max)
        self.dx1 = self.dx2 = self.dx3 = 0.0


class RegionIndcs:
    """src/mesh/mesh.hpp:35-41 + index setup src/mesh/mesh.cpp:285-330"""

    def __init__(self, ng, nx1, nx2, nx3):
        self.ng, self.nx1, self.nx2, self.nx3 = ng, nx1, nx2, nx3
        multi_d, three_d = nx2 > 1, nx3 > 1
        self.is_ = ng
        self.ie = ng + nx1 - 1
        self.js = ng if multi_d else 0
        self.je = ng + nx2 - 1 if multi_d else 0
        self.ks = ng if three_d else 0
        self.ke = ng + nx3 - 1 if three_d else 0
        self.cnx1, self.cnx2, self.cnx3 = nx1//2, max(1, nx2//2), max(1, nx3//2)

    @property
    def ncells(self):
        n1 = self.nx1 + 2*self.ng
        n2 = self.nx2 + 2*self.ng if self.nx2 > 1 else 1
        n3 = self.nx3 + 2*self.ng if self.nx3 > 1 else 1
        return n3, n2, n1


def _morton(x, y, z):
    r = 0
    for b in range(20):
        r |= ((x >> b) & 1) << (3*b)
        r |= ((y >> b) & 1) << (3*b + 1)
        r |= ((z >> b) & 1) << (3*b + 2)
    return r


def LoadBalance(clist, nranks):
    """Mesh::LoadBalance, src/mesh/load_balance.cpp:38-88 -> (rlist, slist, nlist)."""
    nb = len(clist)
    totalcost = float(sum(clist))
    j = nranks - 1
    targetcost = totalcost/nranks
    mycost = 0.0
    rlist = [0]*nb
    for i in range(nb - 1, -1, -1):
        if targetcost == 0.0:
            raise RuntimeError("### FATAL ERROR There is at least one process which has no "
                               "MeshBlock; decrease the number of processes or use smaller "
                               "MeshBlocks.")
        mycost += clist[i]
        rlist[i] = j
        if mycost >= targetcost and j > 0:
            j -= 1
            totalcost -= mycost
            mycost = 0.0
            targetcost = totalcost/(j + 1)
    slist, nlist = [0]*nranks, [0]*nranks
    j = 0
    for i in range(1, nb):
        if rlist[i] != rlist[i - 1]:
            nlist[j] = i - slist[j]
            j += 1
            slist[j] = i
    nlist[j] = nb - slist[j]
    return rlist, slist, nlist


class MeshBlock:
    """Per-block metadata of one pack (src/mesh/meshblock.hpp, meshblock.cpp:25-131)."""

    def __init__(self, pack, igids, nmb):
        pm = pack.pmesh
        ms = pm.mesh_size
        self.pmy_pack = pack
        self.nmb = nmb
        self.mb_gid = np.arange(igids, igids + nmb, dtype=np.int32)
        # logical level of each block: root_level on a uniform mesh (meshblock.cpp:37)
        self.mb_lev = np.array([pm.level_of(igids + m) for m in range(nmb)], dtype=np.int32)
        self.mb_size = [None]*nmb                              # RegionSize per block
        self.mb_bcs = np.zeros((nmb, 6), dtype=np.int32)       # BoundaryFlag per face
        nbr = (pm.nmb_rootx1, pm.nmb_rootx2, pm.nmb_rootx3)
        active = (True, pm.multi_d, pm.three_d)
        mmin = (ms.x1min, ms.x2min, ms.x3min)
        mmax = (ms.x1max, ms.x2max, ms.x3max)
        nxb = (pm.mb_indcs.nx1, pm.mb_indcs.nx2, pm.mb_indcs.nx3)
        for m in range(nmb):
            lloc = pm.lloc_eachmb[igids + m]
            # blocks per direction at the level of this block: nmb_rootx << (lev - root_level)
            nb = tuple(n << (int(self.mb_lev[m]) - pm.root_level) for n in nbr)
            lim = []
            for q in range(3):
                l = lloc[q]
                if not active[q] or l == 0:
                    lo = mmin[q]
                    self.mb_bcs[m, 2*q] = pm.mesh_bcs[2*q]
                else:
                    lo = float(LeftEdgeX(l, nb[q], mmin[q], mmax[q]))
                    self.mb_bcs[m, 2*q] = capi.BC["block"]
                if not active[q] or l == nb[q] - 1:
                    hi = mmax[q]
                    self.mb_bcs[m, 2*q + 1] = pm.mesh_bcs[2*q + 1]
                else:
                    hi = float(LeftEdgeX(l + 1, nb[q], mmin[q], mmax[q]))
                    self.mb_bcs[m, 2*q + 1] = capi.BC["block"]
                lim += [lo, hi]
            rs = RegionSize(*lim)
            rs.dx1 = (rs.x1max - rs.x1min)/float(nxb[0])
            rs.dx2 = (rs.x2max - rs.x2min)/float(nxb[1])
            rs.dx3 = (rs.x3max - rs.x3min)/float(nxb[2])
            self.mb_size[m] = rs
        self.dx = np.array([[s.dx1, s.dx2, s.dx3] for s in self.mb_size], dtype=np.float64)
        self.SetNeighbors()

    def SetNeighbors(self):
        """Same-level neighbour table (reduced NeighborBlock, src/mesh/mesh.hpp:47-52):
        nghbr_gid[m][d], nghbr_rank[m][d] with d=(ox3+1)*9+(ox2+1)*3+(ox1+1); -1 = none.
        Multilevel meshes: nghbr[m] = {slot: NeighborBlock(gid, lev, rank, dest)} with the 56
        slots of NeighborIndex (src/mesh/meshblock.cpp:142-425)."""
        pm = self.pmy_pack.pmesh
        if pm.multilevel:
            from .mesh_tree import SetNeighbors
            self.nghbr = [SetNeighbors(pm.ptree, pm.lloc_eachmb[int(g)], pm.rank_eachmb, True)
                          for g in self.mb_gid]
            self.nghbr_gid = self.nghbr_rank = None
            return
        nb = (pm.nmb_rootx1, pm.nmb_rootx2, pm.nmb_rootx3)
        self.nghbr_gid = -np.ones((self.nmb, 27), dtype=np.int32)
        self.nghbr_rank = -np.ones((self.nmb, 27), dtype=np.int32)
        for m in range(self.nmb):
            lloc = pm.lloc_eachmb[int(self.mb_gid[m])]
            for d in range(27):
                o = (d % 3 - 1, (d//3) % 3 - 1, d//9 - 1)
                if d == 13 or (not pm.multi_d and o[1]) or (not pm.three_d and o[2]):
                    continue
                l, ok = [0, 0, 0], True
                for q in range(3):
                    l[q] = lloc[q] + o[q]
                    if l[q] < 0:
                        if pm.mesh_bcs[2*q] == capi.BC["periodic"]:
                            l[q] += nb[q]
                        else:
                            ok = False
                    elif l[q] >= nb[q]:
                        if pm.mesh_bcs[2*q + 1] == capi.BC["periodic"]:
                            l[q] -= nb[q]
                        else:
                            ok = False
                if ok:
                    gid = pm.gid_of_lloc[tuple(l)]
                    self.nghbr_gid[m, d] = gid
                    self.nghbr_rank[m, d] = pm.rank_eachmb[gid]


class MeshBlockPack:
    """src/mesh/meshblock_pack.hpp:44-97: the MeshBlocks of this rank + physics + task lists."""

    def __init__(self, pm, igids, igide):
        from .tasklist import TaskList
        self.pmesh = pm
        self.gids, self.gide = igids, igide
        self.nmb_thispack = igide - igids + 1
        self.pmb = None
        self.phydro = None
        self.pmhd = None
        self.tl_map = {}
        for name in ("before_timeintegrator", "after_timeintegrator", "before_stagen",
                     "stagen", "after_stagen"):
            self.tl_map[name] = TaskList()     # meshblock_pack.cpp:40-50

    def AddMeshBlocks(self, pin=None):
        self.pmb = MeshBlock(self, self.gids, self.nmb_thispack)

    def AddPhysics(self, pin):
        """MeshBlockPack::AddPhysics (src/meshblock_pack.cpp:102-262): <hydro> / <mhd>."""
        from .hydro import Hydro
        from .mhd import MHD
        nphys = 0
        if pin.DoesBlockExist("hydro"):
            self.phydro = Hydro(self, pin)
            nphys += 1
        if pin.DoesBlockExist("mhd"):
            self.pmhd = MHD(self, pin)
            nphys += 1
        if nphys == 0:
            raise RuntimeError("### FATAL ERROR At least one physics module must be specified "
                               "in input file (<hydro> or <mhd>)")
        if self.phydro is not None:
            self.phydro.AssembleHydroTasks(self.tl_map)
        if self.pmhd is not None:
            self.pmhd.AssembleMHDTasks(self.tl_map)


class Mesh:
    """src/mesh/mesh.hpp:92-185, mesh.cpp:46-330, build_tree.cpp (uniform root grid)."""

    def __init__(self, pin, my_rank=0, nranks=1):
        self.my_rank, self.nranks = my_rank, nranks
        self.pin = pin
        g = pin.GetReal
        self.mesh_size = RegionSize(g("mesh", "x1min"), g("mesh", "x1max"), g("mesh", "x2min"),
                                    g("mesh", "x2max"), g("mesh", "x3min"), g("mesh", "x3max"))
        ng = pin.GetOrAddInteger("mesh", "nghost", 2)
        nx1 = pin.GetInteger("mesh", "nx1")
        nx2 = pin.GetInteger("mesh", "nx2")
        nx3 = pin.GetInteger("mesh", "nx3")
        self.mesh_indcs = RegionIndcs(ng, nx1, nx2, nx3)
        self.one_d = nx2 == 1 and nx3 == 1
        self.two_d = nx2 > 1 and nx3 == 1
        self.three_d = nx3 > 1
        self.multi_d = nx2 > 1
        if nx2 == 1 and nx3 > 1:
            raise RuntimeError("### FATAL ERROR In mesh block in input file nx3>1 requires nx2>1")
        if ng < 2:
            raise RuntimeError("### FATAL ERROR More than 1 ghost zone required")
        bcs = []
        for name in ("ix1_bc", "ox1_bc", "ix2_bc", "ox2_bc", "ix3_bc", "ox3_bc"):
            v = pin.GetOrAddString("mesh", name, "periodic")
            if v not in _BCNAMES:
                raise RuntimeError("### FATAL ERROR boundary flag '%s' not supported on this "
                                   "path (periodic/outflow/reflect/inflow/diode/vacuum/user)" % v)
            bcs.append(_BCNAMES[v])
        self.mesh_bcs = bcs
        per = capi.BC["periodic"]
        self.strictly_periodic = (bcs[0] == per and bcs[1] == per and
                                  (not self.multi_d or (bcs[2] == per and bcs[3] == per)) and
                                  (not self.three_d or (bcs[4] == per and bcs[5] == per)))
        mbx1 = pin.GetOrAddInteger("meshblock", "nx1", nx1)
        mbx2 = pin.GetOrAddInteger("meshblock", "nx2", nx2)
        mbx3 = pin.GetOrAddInteger("meshblock", "nx3", nx3)
        if nx1 % mbx1 or nx2 % mbx2 or nx3 % mbx3:
            raise RuntimeError("### FATAL ERROR Mesh must be evenly divisible by MeshBlocks")
        self.mb_indcs = RegionIndcs(ng, mbx1, mbx2, mbx3)
        self.nmb_rootx1, self.nmb_rootx2, self.nmb_rootx3 = nx1//mbx1, nx2//mbx2, nx3//mbx3
        self.nmb_total = self.nmb_rootx1*self.nmb_rootx2*self.nmb_rootx3
        self.multilevel = False
        self.prolong_prims = False
        self.adaptive = False
        self.root_level = 0
        self.ptree = None
        ref = "none"
        if pin.DoesBlockExist("mesh_refinement"):
            ref = pin.GetOrAddString("mesh_refinement", "refinement", "none")
        if ref == "static":
            # Mesh::BuildTreeFromScratch with <refined_region*> blocks, build_tree.cpp:32-258
            from .mesh_tree import BuildTreeFromScratch
            # mesh_refinement.cpp:52: prolongate primitive instead of conserved variables into fine ghost zones
            self.prolong_prims = pin.GetOrAddBoolean("mesh_refinement", "prolong_primitives", False)
            if ng % 2:
                raise RuntimeError("### FATAL ERROR Number of ghost cells must be divisible by two for "
                                   "SMR/AMR calculations")
            self.ptree, ll, self.root_level, self.max_level = BuildTreeFromScratch(pin)
            self.multilevel = True
            self.lloc_eachmb = ll                      # LogicalLocation(lx1, lx2, lx3, level)
            self.gid_of_lloc = {tuple(l): i for i, l in enumerate(ll)}
            self.nmb_total = len(ll)
        elif ref != "none":
            raise RuntimeError("### FATAL ERROR <mesh_refinement>/refinement = '%s': only static "
                               "refinement is on this build's path" % ref)
        else:
            # Z-ordered list of logical locations (x1 fastest), build_tree.cpp:243-258
            ll = [(l1, l2, l3) for l3 in range(self.nmb_rootx3) for l2 in range(self.nmb_rootx2)
                  for l1 in range(self.nmb_rootx1)]
            ll.sort(key=lambda l: _morton(*l))
            self.lloc_eachmb = ll
            self.gid_of_lloc = {l: i for i, l in enumerate(ll)}
        if self.nmb_total < nranks:
            raise RuntimeError("### FATAL ERROR Fewer MeshBlocks (nmb_total=%d) than ranks "
                               "(nranks=%d)" % (self.nmb_total, nranks))
        self.cost_eachmb = [1.0]*self.nmb_total
        self.rank_eachmb, self.gids_eachrank, self.nmb_eachrank = LoadBalance(
            self.cost_eachmb, nranks)
        self.nmb_thisrank = self.nmb_eachrank[my_rank]
        # time: build_tree.cpp:301-302
        self.time = pin.GetOrAddReal("time", "start_time", 0.0)
        self.dt = FLT_MAX
        self.dtold = 0.0
        self.cfl_no = pin.GetReal("time", "cfl_number")
        self.ncycle = 0
        gids = self.gids_eachrank[my_rank]
        self.pmb_pack = MeshBlockPack(self, gids, gids + self.nmb_thisrank - 1)
        self.nmb_packs_thisrank = 1
        self.pmb_pack.AddMeshBlocks(pin)
        self.pgen = None

    def level_of(self, gid):
        """logical level of MeshBlock gid"""
        return self.lloc_eachmb[gid].level if self.multilevel else self.root_level

    def NumberOfMeshBlockCells(self):
        return self.mb_indcs.nx1*self.mb_indcs.nx2*self.mb_indcs.nx3

    def AddCoordinatesAndPhysics(self, pin):
        self.pmb_pack.AddPhysics(pin)

    def NewTimeStep(self, tlim):
        """Mesh::NewTimeStep, src/mesh/mesh.cpp:573-643."""
        self.dtold = self.dt
        if self.dt == FLT_MAX:
            self.dtold = 0.0
        self.dt = 2.0*self.dt
        pk = self.pmb_pack
        for ph in (pk.phydro, pk.pmhd):
            if ph is None:
                continue
            self.dt = min(self.dt, self.cfl_no*ph.dtnew)
            # viscosity, (MHD) resistivity, conduction: mesh.cpp:589-612
            for d in (ph.pvisc, ph.presist, ph.pcond):
                if d is not None:
                    self.dt = min(self.dt, self.cfl_no*d.dtnew)
        if self.nranks > 1:
            # MPI_Allreduce(MIN) of one Real, mesh.cpp:634-637
            import torch
            import torch.distributed as dist
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            from . import bvals as _bv
            import time as _t
            t0 = _t.perf_counter()
            t = torch.tensor([self.dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            self.dt = float(t.item())
            if _bv.HALO_PROF is not None:
                _bv.HALO_PROF.dt_reduce_s += _t.perf_counter() - t0
                _bv.HALO_PROF.dt_calls += 1
        if self.time < tlim and (self.time + self.dt) > tlim:
            self.dt = tlim - self.time
