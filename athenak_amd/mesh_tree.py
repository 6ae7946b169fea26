"""MeshBlockTree: the octree (quadtree / binary tree) of MeshBlocks of a statically refined mesh.

Host-side restatement of src/mesh/meshblock_tree.cpp (CreateRootGrid :64-93, AddNode :98-116,
Refine with its 2:1 balancing :152-250, CreateZOrderedLLList :336-353, FindNeighbor :360-459), of the
<refined_region*> handling of Mesh::BuildTreeFromScratch (src/mesh/build_tree.cpp:32-258) and of the
56-slot neighbour table of MeshBlock::SetNeighbors (src/mesh/meshblock.cpp:142-425,
src/mesh/nghbr_index.hpp:28-54).

Status: first piece of SURVEY section 8(f) item 1.  The tree, the Z-ordered leaf list and the neighbour
table with levels are complete and tested on their invariants; the level-aware ghost exchange, flux and
EMF correction that consume them are not built yet, so Mesh still refuses `refinement = static`.
"""
from collections import namedtuple

from .mesh import LeftEdgeX

LogicalLocation = namedtuple("LogicalLocation", "lx1 lx2 lx3 level")
NeighborBlock = namedtuple("NeighborBlock", "gid lev rank dest")


def NeighborIndex(ix, iy, iz, n1, n2):
    """slot of the boundary buffer towards offset (ix,iy,iz), sub-block (n1,n2): nghbr_index.hpp:28-54
    (x1faces 0-7, x2faces 8-15, x1x2 edges 16-23, x3faces 24-31, x3x1 edges 32-39, x2x3 edges 40-47,
    corners 48-55)"""
    if abs(ix) + abs(iy) + abs(iz) == 0 or abs(ix*iy*iz) > 1:
        return -1
    if iz == 0:
        if ix*iy == 0:
            return abs(ix)*2*(ix + 1) + abs(iy)*2*(iy + 5) + n1 + 2*n2
        return 16 + (ix + 1) + 2*(iy + 1) + n1
    if ix*iy == 0:
        return 24 + abs(ix)*(ix + 9) + abs(iy)*(iy + 17) + 2*(iz + 1) + n1 + 2*n2
    return 48 + (ix + 1)//2 + (iy + 1) + 2*(iz + 1)


class _Node:
    __slots__ = ("lloc", "leaf", "gid")

    def __init__(self, lloc):
        self.lloc = lloc
        self.leaf = None          # list of 2 / 4 / 8 children (entries may be None in the root grid)
        self.gid = -1


class MeshBlockTree:
    """the tree of one Mesh.  nmb_root: MeshBlocks of the root grid per direction; periodic: six
    booleans in BoundaryFace order (a shear-periodic x1 face counts as periodic, :206-217)"""

    def __init__(self, nmb_root, periodic):
        self.nmb_root = tuple(int(n) for n in nmb_root)
        self.periodic = tuple(bool(p) for p in periodic)
        self.ndim = 1
        self.root_level = 0
        self.root = _Node(LogicalLocation(0, 0, 0, 0))

    # the caller states the dimensionality (a direction with one root block may still be active)
    def setup(self, ndim):
        self.ndim = ndim
        self.nleaf = 1 << ndim
        nmax = max(self.nmb_root)
        self.root_level = 0
        while (1 << self.root_level) < nmax:                 # build_tree.cpp:44
            self.root_level += 1
        self._create_root(self.root)
        return self

    def _child(self, node, n):
        i, j, k = n & 1, (n >> 1) & 1, (n >> 2) & 1
        l = node.lloc
        return _Node(LogicalLocation(l.lx1*2 + i, l.lx2*2 + j, l.lx3*2 + k, l.level + 1))

    def _create_root(self, node):
        """CreateRootGrid: the root grid may be incomplete (fewer than nleaf children)"""
        if node.lloc.level == self.root_level:
            return
        node.leaf = [None]*self.nleaf
        levfac = 1 << (self.root_level - node.lloc.level - 1)
        for n in range(self.nleaf):
            c = self._child(node, n)
            if (c.lloc.lx3*levfac < self.nmb_root[2] and c.lloc.lx2*levfac < self.nmb_root[1]
                    and c.lloc.lx1*levfac < self.nmb_root[0]):
                node.leaf[n] = c
                self._create_root(c)

    @staticmethod
    def _leaf_index(rloc, level):
        sh = rloc.level - level - 1
        return ((rloc.lx1 >> sh) & 1) + (((rloc.lx2 >> sh) & 1) << 1) + (((rloc.lx3 >> sh) & 1) << 2)

    def AddNode(self, rloc, node=None):
        """create the MeshBlock at rloc, refining on the way down (each refinement also creates the
        same-level neighbours of the refined block: the 2:1 rule)"""
        node = node or self.root
        while node.lloc.level != rloc.level:
            if node.leaf is None:
                self.Refine(node)
            node = node.leaf[self._leaf_index(rloc, node.lloc.level)]

    def _wrap(self, l, d, level):
        """logical index l in direction d at `level` after the mesh boundary: None outside a
        non-periodic face"""
        n = self.nmb_root[d] << (level - self.root_level)
        if l < 0:
            return n - 1 if self.periodic[2*d] else None
        if l >= n:
            return 0 if self.periodic[2*d + 1] else None
        return l

    def Refine(self, node):
        if node.leaf is not None:
            return
        node.leaf = [self._child(node, n) for n in range(self.nleaf)]
        l = node.lloc
        rng = [(-1, 0, 1) if d < self.ndim else (0,) for d in range(3)]
        for oz in rng[2]:
            z = self._wrap(l.lx3 + oz, 2, l.level)
            if z is None:
                continue
            for oy in rng[1]:
                y = self._wrap(l.lx2 + oy, 1, l.level)
                if y is None:
                    continue
                for ox in rng[0]:
                    if ox == 0 and oy == 0 and oz == 0:
                        continue
                    x = self._wrap(l.lx1 + ox, 0, l.level)
                    if x is None:
                        continue
                    self.AddNode(LogicalLocation(x, y, z, l.level))
        node.gid = -1

    def CreateZOrderedLLList(self):
        """leaves in tree-traversal (Z) order; assigns gids"""
        out = []

        def walk(node):
            if node.leaf is None:
                node.gid = len(out)
                out.append(node.lloc)
            else:
                for c in node.leaf:
                    if c is not None:
                        walk(c)
        walk(self.root)
        return out

    def FindMeshBlock(self, tloc):
        node = self.root
        while node.lloc.level != tloc.level:
            if node.leaf is None:
                return None
            node = node.leaf[self._leaf_index(tloc, node.lloc.level)]
            if node is None:
                return None
        return node

    def FindNeighbor(self, myloc, ox1, ox2, ox3):
        """the block touching myloc in direction (ox1,ox2,ox3): itself if it is a leaf of the same or
        the coarser level, its PARENT node if the neighbours are finer; None at a mesh boundary"""
        ll = myloc.level
        lx = self._wrap(myloc.lx1 + ox1, 0, ll)
        ly = self._wrap(myloc.lx2 + ox2, 1, ll)
        lz = self._wrap(myloc.lx3 + ox3, 2, ll)
        if lx is None or ly is None or lz is None:
            return None
        if ll < 1:
            return self.root
        bt = self.root
        for level in range(ll):
            if bt.leaf is None:
                if level == ll - 1:
                    return bt                                  # coarser neighbour
                raise RuntimeError("### FATAL ERROR Neighbor search failed; MeshBlockTree broken.")
            sh = ll - level - 1
            bt = bt.leaf[((lx >> sh) & 1) + (((ly >> sh) & 1) << 1) + (((lz >> sh) & 1) << 2)]
            if bt is None:
                raise RuntimeError("### FATAL ERROR Neighbor search failed; MeshBlockTree broken.")
        if bt.leaf is None:
            return bt
        probe = bt.leaf[(1 if ox1 < 0 else 0) + ((1 if ox2 < 0 else 0) << 1) + ((1 if ox3 < 0 else 0) << 2)]
        if probe.leaf is not None:
            raise RuntimeError("### FATAL ERROR Neighbor search failed. The Block Tree is broken.")
        return bt


def SetNeighbors(tree, lloc, ranklist, multilevel=True):
    """the neighbour table of the MeshBlock at lloc: {slot: NeighborBlock}, slots of NeighborIndex.
    meshblock.cpp:142-425 written once for all directions: the free directions of an offset (those with
    offset 0) index the sub-blocks -- of a finer neighbour all of them, of a coarser neighbour the one
    this block occupies on its parent; a coarser edge/corner neighbour exists only where this block
    sits in that corner of its parent."""
    ndim = tree.ndim
    my = (lloc.lx1, lloc.lx2, lloc.lx3)
    myf = [my[d] & 1 for d in range(3)]
    myo = [(my[d] & 1)*2 - 1 if d < ndim else 0 for d in range(3)]
    nf = [2 if (multilevel and d < ndim) else 1 for d in range(3)]
    out = {}
    rng = [(-1, 0, 1) if d < ndim else (0,) for d in range(3)]
    for oz in rng[2]:
        for oy in rng[1]:
            for ox in rng[0]:
                o = (ox, oy, oz)
                if o == (0, 0, 0):
                    continue
                nt = tree.FindNeighbor(lloc, ox, oy, oz)
                if nt is None:
                    continue
                free = [d for d in range(3) if o[d] == 0]
                neg = tuple(-v for v in o)

                def slot(off, f):
                    n1 = f[0] if len(f) > 0 else 0
                    n2 = f[1] if len(f) > 1 else 0
                    return NeighborIndex(off[0], off[1], off[2], n1, n2)
                if nt.leaf is not None:                        # finer: every touching child
                    touch = [1 - (o[d] + 1)//2 if o[d] != 0 else None for d in range(3)]
                    f1s = range(nf[free[0]]) if len(free) > 0 else (0,)
                    f2s = range(nf[free[1]]) if len(free) > 1 else (0,)
                    for f2 in f2s:
                        for f1 in f1s:
                            idx = list(touch)
                            if len(free) > 0:
                                idx[free[0]] = f1
                            if len(free) > 1:
                                idx[free[1]] = f2
                            c = nt.leaf[idx[0] + (idx[1] << 1) + (idx[2] << 2)]
                            f = (f1, f2)[:len(free)]
                            out[slot(o, f)] = NeighborBlock(c.gid, c.lloc.level, ranklist[c.gid],
                                                            slot(neg, f))
                elif nt.lloc.level == lloc.level:
                    out[slot(o, ())] = NeighborBlock(nt.gid, nt.lloc.level, ranklist[nt.gid],
                                                     slot(neg, ()))
                else:                                          # coarser
                    nfree = len(free)
                    if nfree < 2 and not all(myo[d] == o[d] for d in range(3) if o[d] != 0):
                        continue
                    f = tuple(myf[d] for d in free)
                    out[slot(o, f)] = NeighborBlock(nt.gid, nt.lloc.level, ranklist[nt.gid],
                                                    slot(neg, f))
    return out


def BuildTreeFromScratch(pin):
    """tree + Z-ordered leaves of the mesh a ParameterInput describes, with its <refined_region*>
    blocks when <mesh_refinement>/refinement = static (build_tree.cpp:32-258).  Returns
    (tree, lloc_eachmb, root_level, max_level)."""
    gi, gr, gs = pin.GetInteger, pin.GetReal, pin.GetString
    nx = [gi("mesh", "nx%d" % q) for q in (1, 2, 3)]
    mb = [gi("meshblock", "nx%d" % q) if pin.DoesParameterExist("meshblock", "nx%d" % q) else nx[q - 1]
          for q in (1, 2, 3)]
    ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
    names = ("ix1_bc", "ox1_bc", "ix2_bc", "ox2_bc", "ix3_bc", "ox3_bc")
    periodic = [q < 2*ndim and gs("mesh", names[q]) in ("periodic", "shear_periodic") for q in range(6)]
    nmb_root = [nx[d]//mb[d] for d in range(3)]
    tree = MeshBlockTree(nmb_root, periodic).setup(ndim)
    root_level = tree.root_level
    current_level = root_level
    multilevel = pin.DoesBlockExist("mesh_refinement") and \
        gs("mesh_refinement", "refinement") in ("static", "adaptive")
    if multilevel:
        if any(mb[d] % 2 for d in range(ndim)):
            raise RuntimeError("### FATAL ERROR Number of cells in MeshBlock must be divisible by 2 "
                               "with SMR or AMR.")
        xmin = [gr("mesh", "x%dmin" % q) for q in (1, 2, 3)]
        xmax = [gr("mesh", "x%dmax" % q) for q in (1, 2, 3)]
        for name in pin.blocks:
            if not name.startswith("refined_region"):
                continue
            rmin, rmax = list(xmin), list(xmax)
            for d in range(ndim):
                rmin[d] = gr(name, "x%dmin" % (d + 1))
                rmax[d] = gr(name, "x%dmax" % (d + 1))
            phy = gi(name, "level")
            if phy < 1:
                raise RuntimeError("### FATAL ERROR <refined_region> level must be larger than 0 "
                                   "(root level=0)")
            if any(rmin[d] > rmax[d] for d in range(3)):
                raise RuntimeError("### FATAL ERROR Invalid <refined_region> (xmax < xmin in one "
                                   "direction).")
            if any(rmin[d] < xmin[d] or rmax[d] > xmax[d] for d in range(3)):
                raise RuntimeError("### FATAL ERROR <refined_region> must be fully contained within "
                                   "root mesh")
            log = phy + root_level
            current_level = max(current_level, log)
            lo, hi = [0, 0, 0], [1, 1, 1]
            for d in range(ndim):
                lxmax = nmb_root[d]*(1 << phy)
                a = 0
                while a < lxmax and not float(LeftEdgeX(a + 1, lxmax, xmin[d], xmax[d])) > rmin[d]:
                    a += 1
                b = a
                while b < lxmax and not float(LeftEdgeX(b + 1, lxmax, xmin[d], xmax[d])) >= rmax[d]:
                    b += 1
                if a % 2 == 1:
                    a -= 1
                if b % 2 == 0:
                    b += 1
                lo[d], hi[d] = a, b
            for k in range(lo[2], hi[2], 2):
                for j in range(lo[1], hi[1], 2):
                    for i in range(lo[0], hi[0], 2):
                        tree.AddNode(LogicalLocation(i, j, k, log))
    lloc = tree.CreateZOrderedLLList()
    return tree, lloc, root_level, current_level
