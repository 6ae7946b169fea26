"""TEST INFRASTRUCTURE: an independent construction of the statically refined mesh and of the 56-slot neighbour
table, used to break the common mode between the product's hosts and the oracle.

The product builds its MeshBlockTree by walking an octree (athenak_amd/mesh_tree.py, csrc/akmi_host_smr.cpp:
restatements of src/mesh/meshblock_tree.cpp and MeshBlock::SetNeighbors).  This module does NOT walk a tree.
It works on integer boxes:

  * leaves: every MeshBlock is the box [lo, lo + size) on the lattice of the finest level.  Start from the root
    blocks; split every block that does not reach the level of a <refined_region*> it has to resolve (index
    ranges as Mesh::BuildTreeFromScratch computes them, src/mesh/build_tree.cpp:150-238, with the lever-rule
    LeftEdgeX of src/coordinates/cell_locations.hpp:24-27); then split every block that touches -- across a
    face, an edge or a corner, through periodic boundaries too -- a block more than one level finer, until
    nothing changes (the 2:1 rule the tree enforces in MeshBlockTree::AddNode).
  * order: Z-order = Morton order of the low corners (x1 fastest), which is what the depth-first walk of
    CreateZOrderedLLList produces.
  * neighbours: two blocks are neighbours when their closed boxes intersect and their interiors do not; per
    direction the contact is "touching below", "touching above" or "overlapping", which gives the offset
    (ox1, ox2, ox3).  A coarser block that overlaps a fine block's extent in a direction is that block's FACE
    neighbour, never its edge neighbour -- the rule the reference spells out as "only set neighbor for exterior
    edges of coarser face" (src/mesh/meshblock.cpp:283-289) falls out of the geometry.  Sub-block indices: of the
    finer block inside the coarser one's extent along the overlapping directions.  Slot numbers:
    NeighborIndex (src/mesh/nghbr_index.hpp:28-54); the destination slot is the slot of the opposite offset
    with the same sub-block indices (meshblock.cpp:195-214).

tests/parity_util.smr_tables hands THIS table to the oracle; the product keeps its own.
"""
import itertools

import numpy as np


def neighbor_index(ix, iy, iz, n1, n2):
    """NeighborIndex, src/mesh/nghbr_index.hpp:28-54"""
    if abs(ix) + abs(iy) + abs(iz) == 0 or abs(ix*iy*iz) > 1:
        return -1
    if iz == 0:
        if ix*iy == 0:
            return abs(ix)*2*(ix + 1) + abs(iy)*2*(iy + 5) + n1 + 2*n2
        return 16 + (ix + 1) + 2*(iy + 1) + n1
    if ix*iy == 0:
        return 24 + abs(ix)*(ix + 9) + abs(iy)*(iy + 17) + 2*(iz + 1) + n1 + 2*n2
    return 48 + (ix + 1)//2 + (iy + 1) + 2*(iz + 1)


def _left_edge(i, n, xmin, xmax):
    """LeftEdgeX, src/coordinates/cell_locations.hpp:24-27"""
    x = float(i)/float(n)
    return (x*xmax - x*xmin) - (0.5*xmax - 0.5*xmin) + (0.5*xmin + 0.5*xmax)


def _index_range(xmin, xmax, nroot, lev, rmin, rmax):
    """blocks of level `lev` (physical) covering [rmin, rmax], widened to whole sibling pairs
    (build_tree.cpp:150-165)"""
    n = nroot*(1 << lev)
    lo = 0
    while lo < n and not _left_edge(lo + 1, n, xmin, xmax) > rmin:
        lo += 1
    hi = lo
    while hi < n and not _left_edge(hi + 1, n, xmin, xmax) >= rmax:
        hi += 1
    if lo % 2 == 1:
        lo -= 1
    if hi % 2 == 0:
        hi += 1
    return lo, hi            # inclusive


def _morton(i, j, k):
    key = 0
    for b in range(21):
        key |= ((i >> b) & 1) << (3*b) | ((j >> b) & 1) << (3*b + 1) | ((k >> b) & 1) << (3*b + 2)
    return key


class Mesh:
    """leaves (lx1, lx2, lx3, logical level) in Z-order and nghbr[m][56] = (gid, level, dest)"""

    def __init__(self, pin):
        g, gi, gs = pin.GetReal, pin.GetInteger, pin.GetString
        nx = [gi("mesh", "nx%d" % q) for q in (1, 2, 3)]
        mb = [gi("meshblock", "nx%d" % q) if pin.DoesParameterExist("meshblock", "nx%d" % q) else nx[q - 1]
              for q in (1, 2, 3)]
        self.ndim = 1 + (nx[1] > 1) + (nx[2] > 1)
        self.nroot = [nx[q]//mb[q] for q in range(3)]
        xr = [(g("mesh", "x%dmin" % q), g("mesh", "x%dmax" % q)) for q in (1, 2, 3)]
        self.periodic = [gs("mesh", "ix%d_bc" % q) == "periodic" for q in (1, 2, 3)]
        self.root_level = 0
        while (1 << self.root_level) < max(self.nroot):
            self.root_level += 1
        active = [q < self.ndim for q in range(3)]
        # ---- what the refined regions ask for: (physical level, inclusive index box at that level)
        asks = []
        for name in pin.blocks:
            if not name.startswith("refined_region"):
                continue
            lev = gi(name, "level")
            box = []
            for q in range(3):
                if active[q]:
                    box.append(_index_range(xr[q][0], xr[q][1], self.nroot[q], lev, g(name, "x%dmin" % (q + 1)),
                                            g(name, "x%dmax" % (q + 1))))
                else:
                    box.append((0, 0))
            asks.append((lev, box))
        self.max_phys = max([a[0] for a in asks], default=0)
        L = self.max_phys
        dom = [self.nroot[q] << L if active[q] else 1 for q in range(3)]
        # a block: (physical level, lo1, lo2, lo3) on the finest lattice; size 2^(L - level) in active directions
        blocks = {(0, i << L, (j << L) if active[1] else 0, (k << L) if active[2] else 0)
                  for k in range(self.nroot[2]) for j in range(self.nroot[1]) for i in range(self.nroot[0])}

        def size(b):
            return 1 << (L - b[0])

        def box(b):
            s = size(b)
            return [(b[1 + q], b[1 + q] + (s if active[q] else 1)) for q in range(3)]

        def split(b):
            h = size(b)//2
            kids = set()
            for o in itertools.product(*[(0, 1) if active[q] else (0,) for q in range(3)]):
                kids.add((b[0] + 1, b[1] + o[0]*h, b[2] + o[1]*h, b[3] + o[2]*h))
            return kids

        # ---- resolve the regions
        changed = True
        while changed:
            changed = False
            for b in list(blocks):
                bb = box(b)
                for lev, rb in asks:
                    if b[0] >= lev:
                        continue
                    s = 1 << (L - lev)
                    if all((not active[q]) or (bb[q][0] < (rb[q][1] + 1)*s and rb[q][0]*s < bb[q][1]) for q in range(3)):
                        blocks.remove(b)
                        blocks |= split(b)
                        changed = True
                        break

        # ---- 2:1 closure through faces, edges, corners and periodic boundaries
        def touching(a, b):
            """closed boxes of a and (any periodic image of) b intersect"""
            ba, bb = box(a), box(b)
            for q in range(3):
                if not active[q]:
                    continue
                shifts = (-dom[q], 0, dom[q]) if self.periodic[q] else (0,)
                if not any(ba[q][0] <= bb[q][1] + sh and bb[q][0] + sh <= ba[q][1] for sh in shifts):
                    return False
            return True

        changed = True
        while changed:
            changed = False
            bl = sorted(blocks)
            for a in bl:
                for b in bl:
                    if b in blocks and a in blocks and b[0] < a[0] - 1 and touching(a, b):
                        blocks.remove(b)
                        blocks |= split(b)
                        changed = True
        self._L, self._dom, self._active, self._box, self._size = L, dom, active, box, size
        self.blocks = sorted(blocks, key=lambda b: _morton(b[1], b[2], b[3]))
        self.lloc = [(b[1] >> (L - b[0]), (b[2] >> (L - b[0])) if active[1] else 0, (b[3] >> (L - b[0])) if active[2] else 0,
                      b[0] + self.root_level) for b in self.blocks]
        self.nghbr = self._neighbours()

    def _neighbours(self):
        L, dom, active, box = self._L, self._dom, self._active, self._box
        nmb = len(self.blocks)
        out = -np.ones((nmb, 56, 3), dtype=np.int32)
        images = list(itertools.product(*[((-1, 0, 1) if (active[q] and self.periodic[q]) else (0,)) for q in range(3)]))
        for m, a in enumerate(self.blocks):
            ba = box(a)
            for gid, b in enumerate(self.blocks):
                for img in images:
                    if gid == m and img == (0, 0, 0):
                        continue
                    bb = [(box(b)[q][0] + img[q]*dom[q], box(b)[q][1] + img[q]*dom[q]) for q in range(3)]
                    off, ok = [0, 0, 0], True
                    for q in range(3):
                        if not active[q]:
                            continue
                        if bb[q][1] == ba[q][0]:
                            off[q] = -1
                        elif bb[q][0] == ba[q][1]:
                            off[q] = 1
                        elif bb[q][0] < ba[q][1] and ba[q][0] < bb[q][1]:
                            off[q] = 0
                        else:
                            ok = False
                            break
                    if not ok or off == [0, 0, 0]:
                        continue
                    # sub-block indices along the overlapping directions: position of the finer box in the coarser
                    f = []
                    for q in range(3):
                        if not active[q] or off[q] != 0:
                            continue
                        if b[0] > a[0]:            # finer neighbour: which half of MY extent it covers
                            f.append(0 if bb[q][0] == ba[q][0] else 1)
                        elif b[0] < a[0]:          # coarser neighbour: which half of ITS extent I cover
                            f.append(0 if ba[q][0] == bb[q][0] else 1)
                        else:
                            f.append(0)
                    assert abs(b[0] - a[0]) <= 1, "2:1 rule violated"
                    if b[0] == a[0]:
                        f = [0, 0]
                    f = (f + [0, 0])[:2]
                    n = neighbor_index(off[0], off[1], off[2], f[0], f[1])
                    d = neighbor_index(-off[0], -off[1], -off[2], f[0], f[1])
                    assert out[m, n, 0] < 0, "two neighbours in one slot"
                    out[m, n] = (gid, b[0] + self.root_level, d)
        return out
