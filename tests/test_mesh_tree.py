"""not gpu: the MeshBlockTree of statically refined meshes (first piece of SMR) -- root
grids in the order Mesh already uses, the block count of the reference's SMR test deck, the 2:1 rule,
exact tiling of the domain, and the consistency of the 56-slot neighbour table."""
import itertools

import pytest

from athenak_amd import mesh_tree as mt
from athenak_amd.mesh import _morton
from athenak_amd.parameter_input import ParameterInput

# mesh, MeshBlock and refined region of the reference's inputs/tests/linear_wave_hydro_smr.athinput
SMR3D = """
<mesh>
nghost = 2
nx1 = 64
x1min = 0.0
x1max = 3.0
ix1_bc = periodic
ox1_bc = periodic
nx2 = 32
x2min = 0.0
x2max = 1.5
ix2_bc = periodic
ox2_bc = periodic
nx3 = 32
x3min = 0.0
x3max = 1.5
ix3_bc = periodic
ox3_bc = periodic
<meshblock>
nx1 = 16
nx2 = 8
nx3 = 8
<mesh_refinement>
refinement = static
<refined_region1>
level = 1
x1min = 1.2
x1max = 1.8
x2min = 0.7
x2max = 0.8
x3min = 0.7
x3max = 0.8
"""

NESTED2D = """
<mesh>
nghost = 2
nx1 = 64
x1min = -1.0
x1max = 1.0
ix1_bc = outflow
ox1_bc = outflow
nx2 = 64
x2min = -1.0
x2max = 1.0
ix2_bc = periodic
ox2_bc = periodic
nx3 = 1
x3min = -0.5
x3max = 0.5
ix3_bc = periodic
ox3_bc = periodic
<meshblock>
nx1 = 8
nx2 = 8
<mesh_refinement>
refinement = static
<refined_region1>
level = 3
x1min = -0.05
x1max = 0.05
x2min = 0.8
x2max = 0.95
<refined_region2>
level = 1
x1min = 0.5
x1max = 0.9
x2min = -0.9
x2max = -0.5
"""

ONE_D = """
<mesh>
nghost = 2
nx1 = 96
x1min = 0.0
x1max = 1.0
ix1_bc = reflect
ox1_bc = reflect
nx2 = 1
x2min = 0.0
x2max = 1.0
ix2_bc = periodic
ox2_bc = periodic
nx3 = 1
x3min = 0.0
x3max = 1.0
ix3_bc = periodic
ox3_bc = periodic
<meshblock>
nx1 = 16
<mesh_refinement>
refinement = static
<refined_region1>
level = 2
x1min = 0.4
x1max = 0.45
"""


def _build(text):
    tree, lloc, root, maxl = mt.BuildTreeFromScratch(ParameterInput(text=text))
    return tree, lloc, root, maxl


@pytest.mark.parametrize("nmb", [(4, 4, 4), (3, 2, 5), (6, 1, 1), (5, 3, 1), (1, 1, 1), (8, 2, 1)])
def test_root_grid_is_in_the_order_mesh_uses(nmb):
    ndim = 3 if nmb[2] > 1 else (2 if nmb[1] > 1 else 1)
    tree = mt.MeshBlockTree(nmb, [True]*6).setup(ndim)
    lloc = tree.CreateZOrderedLLList()
    want = sorted(itertools.product(range(nmb[0]), range(nmb[1]), range(nmb[2])), key=lambda l: _morton(*l))
    assert [(l.lx1, l.lx2, l.lx3) for l in lloc] == want
    assert all(l.level == tree.root_level for l in lloc)


def test_neighbor_index_covers_the_56_slots_once():
    seen = {}
    for o in itertools.product((-1, 0, 1), repeat=3):
        nz = sum(1 for v in o if v)
        if nz == 0:
            continue
        subs = {1: [(a, b) for a in (0, 1) for b in (0, 1)], 2: [(0, 0), (1, 0)], 3: [(0, 0)]}[nz]
        for n1, n2 in subs:
            s = mt.NeighborIndex(o[0], o[1], o[2], n1, n2)
            assert 0 <= s < 56 and s not in seen
            seen[s] = (o, n1, n2)
    assert len(seen) == 56
    assert mt.NeighborIndex(0, 0, 0, 0, 0) == -1


def _volume(lloc, tree, ndim):
    v = 0.0
    for l in lloc:
        f = 1.0
        for d in range(ndim):
            f /= tree.nmb_root[d] << (l.level - tree.root_level)
        v += f
    return v


def _check_tables(tree, lloc, ndim):
    ranks = [0]*len(lloc)
    tabs = [mt.SetNeighbors(tree, l, ranks) for l in lloc]
    for g, (l, tab) in enumerate(zip(lloc, tabs)):
        assert tab, "every block has neighbours"
        for slot, nb in tab.items():
            assert abs(nb.lev - l.level) <= 1, "2:1 rule"
            assert lloc[nb.gid].level == nb.lev
            # the neighbour lists this block back in the slot this block sends to
            back = tabs[nb.gid].get(nb.dest)
            assert back is not None and back.gid == g and back.dest == slot, (g, slot, nb, back)
    return tabs


def test_reference_smr_deck_has_120_blocks():
    tree, lloc, root, maxl = _build(SMR3D)
    assert (root, maxl) == (2, 3)
    assert len(lloc) == 120
    assert sum(1 for l in lloc if l.level == 3) == 64 and sum(1 for l in lloc if l.level == 2) == 56
    assert abs(_volume(lloc, tree, 3) - 1.0) < 1e-14
    tabs = _check_tables(tree, lloc, 3)
    # a root block next to the refined region sees four finer blocks through that face
    g = next(i for i, l in enumerate(lloc) if (l.lx1, l.lx2, l.lx3, l.level) == (0, 1, 1, 2))
    face = [tabs[g][mt.NeighborIndex(1, 0, 0, a, b)] for a in (0, 1) for b in (0, 1)]
    assert all(nb.lev == 3 for nb in face) and len({nb.gid for nb in face}) == 4
    # gids are the positions in the Z-ordered list
    assert [tree.FindMeshBlock(l).gid for l in lloc] == list(range(120))


def test_nested_2d_regions_force_the_intermediate_levels():
    tree, lloc, root, maxl = _build(NESTED2D)
    assert (root, maxl) == (3, 6)
    levels = sorted({l.level for l in lloc})
    assert levels == [3, 4, 5, 6], "a level-3 region inside a root grid needs rings of levels 1 and 2"
    assert abs(_volume(lloc, tree, 2) - 1.0) < 1e-14
    tabs = _check_tables(tree, lloc, 2)
    # outflow x1: blocks on the x1 faces of the mesh have no neighbour there; periodic x2 wraps
    left = [g for g, l in enumerate(lloc) if l.lx1 == 0]
    assert left and all(mt.NeighborIndex(-1, 0, 0, 0, 0) not in tabs[g] for g in left)
    bottom = [g for g, l in enumerate(lloc) if l.lx2 == 0 and l.level == 3]
    assert bottom and all(any(s in tabs[g] for s in (8, 9)) for g in bottom)
    # the fine region touches the upper x2 boundary zone: refinement propagated across the periodic face
    assert any(l.level > 3 and l.lx2 == 0 for l in lloc) or all(l.level == 3 for l in lloc if l.lx2 == 0)


def test_1d_tree():
    tree, lloc, root, maxl = _build(ONE_D)
    assert (root, maxl) == (3, 5)                       # 6 root blocks -> root level 3
    assert abs(_volume(lloc, tree, 1) - 1.0) < 1e-14
    _check_tables(tree, lloc, 1)
    lv = [l.level for l in lloc]
    assert max(lv) == 5 and all(abs(a - b) <= 1 for a, b in zip(lv, lv[1:]))
    # left to right in x
    pos = [l.lx1/float(6 << (l.level - 3)) for l in lloc]
    assert pos == sorted(pos)


def test_refined_region_errors():
    with pytest.raises(RuntimeError, match="fully contained"):
        _build(SMR3D.replace("x1max = 1.8", "x1max = 3.8"))
    with pytest.raises(RuntimeError, match="larger than 0"):
        _build(SMR3D.replace("level = 1", "level = 0"))
    with pytest.raises(RuntimeError, match="divisible by 2"):
        _build(SMR3D.replace("nx1 = 64", "nx1 = 60").replace("nx1 = 16", "nx1 = 15"))


# ---- round 3: the product's tree and neighbour table against a construction that does not walk a tree -------
import pytest as _pytest

_TREE_CASES = [
    ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 4, 4), ()),
    ("linear_wave_mhd_smr", (32, 16, 16), 3, (8, 4, 4), ("refined_region1/level=2",)),            # three levels
    ("linear_wave_hydro_smr", (32, 16, 1), 2, (8, 4, 1), ()),
    ("linear_wave_hydro_smr", (32, 1, 1), 1, (8, 1, 1), ()),
    ("blast_smr", (32, 32, 32), 3, (8, 8, 8), ()),
    ("blast_smr", (64, 64, 64), 3, (16, 16, 16), ()),                                              # config 5's own deck: 120 blocks
    ("blast_smr", (32, 32, 32), 3, (8, 8, 8), ("mesh/ix1_bc=outflow", "mesh/ox1_bc=outflow", "mesh/ix2_bc=reflect",
                                                "mesh/ox2_bc=reflect", "refined_region1/x1min=-0.5",
                                                "refined_region1/x1max=-0.2", "refined_region1/x2min=-0.5",
                                                "refined_region1/x2max=-0.2")),                   # region on a physical boundary
    ("linear_wave_mhd_smr", (24, 12, 12), 3, (4, 4, 4), ("refined_region1/level=2",)),           # 6x3x3 root grid: not a power of two
]


@_pytest.mark.parametrize("case", _TREE_CASES, ids=lambda c: "%s-%s-mb%s%s" % (c[0], c[1], c[3], "-x" if c[4] else ""))
def test_tree_and_neighbour_table_against_integer_box_construction(case):
    """tests/independent_tree.py (integer boxes, contact classification; src/mesh/build_tree.cpp:150-238,
    src/mesh/meshblock.cpp:142-425 semantics) gives the same Z-ordered leaves and the same 56-slot table
    {gid, level, dest} as the product's tree walk, slot for slot"""
    import numpy as np
    import independent_tree
    import parity_util as pu
    from athenak_amd.main import load_deck
    from athenak_amd.mesh import Mesh
    problem, n, dims, mb, extra = case
    deck, ov = pu.deck_overrides(problem, n, dims, mb, extra=extra)
    pin = load_deck(deck, ov)
    pm = Mesh(pin)
    it = independent_tree.Mesh(pin)
    lloc, ng = pu.product_smr_tables(pm)
    assert np.array_equal(np.array(it.lloc, dtype=np.int32), lloc)
    assert it.root_level == pm.root_level
    nn = {1: 8, 2: 24, 3: 56}[dims]
    assert np.array_equal(it.nghbr[:, :nn], ng[:, :nn]), np.argwhere(it.nghbr[:, :nn] != ng[:, :nn])[:10]
    assert len(set(l[3] for l in it.lloc)) >= 2
