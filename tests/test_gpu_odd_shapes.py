"""GPU: MeshBlocks that are not cubes and not multiples of a wave -- the lane mappings of the fused stage
(flattened rows, waves overlapping by one, two or four lanes, marches chopped into chunks) and of the
task-granular entry points (sweeps in store-all mode, marching CornerE) against the oracle, bit for bit."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_util as pu  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [
    # problem, mesh, MeshBlock, kwargs
    ("orszag_tang", (40, 24, 56), (20, 12, 28), dict(cfl=0.3)),                       # 8 blocks, three different edges
    ("orszag_tang", (36, 36, 20), (36, 12, 10), dict(cfl=0.3, ng=3)),                 # rows of 42 cells, thin in x3
    ("orszag_tang", (66, 10, 14), (66, 10, 14), dict(cfl=0.3)),                       # one long thin block: rows wider than a wave
    ("orszag_tang", (28, 44, 12), (14, 22, 6), dict(cfl=0.3, recon="ppm4", ng=4)),    # five-point scheme
    ("orszag_tang", (24, 40, 16), (12, 20, 8), dict(cfl=0.3, recon="wenoz", ng=3)),
    ("blast", (30, 18, 42), (10, 18, 14), dict(cfl=0.3)),                             # MHD blast wave, strong shock
    ("sod", (52, 12, 20), (26, 6, 10), dict(cfl=0.3, recon="ppm4", ng=3)),            # outflow boundaries
]


@pytest.mark.parametrize("native", [False, True], ids=["python-host", "cpp-host"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "split"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%dx%dx%d-mb%dx%dx%d" % ((c[0],) + c[1] + c[2]))
def test_odd_shapes_are_bit_identical(case, fused, native):
    problem, n, mb, kw = case
    res = pu.compare_run(problem, n, 3, mb, 3, fused=fused, native=native, **kw)
    assert res["cycles"] == 3
    assert res["time"][0] == res["time"][1], res["time"]
    assert res["bitwise_equal"], res["diffs"]
